#!/bin/bash
# counters of k_featurize_quad / k_featurize_pairs (experiment): one counter group per pass, no other tracing
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
for q in 1 0; do
for c in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "TA_BUSY_avr GRBM_GUI_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  tag=q${q}_$(echo $c | tr ' ' '_' | cut -c1-40)
  NC_FEAT_QUAD=$q timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmcq_$tag -o p -- python bench.py --no-extra --no-configs2 --no-cpu-baseline --steps 3 --warmup 1 --repeat 1 > $O/pmcq_$tag.log 2>&1 < /dev/null
  echo "== quad=$q $c"
  if [ -f $O/pmcq_$tag/p_counter_collection.csv ]; then python tools/pmc_summary.py $O/pmcq_$tag/p_counter_collection.csv 2>/dev/null | grep -A3 "k_featurize" | head -4; else tail -2 $O/pmcq_$tag.log; fi
  rm -f $O/pmcq_$tag/p_kernel_trace.csv
done
done
