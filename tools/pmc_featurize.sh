#!/bin/bash
# counters of the two featuriser kernels (k_featurize: scalar-driven read walk; k_featurize_pairs: lanes as read x column pairs), separate passes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
for f in 0 1; do
  for c in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE"; do
    tag=$(echo $c | tr ' ' '_')
    NC_FEAT_PAIRS=$f timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmcf_${f}_$tag -o p -- python bench.py --no-extra --no-configs2 --no-cpu-baseline --steps 3 --warmup 1 --repeat 1 > /dev/null 2>&1 < /dev/null
    python tools/pmc_summary.py $O/pmcf_${f}_$tag/p_counter_collection.csv 2>/dev/null | grep -A3 "k_featurize" | head -4
    rm -f $O/pmcf_${f}_$tag/p_kernel_trace.csv
  done
done
