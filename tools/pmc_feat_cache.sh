#!/bin/bash
# cache-side counters of the featuriser (experiment): one counter group per pass, no other tracing
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
for c in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_LATENCY_sum" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmcc_$tag -o p -- python bench.py --no-extra --no-configs2 --no-cpu-baseline --steps 3 --warmup 1 --repeat 1 > $O/pmcc_$tag.log 2>&1 < /dev/null
  echo "== $c"
  if [ -f $O/pmcc_$tag/p_counter_collection.csv ]; then python tools/pmc_summary.py $O/pmcc_$tag/p_counter_collection.csv 2>/dev/null | grep -A4 "k_featurize" | head -6; else tail -2 $O/pmcc_$tag.log; fi
  rm -f $O/pmcc_$tag/p_kernel_trace.csv
done
