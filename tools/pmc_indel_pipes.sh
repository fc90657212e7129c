#!/bin/bash
# which pipe is busy in every kernel of the indel pass (experiment): one counter group per pass, no other tracing
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
for c in "TA_BUSY_avr GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmcp_$tag -o p -- python tools/bench_indel_pipe.py 64444167 2 > $O/pmcp_$tag.log 2>&1 < /dev/null
  rm -f $O/pmcp_$tag/p_kernel_trace.csv
done
ls $O | grep pmcp_ | head -20
