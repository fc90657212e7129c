#!/bin/bash
# ablation timing of k_event_tiles' phases (experiment): variants built by SRC=nc_indel tools/exp_build.sh evA2 "-DNC_ABL_EVT_A2" ...
for v in "$@"; do
  echo "== $v"; NANOCALLER_HIP_LIB=$GRAFT_REPO_ROOT/build_exp/libnc_$v.so TOP=40 tools/prof_quick.sh $v python tools/bench_indel_pipe.py 64444167 2 2>&1 | grep -E "k_event_tiles"
done
