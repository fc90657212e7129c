#!/bin/bash
# per-kernel times of the indel pipeline under variant builds (experiment): tools/abl_lib.sh "<kernel regex>" NAME ...  (build_exp/libnc_NAME.so; NAME "-" = the in-tree build)
pat=$1; shift
for v in "$@"; do
  echo "== $v"
  if [ "$v" = "-" ]; then TOP=60 tools/prof_quick.sh base python tools/bench_indel_pipe.py 64444167 3 2>&1 | grep -E "$pat"
  else NANOCALLER_HIP_LIB=$GRAFT_REPO_ROOT/build_exp/libnc_$v.so TOP=60 tools/prof_quick.sh $v python tools/bench_indel_pipe.py 64444167 3 2>&1 | grep -E "$pat"; fi
done
