#!/bin/bash
# quick per-kernel times of one command on the GPU box: tools/prof_quick.sh TAG <command...>  -> gpurun_out/TAG_stats.csv (trimmed), top lines on stdout
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o p -- "$@" > $O/${TAG}_out.txt 2>/dev/null < /dev/null
if [ -f $O/prof_$TAG/p_kernel_stats.csv ]; then
  python tools/trim_rocprof.py $O/prof_$TAG/p_kernel_stats.csv $O/${TAG}_stats.csv < /dev/null
  head -${TOP:-40} $O/${TAG}_stats.csv | cut -c1-140
else
  echo "no stats file"; ls $O/prof_$TAG 2>/dev/null
fi
rm -rf $O/prof_$TAG/p_kernel_trace.csv
grep -E "stages|rep 2" $O/${TAG}_out.txt
