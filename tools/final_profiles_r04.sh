#!/bin/bash
# the round's LAST profile pass (after the kernels froze): default bench line, the indel pipeline's kernel stats and its FETCH_SIZE / WRITE_SIZE passes
# (the SNP kernels did not change after r04a: profiles/r04a_kernel_stats.csv and trunk_traffic.json stand)
set -x
TAG=${1:-r04b}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/profi_$TAG -o p -- python tools/bench_indel_pipe.py 64444167 4 > $O/${TAG}_indel_pipe.txt 2>/dev/null
python tools/trim_rocprof.py $O/profi_$TAG/p_kernel_stats.csv $O/${TAG}_indel_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmci_${TAG}_$c -o p -- python tools/bench_indel_pipe.py 64444167 2 > /dev/null 2>&1
  python tools/pmc_summary.py $O/pmci_${TAG}_$c/p_counter_collection.csv > $O/${TAG}_pmc_indel_$c.txt
done
ISITES=$(grep -m1 "^rep 0" $O/${TAG}_indel_pipe.txt | sed 's/^rep 0: \([0-9]*\) sites.*/\1/')
python tools/pmc_indel_to_json.py $O/pmci_${TAG}_FETCH_SIZE/p_counter_collection.csv $O/pmci_${TAG}_WRITE_SIZE/p_counter_collection.csv 2 ${ISITES:-40060} $O/${TAG}_indel_traffic.json "profiles/${TAG}_pmc.md: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE over tools/bench_indel_pipe.py 64444167 2" > /dev/null
rm -rf $O/profi_$TAG/p_kernel_trace.csv $O/pmci_${TAG}_FETCH_SIZE/p_kernel_trace.csv $O/pmci_${TAG}_WRITE_SIZE/p_kernel_trace.csv
tail -3 $O/${TAG}_indel_pipe.txt
