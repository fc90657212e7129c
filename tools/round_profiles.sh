#!/bin/bash
# The profile set of a round, on the GPU box (one gpurun call): tools/round_profiles.sh TAG
#   gpurun_out/TAG_bench.json                default `python bench.py`
#   gpurun_out/TAG_kernel_stats.csv          timeout 600 rocprofv3 --kernel-trace --stats over bench.py (trimmed) + TAG_bench_under_rocprof.json
#   gpurun_out/TAG_indel_kernel_stats.csv    the same over tools/bench_indel_pipe.py (the indel pipeline's kernels)
#   gpurun_out/TAG_pmc_*                     FETCH_SIZE / WRITE_SIZE passes (separate runs, no other tracing): trunk kernel -> trunk_traffic.json;
#                                            indel kernels -> per-kernel sums
set -x
TAG=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout 1500 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err < /dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o p -- python bench.py --no-extra --no-configs2 --no-cpu-baseline --no-wgs --steps 40 --warmup 5 --repeat 1 > $O/${TAG}_bench_under_rocprof.json 2>/dev/null
python tools/trim_rocprof.py $O/prof_$TAG/p_kernel_stats.csv $O/${TAG}_kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/profi_$TAG -o p -- python tools/bench_indel_pipe.py 64444167 4 > $O/${TAG}_indel_pipe.txt 2>/dev/null
python tools/trim_rocprof.py $O/profi_$TAG/p_kernel_stats.csv $O/${TAG}_indel_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${TAG}_$c -o p -- python bench.py --no-extra --no-configs2 --no-cpu-baseline --no-wgs --steps 4 --warmup 1 --repeat 1 > /dev/null 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmci_${TAG}_$c -o p -- python tools/bench_indel_pipe.py 64444167 2 > /dev/null 2>&1
  python tools/pmc_summary.py $O/pmci_${TAG}_$c/p_counter_collection.csv > $O/${TAG}_pmc_indel_$c.txt
  python tools/pmc_summary.py $O/pmc_${TAG}_$c/p_counter_collection.csv > $O/${TAG}_pmc_snp_$c.txt
done
# bytes per site of the dominant kernel / per stage of the indel pipeline, tagged with the sha-256 of the kernel sources they were measured on
# (bench.py reports `traffic` only from passes of the sources it runs): copy to profiles/trunk_traffic.json and profiles/indel_traffic.json
python tools/pmc_to_json.py $O/pmc_${TAG}_FETCH_SIZE/p_counter_collection.csv $O/pmc_${TAG}_WRITE_SIZE/p_counter_collection.csv k5_trunk_lin 624622 $O/${TAG}_trunk_traffic.json "profiles/${TAG}_pmc.md: timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE over bench.py --no-extra --no-configs2 --no-cpu-baseline --no-wgs --steps 4 --warmup 1 --repeat 1" > /dev/null
ISITES=$(grep -m1 "^rep 0" $O/${TAG}_indel_pipe.txt | sed 's/^rep 0: \([0-9]*\) sites.*/\1/')
python tools/pmc_indel_to_json.py $O/pmci_${TAG}_FETCH_SIZE/p_counter_collection.csv $O/pmci_${TAG}_WRITE_SIZE/p_counter_collection.csv 2 ${ISITES:-40060} $O/${TAG}_indel_traffic.json "profiles/${TAG}_pmc.md: timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE over tools/bench_indel_pipe.py 64444167 2" > /dev/null
SITES=$(python -c "import json;b=json.loads(open('$O/${TAG}_bench.json').read().strip().splitlines()[-1]);print(b['config'].get('sites_per_contig') or b['config'].get('sites_per_step') or 0)")
echo sites per contig: $SITES
ls $O/pmc_${TAG}_FETCH_SIZE/
rm -rf $O/prof_$TAG/p_kernel_trace.csv $O/profi_$TAG/p_kernel_trace.csv
