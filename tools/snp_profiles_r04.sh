set -x
TAG=r04a
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o p -- python bench.py --no-extra --no-configs2 --no-cpu-baseline --steps 40 --warmup 5 --repeat 1 > $O/${TAG}_bench_under_rocprof.json 2>/dev/null
python tools/trim_rocprof.py $O/prof_$TAG/p_kernel_stats.csv $O/${TAG}_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${TAG}_$c -o p -- python bench.py --no-extra --no-configs2 --no-cpu-baseline --steps 4 --warmup 1 --repeat 1 > /dev/null 2>&1
  python tools/pmc_summary.py $O/pmc_${TAG}_$c/p_counter_collection.csv > $O/${TAG}_pmc_snp_$c.txt
done
python tools/pmc_to_json.py $O/pmc_${TAG}_FETCH_SIZE/p_counter_collection.csv $O/pmc_${TAG}_WRITE_SIZE/p_counter_collection.csv k5_trunk_h3 624622 $O/${TAG}_trunk_traffic.json "profiles/${TAG}_pmc.md: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE over bench.py --no-extra --no-configs2 --no-cpu-baseline --steps 4 --warmup 1 --repeat 1"
rm -rf $O/prof_$TAG/p_kernel_trace.csv
head -12 $O/${TAG}_kernel_stats.csv
