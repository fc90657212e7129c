#!/bin/bash
# profiles of the device ingest (round 4c): kernel stats of the from-BAM experiment (12 contigs) and of the inflate experiment (7 x 3 Mb = 14.5 k members),
# FETCH_SIZE / WRITE_SIZE and SQ counters of the inflate kernels
set -x
TAG=${1:-r04c}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/profb_$TAG -o p -- python tools/exp_from_bam.py 12 > $O/${TAG}_from_bam.txt 2>/dev/null
python tools/trim_rocprof.py $O/profb_$TAG/p_kernel_stats.csv $O/${TAG}_ingest_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/profi_$TAG -o p -- python tools/exp_inflate.py 7 3000000 > $O/${TAG}_inflate.txt 2>/dev/null
python tools/trim_rocprof.py $O/profi_$TAG/p_kernel_stats.csv $O/${TAG}_inflate_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --kernel-include-regex "k_huff|k_lz" --output-format csv -d $O/pmcz_${TAG}_$c -o p -- python tools/exp_inflate.py 7 3000000 > /dev/null 2>&1
  python tools/pmc_summary.py $O/pmcz_${TAG}_$c/p_counter_collection.csv > $O/${TAG}_pmc_inflate_$c.txt
done
bash tools/pmc_inflate.sh 7 > $O/${TAG}_sq_inflate.txt 2>/dev/null
rm -rf $O/profb_$TAG/p_kernel_trace.csv $O/profi_$TAG/p_kernel_trace.csv $O/pmcz_${TAG}_*/p_kernel_trace.csv
tail -4 $O/${TAG}_inflate.txt; cat $O/${TAG}_pmc_inflate_FETCH_SIZE.txt $O/${TAG}_pmc_inflate_WRITE_SIZE.txt | head -20
