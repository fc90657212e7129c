#!/bin/bash
# SQ counters of the SNP trunk kernel (experiment): tools/pmc_trunk_sq.sh TAG [lib.so]  -> gpurun_out/TAG_sq.txt
# separate rocprofv3 --pmc passes (no other tracing) over tools/exp_trunk.py --one
TAG=$1; LIB=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
: > $O/${TAG}_sq.txt
for set in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" ${SQ_MORE:+"SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM"}; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/sq_${TAG}_$n -o p -- python tools/exp_trunk.py --one $LIB > /dev/null 2>&1
  python tools/pmc_summary.py $O/sq_${TAG}_$n/p_counter_collection.csv k5_trunk >> $O/${TAG}_sq.txt 2>&1
  rm -rf $O/sq_${TAG}_$n
done
cat $O/${TAG}_sq.txt
