#!/bin/bash
# SQ counters of the two inflate kernels (experiment helper): tools/pmc_inflate.sh [contigs]
cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/pmc_infl
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-include-regex "k_huff|k_lz" -d /root/repo/gpurun_out/pmc_infl -o a --output-format csv -- python /root/repo/tools/exp_inflate.py ${1:-8} 3000000 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-include-regex "k_huff|k_lz" -d /root/repo/gpurun_out/pmc_infl -o b --output-format csv -- python /root/repo/tools/exp_inflate.py ${1:-8} 3000000 > /dev/null 2>&1
python - <<'P'
import csv,glob,collections
for f in sorted(glob.glob('/root/repo/gpurun_out/pmc_infl/*counter_collection.csv')):
    acc=collections.defaultdict(float); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=(r['Kernel_Name'].split('(')[1][-30:] if False else ('huff' if 'k_huff' in r['Kernel_Name'] else 'lz'), r['Counter_Name'])
        acc[k]+=float(r['Counter_Value']); n[k]+=1
    for k in sorted(acc): print(k[0], k[1], "%.4g" % (acc[k]/n[k]))
P
