#!/bin/bash
# SQ counter pass over the tower GEMM micro-benchmark:  bash tools/gemm_pmc.sh <shape names...>
set -e
export TMPDIR=/tmp F16=1
R=$PWD
OUT=$R/gpurun_out/gemm_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $OUT/a -- python3 $R/tools/gemm_bench.py one "$@" > $OUT/bench_a.txt
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/b -- python3 $R/tools/gemm_bench.py one "$@" > $OUT/bench_b.txt
cd $R
python3 - <<'PY'
import csv, glob, collections
for p in ('a', 'b'):
    f = glob.glob('gpurun_out/gemm_pmc/%s/*/*counter_collection.csv' % p)
    if not f: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name'][:60] + ' grid=' + r['Grid_Size']
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[(k, r['Counter_Name'])] += 1
    for k, d in acc.items():
        if 'gemm' not in k: continue
        print(k)
        for c, v in d.items():
            print('   %-28s %14.0f per launch' % (c, v / cnt[(k, c)]))
PY
