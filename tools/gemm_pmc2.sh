#!/bin/bash
# Texture-path counter passes over the tower GEMM micro-benchmark:  bash tools/gemm_pmc2.sh <shape names...>
set -e
export TMPDIR=/tmp F16=1
R=$PWD
OUT=$R/gpurun_out/gemm_pmc2
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- python3 $R/tools/gemm_bench.py one "$@" > $OUT/bench_a.txt
timeout -s KILL 120 rocprofv3 --pmc TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum --output-format csv -d $OUT/b -- python3 $R/tools/gemm_bench.py one "$@" > $OUT/bench_b.txt
cd $R
python3 - <<'PY'
import csv, glob, collections
for p in ('a', 'b'):
    f = glob.glob('gpurun_out/gemm_pmc2/%s/*/*counter_collection.csv' % p)
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
            print('   %-40s %14.0f per launch' % (c, v / cnt[(k, c)]))
PY
