#!/bin/bash
# Per-kernel times of the cross operators run alone (nothing beside them on the GPU):  bash tools/cross_prof.sh [stack ld]
set -e
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/prof_cross_${1:-2704}
rm -rf $OUT; mkdir -p $OUT
export CROSS_STACK_LD=${1:-2704}
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/tools/cross_bench.py > $OUT/out.txt
cd $R
cat $OUT/out.txt
python3 - <<PY
import csv, glob
f = glob.glob('$OUT/stats/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'cross' in r['Name']:
        print(r['Name'][:60].ljust(60), r['Calls'], round(float(r['AverageNs']) / 1e3, 1))
PY
