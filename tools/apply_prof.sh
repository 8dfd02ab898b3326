#!/bin/bash
# Per-kernel times of the table backward run alone with cold caches:  bash tools/apply_prof.sh
set -e
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/prof_apply
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/tools/apply_bench.py > $OUT/out.txt 2>&1
cd $R
python3 - <<PY
import csv, glob
f = glob.glob('$OUT/stats/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'ctr::' in r['Name']:
        print(r['Name'][:70].ljust(70), r['Calls'], round(float(r['AverageNs']) / 1e3, 1))
PY
