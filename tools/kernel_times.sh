#!/bin/bash
# Per-kernel times of a short bench.py run:  bash tools/kernel_times.sh [DeepFM|DCN] [pattern]
set -e
export TMPDIR=/tmp
R=$PWD
MODEL=${1:-DeepFM}
OUT=$R/gpurun_out/kt_$MODEL
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --model $MODEL --steps 20 --warmup 3 --no-cpu-baseline --no-dcn --no-bf16-leg > $OUT/bench.json 2>/dev/null
cd $R
python3 - <<PY
import csv, glob
f = glob.glob('$OUT/stats/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:45]:
    if '${2:-}' in r['Name']:
        print(r['Name'][:90].ljust(90), r['Calls'], round(float(r['AverageNs']) / 1e3, 1), r['Percentage'])
PY
