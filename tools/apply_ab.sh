# Per-kernel times of the table backward alone (cold caches) with Adam's slots separate and side by side
set -e
export TMPDIR=/tmp
R=$PWD
for mv in 0 1; do
OUT=$R/gpurun_out/prof_apply_mv$mv
rm -rf $OUT; mkdir -p $OUT
cd /tmp
APPLY_MV=$mv timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/tools/apply_bench.py > $OUT/out.txt 2>&1
cd $R
echo "APPLY_MV=$mv"
python3 - <<PY
import csv, glob
f = glob.glob('$OUT/stats/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'ctr::' in r['Name']:
        print(r['Name'][:70].ljust(70), r['Calls'], round(float(r['AverageNs']) / 1e3, 1))
PY
done
