# per-kernel durations of ctr_emb_sort on the benchmark shapes (rocprofv3 kernel trace of tools/sort_bench.py)
set -e
export TMPDIR=/tmp
R=$PWD
for dbg in ${DBGS:-0}; do
OUT=$R/gpurun_out/sort_prof_$dbg
rm -rf $OUT; mkdir -p $OUT
cd /tmp
CTR_FS_DEBUG=$dbg timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/sort_bench.py > $OUT/bench.txt 2>&1 || true
done
