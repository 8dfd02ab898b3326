#!/bin/bash
# rocprofv3 passes over a bench.py run on one GPU:  bash tools/profile_bench.sh [DeepFM|DCN]
#   1. kernel trace + stats  -> gpurun_out/prof_<model>/stats/…_kernel_stats.csv
#   2. FETCH_SIZE counters   -> gpurun_out/prof_<model>/fetch/…_counter_collection.csv
#   3. WRITE_SIZE counters   -> gpurun_out/prof_<model>/write/…_counter_collection.csv
# then folds 2 and 3 into gpurun_out/prof_<model>/pmc_traffic.json.  Copy the summaries into profiles/ afterwards
# (profiles/rNN_<model lower case>_..._pmc_traffic.json is what bench.py reads for the `traffic` fields).
set -e
export TMPDIR=/tmp
MODEL=${1:-DeepFM}
R=$PWD
OUT=$R/gpurun_out/prof_$MODEL
rm -rf $OUT; mkdir -p $OUT
cd /tmp
ARGS="--model $MODEL --steps 30 --warmup 3 --repeats 1 --no-cpu-baseline --no-dcn --no-legs"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py $ARGS > $OUT/bench_under_rocprof.json
echo "stats pass done"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py --model $MODEL --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-dcn --no-legs --no-kernel-timers > /dev/null
echo "fetch pass done"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py --model $MODEL --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-dcn --no-legs --no-kernel-timers > /dev/null
echo "write pass done"
cd $R
python3 tools/pmc_summary.py $(find $OUT/fetch -name "*counter_collection.csv" | head -1) $(find $OUT/write -name "*counter_collection.csv" | head -1) $OUT/pmc_traffic.json
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
