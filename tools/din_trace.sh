set -e
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/din_trace
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/tools/din_bench.py --H 32 --steps 10 --graph 1 > $OUT/bench.txt
