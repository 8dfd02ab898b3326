# tower products against the number of resident blocks per CU (extra dynamic LDS per block)
set -e
for pad in 0 24000 50000 100000; do
  echo "CTR_GEMM_LDS_PAD=$pad"
  CTR_GEMM_LDS_PAD=$pad F16=1 timeout -k 10 200 python tools/gemm_bench.py 2>&1 | grep -E "fwd0|fwd1|dx0|dx1|dw0|dw1|sq4k"
done
echo DCN
for pad in 0 50000; do
  echo "CTR_GEMM_LDS_PAD=$pad"
  CTR_GEMM_LDS_PAD=$pad F16=1 timeout -k 10 200 python tools/gemm_bench.py dcn 2>&1 | grep -E "fwd|dx|dw"
done
