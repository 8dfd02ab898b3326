# what bounds the tower products: the same shapes with the A stream served by the caches / without write traffic
set -e
for v in "" "CTR_GEMM_DEBUG_A0=1" "CTR_GEMM_DEBUG_C0=1" "CTR_GEMM_DEBUG_A0=1 CTR_GEMM_DEBUG_C0=1"; do
  echo "== $v"
  env F16=1 $v timeout -k 10 100 python3 tools/gemm_bench.py one fwd0 fwd1 dx0 dx1 sq4k | cut -c1-60
done
