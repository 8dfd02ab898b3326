for side in 0 1; do for f16 in 0 1; do
  CTR_NO_SIDE=$side CTR_GEMM_F16=$f16 timeout -k 10 200 python bench.py --no-cpu-baseline --no-bf16-leg --no-kernel-timers --no-dcn --steps 200 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('noside=$side f16=$f16 ms=%.4f'%d['ms_per_step'])"
done; done
