# A/B of the FM row term in the table update (CTR_FM_Z_IN_UPDATE) on one box: DeepFM c2
set -e
mkdir -p gpurun_out
for v in 0 1 0 1; do
  CTR_FM_Z_IN_UPDATE=$v timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-legs --no-dcn --no-cpu-baseline --no-bf16-leg > gpurun_out/ab_fz_$v.json 2> gpurun_out/ab_fz_$v.err
  python - <<PY
import json
d=json.load(open('gpurun_out/ab_fz_$v.json'))
k=d['kernel_ms_per_step']
print('late_z=$v deepfm', round(d['ms_per_step'],4), 'dx0', k.get('gemm_65536x429x256_NT'), 'apply', k.get('emb_sparse_apply'), 'roofline', d['roofline']['kernel'][:30], round(d['roofline']['frac'],3), 'loss', d['final_loss'])
PY
done
