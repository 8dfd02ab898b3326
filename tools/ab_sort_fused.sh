# A/B inside the steps on one box: passes of the field sort as one launch each (CTR_FS_FUSED_MAX_TF=32) against a
# histogram launch per pass (0); DeepFM c2 and the DCN share, then the sort alone
set -e
mkdir -p gpurun_out
for v in 0 32 0 32; do
  CTR_FS_FUSED_MAX_TF=$v timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --no-bf16-leg > gpurun_out/ab_fu_$v.json 2> gpurun_out/ab_fu_$v.err
  python - <<PY
import json
d=json.load(open('gpurun_out/ab_fu_$v.json'))
print('fused_max_tf=$v deepfm', round(d['ms_per_step'],4), 'sort', d['kernel_ms_per_step'].get('emb_sort'), 'dcn', round(d['dcn']['ms_per_step'],4), 'sort', d['dcn']['kernel_ms_per_step'].get('emb_sort'), 'path frac', round(d['roofline_emb_path']['frac'],3), round(d['dcn']['roofline_emb_path']['frac'],3))
PY
done
for v in 0 4 32; do echo "alone fused_max_tf=$v"; CTR_FS_FUSED_MAX_TF=$v timeout -k 10 100 python tools/sort_bench.py | grep -v amdgpu; done
