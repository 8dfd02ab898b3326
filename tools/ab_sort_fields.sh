# A/B of the field-segmented pair sort (CTR_SORT_FIELDS) inside the steps on one box: DeepFM c2 and the DCN share
set -e
mkdir -p gpurun_out
for v in 0 1 0 1; do
  CTR_SORT_FIELDS=$v timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --no-bf16-leg > gpurun_out/ab_sf_$v.json 2> gpurun_out/ab_sf_$v.err
  python - <<PY
import json
d=json.load(open('gpurun_out/ab_sf_$v.json'))
print('fields=$v deepfm', round(d['ms_per_step'],4), 'sort', d['kernel_ms_per_step'].get('emb_sort'), 'dcn', round(d['dcn']['ms_per_step'],4), 'sort', d['dcn']['kernel_ms_per_step'].get('emb_sort'), 'path frac', round(d['roofline_emb_path']['frac'],3), round(d['dcn']['roofline_emb_path']['frac'],3))
PY
done
