# A/B of the pair sort issued beside the forward pass (CTR_EARLY_SORT) on one box: DeepFM c2 and the DCN share
set -e
mkdir -p gpurun_out
for v in 0 1 0 1; do
  CTR_EARLY_SORT=$v timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --no-bf16-leg > gpurun_out/ab_es_$v.json 2> gpurun_out/ab_es_$v.err
  python - <<PY
import json
d=json.load(open('gpurun_out/ab_es_$v.json'))
print('early_sort=$v deepfm', round(d['ms_per_step'],4), 'dcn', round(d['dcn']['ms_per_step'],4), 'dcn eager', d['dcn'].get('eager',{}).get('timing',{}).get('ms_per_step_median'))
PY
done
