#!/bin/bash
# One bench.py line per model family at its BASELINE shape, single GPU; output in gpurun_out/family_sweep.jsonl
set -e
mkdir -p gpurun_out
: > gpurun_out/family_sweep.jsonl
for m in DeepFM NFM ESMM PNN PNN_outer AFM AFM_pairs DCN; do
  timeout -k 10 150 python bench.py --model $m --steps 20 --warmup 3 --no-cpu-baseline >> gpurun_out/family_sweep.jsonl
  echo "done $m"
done
