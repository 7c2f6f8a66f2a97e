#!/bin/bash
# ms per pair of the headline path as a function of the batch size (fixed cost per call vs per-pair cost)
set -e
mkdir -p gpurun_out
: > gpurun_out/batch_sweep.jsonl
for B in ${BATCHES:-16 32 48 64 96 128 192 256}; do
  timeout -k 10 300 python bench.py --pairs $B --steps 3 --warmup 1 --no-cpu-baseline "$@" >> gpurun_out/batch_sweep.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/batch_sweep.jsonl'):
    d = json.loads(l)
    r = d['roofline']
    print(d['config']['pairs_per_gpu_per_step'], round(d['ms_per_step'], 2), 'ms/step', round(r['ms_per_pair'], 4), 'ms/pair (events)', round(d['ms_per_step'] / d['config']['pairs_per_gpu_per_step'], 4), 'ms/pair (wall)')
PY
