#!/bin/bash
# headline rate under a list of tuning settings: scripts/tune_sweep.sh "SGM_BAND_SLACK=64" "SGM_PAIR_GROUP=16 SGM_BAND_SLACK=64" ...
mkdir -p gpurun_out
: > gpurun_out/tune_sweep.txt
for cfg in "" "$@"; do
  args=""
  for kv in $cfg; do args="$args --tune $kv"; done
  out=$(timeout -k 10 300 python bench.py --pairs ${PAIRS:-64} --steps ${STEPS:-4} --warmup 1 --no-cpu-baseline $args 2>/dev/null) || { echo "FAILED: $cfg" >> gpurun_out/tune_sweep.txt; exit 1; }
  echo "$out" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-50s %8.4f ms/pair  %8.0f Mdisp/s' % ('$cfg', d['roofline']['ms_per_pair'], d['value']))" >> gpurun_out/tune_sweep.txt
done
cat gpurun_out/tune_sweep.txt
