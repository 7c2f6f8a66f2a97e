#!/bin/bash
# StereoSGBM replacement (1920x1080, 256 disparities) against variant libraries built by scripts/exp_build.sh:
# scripts/exp_sgbm.sh NAME [NAME ...]   ("base" = the regular build).  Prints ms per pair, MODE_SGBM and MODE_HH.
for n in "$@"; do
  if [ "$n" = base ]; then unset MDISP_LIB_PATH; else export MDISP_LIB_PATH=$PWD/cameracalibration_amd/lib/libmdisp_$n.so; fi
  for m in "" "--hh"; do
    timeout -k 10 120 python bench.py --path stereosgbm $m --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/exp_sgbm_$n$m.json 2> gpurun_out/exp_sgbm_$n.err || { echo "$n failed"; tail -5 gpurun_out/exp_sgbm_$n.err; exit 1; }
    python - "$n$m" gpurun_out/exp_sgbm_$n$m.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], "ms_per_step", round(d["ms_per_step"],3))
PY
  done
done
