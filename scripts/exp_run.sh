#!/bin/bash
# Runs the headline benchmark against variant libraries built by scripts/exp_build.sh: scripts/exp_run.sh NAME [NAME ...]
# ("base" = the regular build).  Prints ms per pair per variant.
for n in "$@"; do
  if [ "$n" = base ]; then unset MDISP_LIB_PATH; else export MDISP_LIB_PATH=$PWD/cameracalibration_amd/lib/libmdisp_$n.so; fi
  timeout -k 10 120 python bench.py --no-cpu-baseline --steps 6 --warmup 2 $EXP_BENCH_ARGS > gpurun_out/exp_$n.json 2> gpurun_out/exp_$n.err || { echo "$n failed"; tail -5 gpurun_out/exp_$n.err; exit 1; }
  python - "$n" <<'PY'
import json,sys
n=sys.argv[1]
d=json.loads(open(f"gpurun_out/exp_{n}.json").read().strip().splitlines()[-1])
print(n, "ms_per_pair", round(d["roofline"]["ms_per_pair"],4), "ms_per_step", round(d["ms_per_step"],2))
PY
done
