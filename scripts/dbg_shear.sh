#!/bin/bash
# measurement aid: per-kernel times of the sheared sweeps with pieces switched off (results invalid)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for d in ${DBGS:-0 1}; do
  export MDISP_SHEAR_DEBUG=$d
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dbg$d -- python3 $R/bench.py --steps 2 --warmup 1 --pairs ${PAIRS:-32} --no-cpu-baseline > $O/dbg$d.log 2>&1 || exit 1
  echo "debug=$d"; python3 - <<PY
import csv,glob
for r in csv.DictReader(open(sorted(glob.glob('$O/dbg$d/*/*kernel_stats.csv'))[-1])):
    if 'shear' in r['Name'] or 'hscan' in r['Name']: print('   ', r['Name'][30:62], round(float(r['AverageNs'])/1e6,2), 'ms')
PY
done
