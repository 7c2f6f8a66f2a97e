#!/bin/bash
# Per-kernel times of the headline batch with the two sweeps run one after the other (so their durations add up):
# scripts/exp_kt.sh NAME [extra bench args]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
n=${1:-base}; shift
if [ "$n" != base ]; then export MDISP_LIB_PATH=$PWD/cameracalibration_amd/lib/libmdisp_$n.so; fi
O=$GRAFT_REPO_ROOT/gpurun_out/kt_$n
rm -rf $O; mkdir -p $O
timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt -o kt -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --tune SGM_MEET_MAX=1 "$@" > $O/kt.log 2>&1 || { tail -5 $O/kt.log; exit 1; }
python3 scripts/kstats.py $O 5
