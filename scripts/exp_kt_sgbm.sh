#!/bin/bash
# Per-kernel times of the StereoSGBM replacement at 1920x1080, 256 disparities: scripts/exp_kt_sgbm.sh NAME [extra bench args]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
n=${1:-base}; shift
if [ "$n" != base ]; then export MDISP_LIB_PATH=$PWD/cameracalibration_amd/lib/libmdisp_$n.so; fi
O=$GRAFT_REPO_ROOT/gpurun_out/kts_$n
rm -rf $O; mkdir -p $O
timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt -o kt -- python3 bench.py --path stereosgbm --steps 10 --warmup 2 --no-cpu-baseline "$@" > $O/kt.log 2>&1 || { tail -5 $O/kt.log; exit 1; }
python3 scripts/kstats.py $O 12
