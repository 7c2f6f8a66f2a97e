#!/bin/bash
# StereoSGBM replacement at 1080p: parity tests, wall time and per-kernel times
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 python -u -m pytest tests/test_gpu_stereosgbm.py -m gpu -x -q --timeout 200 --timeout-method thread 2>&1 | tail -2 || exit 1
O=gpurun_out/kt_sgbm; rm -rf $O; mkdir -p $O
timeout -k 10 200 python3 scripts/prof_stereosgbm.py 1080p || exit 1
timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt -o kt -- python3 scripts/prof_stereosgbm.py 1080p > $O/kt.log 2>&1 || { tail -5 $O/kt.log; exit 1; }
python3 scripts/kstats.py $O 12
