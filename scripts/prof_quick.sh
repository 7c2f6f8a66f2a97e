# quick look at the SGM batch kernels: per-kernel times and the SQ issue counters (one call each)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-quick}
PAIRS=${2:-64}
rm -rf $O; mkdir -p $O
timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt -o kt -- python3 bench.py --steps 3 --warmup 1 --pairs $PAIRS --no-cpu-baseline > $O/kt.log 2>&1
timeout -s KILL 200 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU -d $O/sq -o sq -- python3 bench.py --steps 1 --warmup 0 --pairs 32 --no-cpu-baseline > $O/sq.log 2>&1
find $O -name "*.csv" | head
