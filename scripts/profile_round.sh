set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r01h
rm -rf $O; mkdir -p $O
timeout -k 10 240 python3 bench.py > $O/bench.json 2> $O/bench.err
cat $O/bench.json
timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt -o kt -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/kt.log 2>&1
timeout -s KILL 200 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $O/sq -o sq -- python3 bench.py --steps 1 --warmup 0 --pairs 32 --no-cpu-baseline > $O/sq.log 2>&1
# the traffic passes run a small batch with the strip width of the benchmark batch (narrow strips: 256 lanes)
MDISP_SGM_STRIP_LANES=256 timeout -s KILL 200 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/fetch -o f -- python3 bench.py --steps 1 --warmup 0 --pairs 8 --no-cpu-baseline > $O/fetch.log 2>&1
MDISP_SGM_STRIP_LANES=256 timeout -s KILL 200 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/write -o w -- python3 bench.py --steps 1 --warmup 0 --pairs 8 --no-cpu-baseline > $O/write.log 2>&1
timeout -s KILL 200 rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d $O/lds -o l -- python3 bench.py --steps 1 --warmup 0 --pairs 32 --no-cpu-baseline > $O/lds.log 2>&1
find $O -name "*.csv" | head -30
