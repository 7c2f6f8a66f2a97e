# second SQ pass: where the waves' cycles go (active VALU / LDS / scalar cycles, LDS stalls) and the clock the chip held
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc2}
rm -rf $O; mkdir -p $O
timeout -s KILL 200 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d $O/a -o a -- python3 bench.py --steps 1 --warmup 0 --pairs 32 --no-cpu-baseline ${2:-} > $O/a.log 2>&1
timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace -d $O/kt -o kt -- python3 bench.py --steps 1 --warmup 0 --pairs 32 --no-cpu-baseline ${2:-} > $O/kt.log 2>&1
find $O -name "*.csv"
