# The counter passes of scripts/profile_round.sh alone (SQ issue, fetched / written bytes, LDS): bash scripts/profile_counters.sh TAG
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r02}; mkdir -p $O
rm -rf $O/sq $O/fetch $O/write $O/lds
timeout -s KILL 200 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $O/sq -o sq -- python3 bench.py --steps 1 --warmup 0 --pairs 32 --no-cpu-baseline --no-verify > $O/sq.log 2>&1
timeout -s KILL 200 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/fetch -o f -- python3 bench.py --steps 1 --warmup 0 --pairs 16 --no-cpu-baseline --no-verify > $O/fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/write -o w -- python3 bench.py --steps 1 --warmup 0 --pairs 16 --no-cpu-baseline --no-verify > $O/write.log 2>&1
timeout -s KILL 200 rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d $O/lds -o l -- python3 bench.py --steps 1 --warmup 0 --pairs 32 --no-cpu-baseline --no-verify > $O/lds.log 2>&1
python3 scripts/sq_summary.py $O/sq/sq_counter_collection.csv 32 $O/sgm_sq_issue.json > $O/sgm_sq_issue.txt
python3 scripts/pmc_summary.py $O/fetch/f_counter_collection.csv $O/write/w_counter_collection.csv 16 $O/sgm_pmc_traffic.json > $O/sgm_pmc_traffic.txt
python3 scripts/lds_summary.py $O/lds/l_counter_collection.csv $O/sgm_lds.json > $O/sgm_lds.txt
for f in sgm_sq_issue sgm_pmc_traffic sgm_lds; do tail -n 3 $O/$f.txt; done
