# Every rocprofv3 pass behind the numbers quoted for this round; results under gpurun_out/<tag>/, summaries to copy into profiles/.
# usage (on the GPU box, from the repository root): bash scripts/profile_round.sh r03a
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r03}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
# --- the benchmark itself (SGM 8 paths, 64 pairs of 1920x1080 x 256)
timeout -k 10 300 python3 bench.py > $O/bench.json 2> $O/bench.err
timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt -o kt -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/kt.log 2>&1
timeout -s KILL 200 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $O/sq -o sq -- python3 bench.py --steps 1 --warmup 0 --pairs 64 --no-cpu-baseline --no-verify > $O/sq.log 2>&1
timeout -s KILL 200 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/fetch -o f -- python3 bench.py --steps 1 --warmup 0 --pairs 64 --no-cpu-baseline --no-verify > $O/fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/write -o w -- python3 bench.py --steps 1 --warmup 0 --pairs 64 --no-cpu-baseline --no-verify > $O/write.log 2>&1
timeout -s KILL 200 rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d $O/lds -o l -- python3 bench.py --steps 1 --warmup 0 --pairs 64 --no-cpu-baseline --no-verify > $O/lds.log 2>&1
python3 scripts/sq_summary.py $O/sq/sq_counter_collection.csv 64 $O/sgm_sq_issue.json > $O/sgm_sq_issue.txt
python3 scripts/pmc_summary.py $O/fetch/f_counter_collection.csv $O/write/w_counter_collection.csv 64 $O/sgm_pmc_traffic.json > $O/sgm_pmc_traffic.txt
python3 scripts/lds_summary.py $O/lds/l_counter_collection.csv $O/sgm_lds.json > $O/sgm_lds.txt
# side measurements of the SGM entry point: 16-bit path states (P2 = 300), five paths, the sheared-strip schedule
for v in "--P2 300 --pairs 16" "--paths 5" "--tune SGM_PIPELINE=1"; do
  timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline $v >> $O/bench_side.jsonl 2>> $O/bench.err
done
# --- the other paths: one target script per family, three runs each (kernel stats, fetched bytes, written bytes)
for T in other_paths stereosgbm_1080p main_pipeline guidedf; do
  case $T in
    guidedf) CMD="scripts/prof_guidedf.py 1";;
    other_paths) CMD="scripts/other_paths_rate.py";;
    stereosgbm_1080p) CMD="scripts/prof_stereosgbm.py 1080p";;
    main_pipeline) CMD="scripts/main_pipeline_rate.py";;
  esac
  timeout -k 10 280 rocprofv3 --output-format csv --kernel-trace --stats -d $O/$T/kt -o kt -- python3 $CMD > $O/$T.log 2>&1
  timeout -s KILL 280 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/$T/fetch -o f -- python3 $CMD > $O/$T.fetch.log 2>&1
  timeout -s KILL 280 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/$T/write -o w -- python3 $CMD > $O/$T.write.log 2>&1
  python3 scripts/traffic_by_kernel.py $O/$T/kt/kt_kernel_stats.csv $O/$T/fetch/f_counter_collection.csv $O/$T/write/w_counter_collection.csv $O/${T}_kernels.json > $O/${T}_kernels.txt
done
cp $O/kt/kt_kernel_stats.csv $O/sgm_kernel_stats.csv
ls $O
