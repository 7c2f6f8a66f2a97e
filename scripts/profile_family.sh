# One family of scripts/profile_round.sh alone (kernel statistics, fetched and written bytes per kernel):
# bash scripts/profile_family.sh TAG other_paths|stereosgbm_1080p|main_pipeline
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r02}; T=$2
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $O; rm -rf $O/$T
case $T in
  other_paths) CMD="scripts/other_paths_rate.py";;
  stereosgbm_1080p) CMD="scripts/prof_stereosgbm.py 1080p";;
  main_pipeline) CMD="scripts/main_pipeline_rate.py";;
  *) echo "unknown family $T"; exit 2;;
esac
timeout -k 10 280 rocprofv3 --output-format csv --kernel-trace --stats -d $O/$T/kt -o kt -- python3 $CMD > $O/$T.log 2>&1
timeout -s KILL 280 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/$T/fetch -o f -- python3 $CMD > $O/$T.fetch.log 2>&1
timeout -s KILL 280 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/$T/write -o w -- python3 $CMD > $O/$T.write.log 2>&1
python3 scripts/traffic_by_kernel.py $O/$T/kt/kt_kernel_stats.csv $O/$T/fetch/f_counter_collection.csv $O/$T/write/w_counter_collection.csv $O/${T}_kernels.json > $O/${T}_kernels.txt
cat $O/${T}_kernels.txt
