set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3d
mkdir -p $O
cd $R
python3 -m pytest tests -m gpu -q --durations=8 > $O/gputests.log 2>&1; echo "pytest rc $?" >> $O/gputests.log
tail -22 $O/gputests.log
python3 tools/hbm_bench.py predict > $O/hbm.log 2>&1; cat $O/hbm.log
STPY_HIP_LIB=lab python3 tools/potrf_sweep.py "65536" "23=1073741824|16384|24576|32768" > $O/sweep_nb2048.log 2>&1
cat $O/sweep_nb2048.log
