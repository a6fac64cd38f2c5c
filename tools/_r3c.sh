set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c
mkdir -p $O
cd $R
python3 -m pytest tests -m gpu -x -q --durations=12 > $O/gputests.log 2>&1; echo "pytest rc $?" >> $O/gputests.log
tail -25 $O/gputests.log
python3 bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
tail -c 3000 $O/bench.json
STPY_HIP_LIB=lab python3 tools/potrf_sweep.py "65536" "23=1073741824|32768|49152" > $O/sweep_nb2048.log 2>&1
cat $O/sweep_nb2048.log
