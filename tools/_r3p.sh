R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3p
mkdir -p $O
cd $R
SECONDS=0
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $? in $SECONDS s"
SECONDS=0
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs > $O/bench2.json 2> $O/bench2.err; echo "bench (no extras) rc $? in $SECONDS s"
SECONDS=0
python3 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench3.json 2> $O/bench3.err; echo "bench (extras, 3 steps, no cpu baseline) rc $? in $SECONDS s"
python3 -m pytest tests/test_gpu_gp.py -m gpu -q -k "kernelized" > $O/t.log 2>&1; tail -2 $O/t.log
