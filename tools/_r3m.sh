R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3m
mkdir -p $O
cd $R
python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "potrf or workspace" > $O/t.log 2>&1; echo "pytest rc $?" >> $O/t.log; tail -6 $O/t.log
python3 -m pytest tests/test_gpu_configs.py tests/test_gpu_gp.py -m gpu -q -k "config3 or fp32 or kernelized" > $O/t2.log 2>&1; echo "pytest rc $?" >> $O/t2.log; tail -6 $O/t2.log
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python3 - <<'PY'
import json
b=json.loads(open('gpurun_out/r3m/bench.json').read().strip().splitlines()[-1])
print(b['value'], (b.get('roofline') or {}).get('frac'), b.get('error'))
for k,v in (b.get('extra_configs') or {}).items(): print(k, v.get('seconds'), v.get('achieved'), v.get('frac'), json.dumps(v.get('parity'))[:200])
PY
python3 tools/potrf_sweep.py "16384,32768,65536" "26=64|0" 0 f32 > $O/sweep_f32.log 2>&1; cat $O/sweep_f32.log
