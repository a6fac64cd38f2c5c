R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3l
mkdir -p $O
cd $R
timeout -k 10 200 python3 tools/_dbg_kf.py > $O/dbg.log 2>&1; cat $O/dbg.log
python3 -m pytest tests/test_gpu_kernels.py -m gpu -q -k "bf16_matrix" > $O/t.log 2>&1; echo "pytest rc $?" >> $O/t.log; tail -5 $O/t.log
python3 -m pytest tests/test_gpu_configs.py tests/test_gpu_gp.py -m gpu -q -k "config3 or fp32 or kernelized" > $O/t2.log 2>&1; echo "pytest rc $?" >> $O/t2.log; tail -5 $O/t2.log
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python3 - <<'PY'
import json
b=json.loads(open('gpurun_out/r3l/bench.json').read().strip().splitlines()[-1])
print(b['value'], (b.get('roofline') or {}).get('frac'), b.get('error'))
for k,v in (b.get('extra_configs') or {}).items(): print(k, v.get('seconds'), v.get('achieved'), v.get('frac'), json.dumps(v.get('parity'))[:200])
PY
python3 tools/f32_gemm_bench.py > $O/f32.log 2>&1; cat $O/f32.log
