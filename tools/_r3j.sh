R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3j
mkdir -p $O
cd $R
python3 -m pytest tests/test_gpu_block_cyclic.py -m gpu -q -k "self_launch" > $O/t.log 2>&1; echo "pytest rc $?" >> $O/t.log; tail -5 $O/t.log
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python3 - <<'PY'
import json
b=json.loads(open('gpurun_out/r3j/bench.json').read().strip().splitlines()[-1])
print(b['value'], b['roofline']['frac'], b.get('error'))
for k,v in (b.get('extra_configs') or {}).items(): print(k, v.get('seconds'), v.get('achieved'), v.get('frac'), json.dumps(v.get('parity'))[:200])
PY
