R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3i
mkdir -p $O
cd $R
python3 -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo "pytest rc $?" >> $O/gputests.log
tail -12 $O/gputests.log
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python3 - <<'PY'
import json
b=json.loads(open('gpurun_out/r3i/bench.json').read().strip().splitlines()[-1])
print(b['value'], b['roofline']['frac'])
for k,v in b['extra_configs'].items(): print(k, v.get('seconds'), v.get('achieved'), v.get('frac'), json.dumps(v.get('parity'))[:200])
PY
python3 tools/f32_gemm_bench.py > $O/f32.log 2>&1; cat $O/f32.log
