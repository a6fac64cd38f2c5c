R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3o
mkdir -p $O
cd $R
python3 -m pytest tests -m gpu -q > $O/gputests.log 2>&1; echo "pytest rc $?" >> $O/gputests.log
tail -8 $O/gputests.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -3 $O/smoke.log
SECONDS=0
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $? in $SECONDS s"
python3 - <<'PY'
import json
b=json.loads(open('gpurun_out/r3o/bench.json').read().strip().splitlines()[-1])
print(b['value'], b['roofline']['frac'], b['roofline']['traffic'], b['roofline'].get('traffic_note','')[:80])
for k,v in (b.get('extra_configs') or {}).items(): print(k, v.get('seconds'), v.get('achieved'), v.get('frac'))
print(b['cpu_baseline']['value'], b['cpu_baseline']['parity_rel_err'])
PY
