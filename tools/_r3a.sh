set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3a
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
python3 -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "pytest rc $?" >> $O/gputests.log
tail -3 $O/gputests.log
STPY_HIP_LIB=lab python3 -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "trsv or strip or beside" > $O/gputests_lab.log 2>&1; echo "pytest rc $?" >> $O/gputests_lab.log
tail -3 $O/gputests_lab.log
rocprofv3 --kernel-trace --output-format csv -d $O/kt16 -o p -- python3 tools/potrf_only.py 16384 > $O/kt16.log 2>&1 || exit 1
f=$(ls $O/kt16/*/*kernel_trace.csv 2>/dev/null | head -1); [ -z "$f" ] && f=$(find $O/kt16 -name '*kernel_trace.csv' | head -1)
python3 tools/trace_union.py $f --steps 3 --flops-per-step 1.466e12 > $O/kt16_union.txt 2>&1
python3 - "$f" > $O/kt16_timeline.txt <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last potrf: take the last third of rows
t0=int(rows[0]['Start_Timestamp'])
n=len(rows)
sel=rows[2*n//3:]
for r in sel:
    nm=re.search(r'stpy::(\w+)',r['Kernel_Name']); nm=nm.group(1) if nm else r['Kernel_Name'][:30]
    s=int(r['Start_Timestamp'])-t0; e=int(r['End_Timestamp'])-t0
    print(r.get('Queue_Id','?'), nm, s/1e3, (e-s)/1e3, r.get('Grid_Size','?'), r.get('Workgroup_Size','?'))
PY
STPY_HIP_LIB=lab python3 tools/potrf_sweep.py "8192,16384,32768" "0=40000" > $O/potrf_sweep.log 2>&1
cat $O/potrf_sweep.log
