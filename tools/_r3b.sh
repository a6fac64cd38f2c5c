set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3b
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
python3 -m pytest tests/test_gpu_gp.py -m gpu -x -q -k "gradient" > $O/grad.log 2>&1; echo "pytest rc $?" >> $O/grad.log; tail -3 $O/grad.log
STPY_HIP_LIB=lab python3 -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "trsv" > $O/lab.log 2>&1; echo "pytest rc $?" >> $O/lab.log; tail -3 $O/lab.log
export STPY_HIP_LIB=lab
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o p -- python3 tools/potrf_only.py 16384 0 12=16384 18=3 > $O/kt.log 2>&1 || exit 1
f=$(find $O/kt -name '*kernel_trace.csv' | head -1)
python3 - "$f" > $O/timeline.txt <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
n=len(rows)
for r in rows[2*n//3:]:
    nm=re.search(r'stpy::(\w+)',r['Kernel_Name']); nm=nm.group(1) if nm else r['Kernel_Name'][:30]
    s=int(r['Start_Timestamp'])-t0; e=int(r['End_Timestamp'])-t0
    print(r.get('Queue_Id','?'), nm, s/1e3, (e-s)/1e3, r.get('Grid_Size','?'), r.get('Workgroup_Size','?'))
PY
python3 tools/potrf_sweep.py "8192,16384,32768" "12=0|16384|32768;18=0|3" > $O/sweep.log 2>&1
cat $O/sweep.log
