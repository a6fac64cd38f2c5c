R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3e
mkdir -p $O
cd $R
export STPY_HIP_LIB=lab
python3 tools/potrf_sweep.py "65536" "24=0|256|512|1024;25=1073741824|49152" > $O/sweep.log 2>&1
cat $O/sweep.log
python3 tools/potrf_sweep.py "16384,32768" "24=0|256" > $O/sweep2.log 2>&1
cat $O/sweep2.log
