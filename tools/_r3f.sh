R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3f
mkdir -p $O
cd $R
hipcc --offload-arch=gfx950 -O2 tools/waitvalue_probe.hip -o /tmp/wv && timeout -k 5 60 /tmp/wv > $O/wv.log 2>&1; echo "probe rc $?" >> $O/wv.log; cat $O/wv.log
export STPY_HIP_LIB=lab
python3 tools/potrf_sweep.py "65536" "24=0|512;25=1073741824|49152" > $O/sweep.log 2>&1
cat $O/sweep.log
python3 tools/dist_single_rank.py 65536 1024 > $O/dsr_1024.log 2>&1; cat $O/dsr_1024.log
python3 tools/dist_single_rank.py 65536 2048 > $O/dsr_2048.log 2>&1; cat $O/dsr_2048.log
