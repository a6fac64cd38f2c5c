R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3r
mkdir -p $O
cd $R
python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_gp.py -m gpu -q -x -k "gram or K1 or K2 or G1 or G2 or G3 or G4 or mid_size" > $O/t.log 2>&1; echo "pytest rc $?" >> $O/t.log; tail -6 $O/t.log
python3 tools/hbm_bench.py gram > $O/gram.log 2>&1; cat $O/gram.log
