R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3h
mkdir -p $O
cd $R
timeout -k 10 300 python3 -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "bf16_matrix or gemm_nt" > $O/t.log 2>&1; echo "pytest rc $?" >> $O/t.log; tail -15 $O/t.log
timeout -k 10 300 python3 tools/f32_gemm_bench.py > $O/bench.log 2>&1; cat $O/bench.log
