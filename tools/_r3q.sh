set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_r03_f32
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o g -- python3 tools/f32_gemm_only.py > $O/kt.log 2>&1 || exit 1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/sq -o g -- python3 tools/f32_gemm_only.py > $O/sq.log 2>&1 || exit 1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/lds -o g -- python3 tools/f32_gemm_only.py > $O/lds.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/mem_f -o g -- python3 tools/f32_gemm_only.py > $O/mem_f.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/mem_w -o g -- python3 tools/f32_gemm_only.py > $O/mem_w.log 2>&1 || exit 1
python3 tools/f32_gemm_pmc.py $O $O/r03_a_f32_gemm_pmc.json
cp $(find $O/kt -name '*kernel_stats.csv' | head -1) $O/r03_a_f32_gemm_kernel_stats.csv
