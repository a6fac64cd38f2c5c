#!/bin/bash
# The round's profiling passes on the GPU box (run from the repo root through gpurun); raw output under gpurun_out/prof_r04/,
# the summaries that DESIGN.md / bench lines cite are copied into profiles/ afterwards (tools/trace_union.py, tools/pmc_traffic.py).
# rocprofv3 is given the program itself after `--`; counters are collected in their own passes (no tracing options with --pmc).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
echo "== bench line (not profiled)"; python3 bench.py --steps 5 --warmup 2 > $O/bench_line.json 2> $O/bench_line.err || exit 1
tail -c 600 $O/bench_line.json; echo
echo "== kernel trace + stats"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs > $O/kt_line.json 2> $O/kt.err || exit 1
echo "== pmc FETCH_SIZE"; rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o bench -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra-configs > $O/pmc_fetch.json 2> $O/pmc_fetch.err || exit 1
echo "== pmc WRITE_SIZE"; rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o bench -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra-configs > $O/pmc_write.json 2> $O/pmc_write.err || exit 1
echo "== gram / rff: kernel stats, then counters"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/gr_kt -o gr -- python3 tools/gram_rff_only.py > $O/gr_kt.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/gr_write -o gr -- python3 tools/gram_rff_only.py > $O/gr_write.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/gr_fetch -o gr -- python3 tools/gram_rff_only.py > $O/gr_fetch.log 2>&1 || exit 1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/gr_sq -o gr -- python3 tools/gram_rff_only.py > $O/gr_sq.log 2>&1 || exit 1
# summaries for profiles/ (copied into the tracked directory by hand after the call)
TAG=${1:-r04_g}
S=$O/summaries; mkdir -p $S
f=$(find $O/kt -name '*kernel_trace.csv' | head -1)
python3 tools/trace_union.py $f --steps 4 --json $S/${TAG}_bench_n65536_union.json > $S/${TAG}_union.txt 2>&1
cp $(find $O/kt -name '*kernel_stats.csv' | head -1) $S/${TAG}_bench_n65536_kernel_stats.csv
cp $O/bench_line.json $S/${TAG}_bench_n65536_line.json
cp $O/kt_line.json $S/${TAG}_bench_n65536_line_under_profiler.json
python3 tools/pmc_traffic.py $(find $O/pmc_fetch -name '*counter_collection.csv' | head -1) $(find $O/pmc_write -name '*counter_collection.csv' | head -1) $S/${TAG}_bench_n65536_pmc_traffic.json
cp $(find $O/gr_kt -name '*kernel_stats.csv' | head -1) $S/${TAG}_gram_rff_kernel_stats.csv
python3 tools/gram_rff_pmc.py $O $S/${TAG}_gram_rff_pmc.json > $S/${TAG}_gram_rff_pmc.txt 2>&1 || true
ls -la $S
# round 4: the evidence gradient (tools/grad_bench.py) and the fp64 potrf at BASELINE config 2's size under the kernel trace
rocprofv3 --kernel-trace --stats --output-format csv -d $O/grad_kt -o grad -- python3 tools/grad_bench.py > $O/grad_kt.log 2>&1 || true
cp $(find $O/grad_kt -name '*kernel_stats.csv' | head -1) $S/${TAG}_grad_n32768_kernel_stats.csv 2>/dev/null || true
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2_kt -o c2 -- python3 tools/potrf_only.py 16384 > $O/c2_kt.log 2>&1 || true
cp $(find $O/c2_kt -name '*kernel_stats.csv' | head -1) $S/${TAG}_potrf_n16384_kernel_stats.csv 2>/dev/null || true
python3 tools/trace_union.py $(find $O/c2_kt -name "*kernel_trace.csv" | head -1) --steps 3 --flops-per-step 1.46602e12 --json $S/${TAG}_potrf_n16384_union.json > /dev/null 2>&1 || true
ls -la $S
# round 4: the fp32 factorisation at config 3's size, both routes (operands split once into bf16 planes / split on the fly), kernel stats + counters
rocprofv3 --kernel-trace --stats --output-format csv -d $O/f32_kt -o f32 -- python3 tools/f32_presplit_ab.py 65536 > $O/f32_kt.log 2>&1 || true
cp $(find $O/f32_kt -name '*kernel_stats.csv' | head -1) $S/${TAG}_potrf_f32_n65536_kernel_stats.csv 2>/dev/null || true
grep potrf $O/f32_kt.log > $S/${TAG}_potrf_f32_n65536_ab.txt 2>/dev/null || true
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/f32_sq -o f32 -- python3 tools/f32_presplit_ab.py 32768 > $O/f32_sq.log 2>&1 || true
python3 tools/pmc_by_kernel.py $O/f32_sq > $S/${TAG}_potrf_f32_n32768_sq.txt 2>&1 || true
ls -la $S
