#!/bin/bash
# round 2, GPU call 2: parity of the new kernels / step graph, GEMM microbenchmark, A/B benches, kernel trace
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
ms() { grep -o '"ms_per_step": [0-9.]*' "$1" | tail -1; }
# 1. the new kernels first, each group in its own process (a device fault must not take the other results down)
timeout 900 python -m pytest tests/test_ops.py -m gpu -q -x -k "gemm2 or inlaunch or adamw_step or many_tokens" > gpurun_out/r2b_tests_new.log 2>&1 < /dev/null; echo "new-kernel tests: $(tail -1 gpurun_out/r2b_tests_new.log)"
timeout 600 python -m pytest tests/test_step.py -m gpu -q -k "graphed" > gpurun_out/r2b_tests_graph.log 2>&1 < /dev/null; echo "graphed step: $(tail -1 gpurun_out/r2b_tests_graph.log)"
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r2b_tests_all.log 2>&1 < /dev/null; echo "all gpu tests: $(tail -1 gpurun_out/r2b_tests_all.log)"
# 2. kernel microbenchmark
timeout 900 python tools/mb_gemm2.py > gpurun_out/r2b_mb_gemm2.txt 2>&1 < /dev/null; grep -c BEST gpurun_out/r2b_mb_gemm2.txt
# 3. A/B of the whole step
B="python bench.py --no-cpu-baseline --no-kernel-timing --steps 8 --warmup 2"
COMAT_GEMM2=0 COMAT_STEP_GRAPH=0 timeout 300 $B > gpurun_out/r2b_bench_g0_e.log 2>&1 < /dev/null; echo "general kernel, eager   $(ms gpurun_out/r2b_bench_g0_e.log)"
COMAT_GEMM2=1 COMAT_STEP_GRAPH=0 timeout 300 $B > gpurun_out/r2b_bench_g1_e.log 2>&1 < /dev/null; echo "gemm2, eager            $(ms gpurun_out/r2b_bench_g1_e.log)"
COMAT_GEMM2=0 COMAT_STEP_GRAPH=1 timeout 300 $B > gpurun_out/r2b_bench_g0_g.log 2>&1 < /dev/null; echo "general kernel, graph   $(ms gpurun_out/r2b_bench_g0_g.log)"
COMAT_GEMM2=1 COMAT_STEP_GRAPH=1 timeout 300 $B > gpurun_out/r2b_bench_g1_g.log 2>&1 < /dev/null; echo "gemm2, graph (default)  $(ms gpurun_out/r2b_bench_g1_g.log)"
# 4. kernel trace of the default configuration (eager launches: one dispatch record per kernel either way)
cd /tmp && COMAT_STEP_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2b_prof -o c2 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/r2b_prof.log 2>&1 < /dev/null
cd $R
DB=$(ls gpurun_out/r2b_prof/*results.db gpurun_out/r2b_prof/*/*results.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocpd_summary.py $DB 3 > gpurun_out/r2b_kernel_trace.txt; python tools/rocpd_timeline.py $DB 0.5 > gpurun_out/r2b_timeline.txt; rm -rf gpurun_out/r2b_prof; head -25 gpurun_out/r2b_kernel_trace.txt; fi
tail -3 gpurun_out/r2b_prof.log | cut -c1-400
