#!/bin/bash
# round 2, GPU call 6: k-major pipelined GEMM + transposed up factors, GroupNorm apply rewrite, D stream forked inside the
# capture without nested forks, benches
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
ms() { grep -o '"ms_per_step": [0-9.]*' "$1" | tail -1; }
timeout 600 python -X faulthandler -m pytest tests/test_ops.py -m gpu -q -x -k "k_major or groupnorm or lora or layernorm" > gpurun_out/r2f_tests_new.log 2>&1 < /dev/null; echo "new tests: $(tail -1 gpurun_out/r2f_tests_new.log)"
for st in gan2 full2; do
  timeout 120 python -X faulthandler tools/debug_stepgraph.py $st > gpurun_out/r2f_graph_$st.log 2>&1 < /dev/null
  echo "graph stage $st: rc=$? $(grep -E 'OK|Error|error|Fatal' gpurun_out/r2f_graph_$st.log | tail -1 | cut -c1-200)"
done
timeout 1500 python -X faulthandler -m pytest tests -m gpu -q > gpurun_out/r2f_tests_all.log 2>&1 < /dev/null; echo "all gpu tests: $(tail -1 gpurun_out/r2f_tests_all.log)"
B="python bench.py --no-cpu-baseline --no-kernel-timing --steps 8 --warmup 2"
COMAT_STEP_GRAPH=0 timeout 300 $B > gpurun_out/r2f_bench_eager.log 2>&1 < /dev/null; echo "eager                 $(ms gpurun_out/r2f_bench_eager.log)"
COMAT_STEP_GRAPH=1 timeout 300 $B > gpurun_out/r2f_bench_graph.log 2>&1 < /dev/null; echo "graph (serial D)      $(ms gpurun_out/r2f_bench_graph.log)"
COMAT_STEP_GRAPH=1 COMAT_GEMM2_TT=0 timeout 300 $B > gpurun_out/r2f_bench_graph_tt0.log 2>&1 < /dev/null; echo "graph, general TT     $(ms gpurun_out/r2f_bench_graph_tt0.log)"
if grep -q "OK" gpurun_out/r2f_graph_full2.log; then COMAT_STEP_GRAPH=1 COMAT_GRAPH_D=fork timeout 300 $B > gpurun_out/r2f_bench_graph_fork.log 2>&1 < /dev/null; echo "graph (forked D)      $(ms gpurun_out/r2f_bench_graph_fork.log)"; fi
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r2f_bench_timed.log 2>&1 < /dev/null; tail -1 gpurun_out/r2f_bench_timed.log | cut -c1-3000
