#!/bin/bash
# round 2, GPU call 4: after the workspace fix — full parity suite, step graph (serial D; forked D without record_stream),
# clean per-shape tuning, eager vs graph benches
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
ms() { grep -o '"ms_per_step": [0-9.]*' "$1" | tail -1; }
timeout 1500 python -X faulthandler -m pytest tests -m gpu -q > gpurun_out/r2d_tests_all.log 2>&1 < /dev/null; echo "all gpu tests: $(tail -1 gpurun_out/r2d_tests_all.log)"
for st in gan1 full; do
  timeout 120 python -X faulthandler tools/debug_stepgraph.py $st > gpurun_out/r2d_graph_$st.log 2>&1 < /dev/null
  echo "graph stage $st: rc=$? $(grep -E 'OK|Error|error|Fatal' gpurun_out/r2d_graph_$st.log | tail -1 | cut -c1-200)"
done
timeout 900 python tools/tune_gemm2.py c2 > gpurun_out/r2d_g2_tune.jsonl 2> gpurun_out/r2d_g2_tune.err < /dev/null; wc -l gpurun_out/r2d_g2_tune.jsonl
B="python bench.py --no-cpu-baseline --no-kernel-timing --steps 8 --warmup 2"
COMAT_STEP_GRAPH=0 timeout 300 $B > gpurun_out/r2d_bench_eager.log 2>&1 < /dev/null; echo "eager               $(ms gpurun_out/r2d_bench_eager.log)"
COMAT_STEP_GRAPH=1 timeout 300 $B > gpurun_out/r2d_bench_graph.log 2>&1 < /dev/null; echo "graph               $(ms gpurun_out/r2d_bench_graph.log)"
COMAT_STEP_GRAPH=1 COMAT_NORM_FUSED=1 timeout 300 $B > gpurun_out/r2d_bench_graph_gn2.log 2>&1 < /dev/null; echo "graph, GN 2-launch  $(ms gpurun_out/r2d_bench_graph_gn2.log)"
COMAT_STEP_GRAPH=1 COMAT_GEMM2=0 timeout 300 $B > gpurun_out/r2d_bench_graph_g0.log 2>&1 < /dev/null; echo "graph, general GEMM $(ms gpurun_out/r2d_bench_graph_g0.log)"
grep -h "launch_mode" gpurun_out/r2d_bench_graph.log | grep -o '"launch_mode": "[^"]*"'
tail -2 gpurun_out/r2d_bench_graph.log | cut -c1-300
