#!/bin/bash
# round 2, GPU call 3: pipelined mainloop + fence-free split-K parity, step-graph capture located stage by stage,
# fused-attention variants timed, per-shape plan tuning, eager bench
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
ms() { grep -o '"ms_per_step": [0-9.]*' "$1" | tail -1; }
timeout 900 python -X faulthandler -m pytest tests/test_ops.py -m gpu -q -x > gpurun_out/r2c_tests_ops.log 2>&1 < /dev/null; echo "test_ops: $(tail -1 gpurun_out/r2c_tests_ops.log)"
for st in fwd bwd0 bwd1 opt gan0 gan1 full; do
  timeout 120 python -X faulthandler tools/debug_stepgraph.py $st > gpurun_out/r2c_graph_$st.log 2>&1 < /dev/null
  echo "graph stage $st: rc=$? $(grep -E 'OK|Error|error|Fatal' gpurun_out/r2c_graph_$st.log | tail -1 | cut -c1-200)"
done
timeout 300 python tools/mb_flash.py > gpurun_out/r2c_mb_flash.txt 2>&1 < /dev/null; tail -2 gpurun_out/r2c_mb_flash.txt | cut -c1-200
timeout 900 python tools/tune_gemm2.py c2 > gpurun_out/r2c_g2_tune.jsonl 2> gpurun_out/r2c_g2_tune.err < /dev/null; wc -l gpurun_out/r2c_g2_tune.jsonl; tail -2 gpurun_out/r2c_g2_tune.err | cut -c1-300
B="python bench.py --no-cpu-baseline --no-kernel-timing --steps 8 --warmup 2"
COMAT_STEP_GRAPH=0 timeout 300 $B > gpurun_out/r2c_bench_eager.log 2>&1 < /dev/null; echo "gemm2 pipelined, eager  $(ms gpurun_out/r2c_bench_eager.log)"
COMAT_STEP_GRAPH=0 COMAT_NORM_FUSED=0 timeout 300 $B > gpurun_out/r2c_bench_eager_gn3.log 2>&1 < /dev/null; echo "  ... GroupNorm 3-launch $(ms gpurun_out/r2c_bench_eager_gn3.log)"
timeout 900 python -X faulthandler -m pytest tests -m gpu -q --deselect tests/test_step.py::test_graphed_step_matches_eager --ignore=tests/test_ops.py > gpurun_out/r2c_tests_rest.log 2>&1 < /dev/null; echo "other gpu tests: $(tail -1 gpurun_out/r2c_tests_rest.log)"
