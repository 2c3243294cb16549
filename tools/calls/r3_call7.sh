#!/bin/bash
# round 3, GPU call 7: D step inside the head's backward graph (with its eager warm-up) - segment + norm parity tests, C2 / C3
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
echo "== segment + norm + fp8-step tests"; timeout 900 python -m pytest tests/test_segments.py tests/test_ops.py tests/test_fp8.py -m gpu -q -p no:cacheprovider -k "segment or groupnorm or fp8_step" > $O/r3g_test.log 2>&1; tail -6 $O/r3g_test.log
echo "== bench default (C2, auto)"; timeout 900 python bench.py --no-cpu-baseline > $O/r3g_bench_default.log 2>&1; tail -c 8000 $O/r3g_bench_default.log | grep -o '"ms_per_step": [0-9.]*\|"launch_mode": "[^"]*"\|"probe_ms_per_step": {[^}]*}\|"eager_ms_per_step": [0-9.]*\|"gpu_ms_per_step_by_piece": {[^]]*}}\|"secondary": {.*}}' | head -12
tail -3 $O/r3g_bench_default.log | grep -v "^{" | tail -3
echo done
