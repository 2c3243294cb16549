#!/bin/bash
# round 3, GPU call 4: segment / step / op parity tests on the fixed workspace preparation (full logs kept), default bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
echo "== segment tests"; timeout 600 python -m pytest tests/test_segments.py -m gpu -q -p no:cacheprovider > $O/r3d_test_segments.log 2>&1; tail -6 $O/r3d_test_segments.log
echo "== step + sdxl + models + ops tests"; timeout 900 python -m pytest tests/test_step.py tests/test_sdxl.py tests/test_models.py tests/test_ops.py tests/test_checkpoint.py -m gpu -q -p no:cacheprovider > $O/r3d_test_step.log 2>&1; tail -6 $O/r3d_test_step.log
echo "== bench default (C2, auto)"; timeout 900 python bench.py --no-cpu-baseline > $O/r3d_bench_default.log 2>&1; tail -c 6000 $O/r3d_bench_default.log | grep -o '"ms_per_step": [0-9.]*\|"launch_mode": "[^"]*"\|"probe_ms_per_step": {[^}]*}\|"eager_ms_per_step": [0-9.]*\|"gpu_ms_per_step_by_piece": {.*}}}\|"secondary": {.*}}' | head -12
echo done
