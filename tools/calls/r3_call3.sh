#!/bin/bash
# round 3, GPU call 3: step segments - parity tests (replay == eager, bit for bit), then C2 / C3 benches
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
echo "== segment tests"; timeout 600 python -m pytest tests/test_segments.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -25
echo "== op + step + sdxl tests"; timeout 900 python -m pytest tests/test_ops.py tests/test_step.py tests/test_sdxl.py tests/test_models.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8
echo "== bench default (C2, auto)"; timeout 900 python bench.py --no-cpu-baseline > $O/r3c_bench_default.log 2>&1; tail -c 4500 $O/r3c_bench_default.log | grep -o '"ms_per_step": [0-9.]*\|"launch_mode": "[^"]*"\|"probe_ms_per_step": {[^}]*}\|"eager_ms_per_step": [0-9.]*\|"secondary": {.*}}' | head -8
tail -5 $O/r3c_bench_default.log | grep -v "^{" | tail -3
echo "== bench C3 segments"; timeout 900 python bench.py --config c3 --no-cpu-baseline --no-kernel-timing > $O/r3c_bench_c3.log 2>&1; grep -o '"ms_per_step": [0-9.]*\|"launch_mode": "[^"]*"\|"host_enqueue_ms_per_step": [0-9.]*' $O/r3c_bench_c3.log; tail -3 $O/r3c_bench_c3.log | grep -v "^{"
echo done
