#!/bin/bash
# round 3, GPU call 8: generator-side D loss on its own stream (eager, whole-step graph, segments) - parity tests; C2 / C4 / C5
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
echo "== segment + step + sdxl tests"; timeout 900 python -m pytest tests/test_segments.py tests/test_step.py tests/test_sdxl.py -m gpu -q -p no:cacheprovider > $O/r3h_test.log 2>&1; tail -6 $O/r3h_test.log
echo "== bench default (C2, auto)"; timeout 900 python bench.py --no-cpu-baseline > $O/r3h_bench_default.log 2>&1; tail -c 8000 $O/r3h_bench_default.log | grep -o '"ms_per_step": [0-9.]*\|"launch_mode": "[^"]*"\|"probe_ms_per_step": {[^}]*}\|"eager_ms_per_step": [0-9.]*\|"gpu_ms_per_step_by_piece": {[^]]*}}' | head -12
tail -3 $O/r3h_bench_default.log | grep -v "^{" | tail -3
for c in c4 c5; do
echo "== bench $c (segments)"; timeout 1200 python bench.py --config $c --no-cpu-baseline --no-kernel-timing > $O/r3h_bench_$c.log 2>&1; grep -o '"ms_per_step": [0-9.]*\|"launch_mode": "[^"]*"\|"host_enqueue_ms_per_step": [0-9.]*' $O/r3h_bench_$c.log; tail -4 $O/r3h_bench_$c.log | grep -v "^{" | cut -c1-400
done
echo done
