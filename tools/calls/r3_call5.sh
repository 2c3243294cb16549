#!/bin/bash
# round 3, GPU call 5: segments with the autograd anchor - parity tests (full logs kept), fp8 step test, C2 / C3 benches
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
echo "== segment tests"; timeout 600 python -m pytest tests/test_segments.py -m gpu -q -p no:cacheprovider > $O/r3e_test_segments.log 2>&1; tail -8 $O/r3e_test_segments.log
echo "== fp8 step + dist tests"; timeout 600 python -m pytest tests/test_fp8.py tests/test_dist.py -m gpu -q -p no:cacheprovider -k "fp8_step or rccl" > $O/r3e_test_fp8.log 2>&1; tail -4 $O/r3e_test_fp8.log
echo "== bench default (C2, auto)"; timeout 900 python bench.py --no-cpu-baseline > $O/r3e_bench_default.log 2>&1; tail -c 8000 $O/r3e_bench_default.log | grep -o '"ms_per_step": [0-9.]*\|"launch_mode": "[^"]*"\|"probe_ms_per_step": {[^}]*}\|"eager_ms_per_step": [0-9.]*\|"gpu_ms_per_step_by_piece": {[^]]*}}\|"secondary": {.*}}' | head -12
tail -3 $O/r3e_bench_default.log | grep -v "^{" | tail -3
echo "== bench C3 segments"; timeout 900 python bench.py --config c3 --no-cpu-baseline --no-kernel-timing > $O/r3e_bench_c3.log 2>&1; grep -o '"ms_per_step": [0-9.]*\|"launch_mode": "[^"]*"\|"host_enqueue_ms_per_step": [0-9.]*' $O/r3e_bench_c3.log; tail -3 $O/r3e_bench_c3.log | grep -v "^{"
echo done
