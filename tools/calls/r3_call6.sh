#!/bin/bash
# round 3, GPU call 6: one-launch GroupNorm (microbenchmark + parity), D step inside the head's backward graph,
# the whole GPU suite on the new build, C2 / C3 benches
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
echo "== GN microbenchmark"; timeout 300 python tools/mb_gn.py > $O/r3f_mb_gn.txt 2>&1; cat $O/r3f_mb_gn.txt | tail -14
echo "== GPU suite"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/r3f_test_all.log 2>&1; tail -6 $O/r3f_test_all.log
echo "== bench default (C2, auto)"; timeout 900 python bench.py --no-cpu-baseline > $O/r3f_bench_default.log 2>&1; tail -c 8000 $O/r3f_bench_default.log | grep -o '"ms_per_step": [0-9.]*\|"launch_mode": "[^"]*"\|"probe_ms_per_step": {[^}]*}\|"eager_ms_per_step": [0-9.]*\|"gpu_ms_per_step_by_piece": {[^]]*}}\|"secondary": {.*}}' | head -12
tail -3 $O/r3f_bench_default.log | grep -v "^{" | tail -3
echo done
