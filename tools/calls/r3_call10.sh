#!/bin/bash
# round 3, GPU call 10: text K/V shared by the no-grad graphs of a sampler call; strided-conv dgrad on the pipelined kernel
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
echo "== tests"; timeout 900 python -m pytest tests/test_ops.py tests/test_models.py tests/test_step.py tests/test_sdxl.py tests/test_segments.py -m gpu -q -p no:cacheprovider -k "conv or sampler or models or step or sdxl or segment or gt_latent" > $O/r3j_test.log 2>&1; tail -5 $O/r3j_test.log
echo "== bench default (C2, auto)"; timeout 900 python bench.py --no-cpu-baseline > $O/r3j_bench_default.log 2>&1; tail -c 9000 $O/r3j_bench_default.log | grep -o '"ms_per_step": [0-9.]*\|"launch_mode": "[^"]*"\|"probe_ms_per_step": {[^}]*}\|"unet no-grad": {[^}]*}' | head -12
tail -3 $O/r3j_bench_default.log | grep -v "^{" | tail -3 | cut -c1-300
echo "== bench c4 (segments)"; timeout 1200 python bench.py --config c4 --no-cpu-baseline --no-kernel-timing > $O/r3j_bench_c4.log 2>&1; grep -o '"ms_per_step": [0-9.]*\|"launch_mode": "[^"]*"' $O/r3j_bench_c4.log; tail -4 $O/r3j_bench_c4.log | grep -v "^{" | cut -c1-300
echo done
