#!/bin/bash
# round 3, GPU call 9: hardware transpose reads in the fused attention kernels - bit-identity tests, microbenchmark, bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
echo "== flash tests"; timeout 600 python -m pytest tests/test_ops.py tests/test_fullsize.py tests/test_blip.py -m gpu -q -p no:cacheprovider -k "flash or attention or fullsize or blip" > $O/r3i_test.log 2>&1; tail -5 $O/r3i_test.log
echo "== microbenchmark"; timeout 300 python tools/mb_flash.py > $O/r3i_mb_flash.txt 2>&1; grep "trim=1" $O/r3i_mb_flash.txt
echo "== bench default (C2, auto)"; timeout 900 python bench.py --no-cpu-baseline > $O/r3i_bench_default.log 2>&1; tail -c 9000 $O/r3i_bench_default.log | grep -o '"ms_per_step": [0-9.]*\|"launch_mode": "[^"]*"\|"probe_ms_per_step": {[^}]*}\|"flash_[a-z_ (+)]*": {[^}]*}' | head -12
tail -3 $O/r3i_bench_default.log | grep -v "^{" | tail -3 | cut -c1-300
echo done
