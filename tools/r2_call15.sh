#!/bin/bash
# round 2, GPU call 15: last check of the final tree - step tests (incl. the split graph under thread-local capture) and the
# default bench exactly as the driver runs it
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
echo "== tests"; timeout 400 python -m pytest tests/test_step.py tests/test_abi.py tests/test_fp8.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
echo "== bench (driver form)"; timeout 500 python bench.py > $O/r2o_bench_default.log 2>&1; grep -o '"ms_per_step": [0-9.]*\|"launch_mode": "[^"]*"\|"value": [0-9.]*\|"frac": [0-9.]*\|"event_pair_overhead_us": [0-9.]*' $O/r2o_bench_default.log | head -8
echo done
