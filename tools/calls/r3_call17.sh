#!/bin/bash
# round 3, call 17: fused attention backward as ONE launch (option flash_merge) - bit identity against the separate kernels,
# the other flash tests on the refactored bodies, the microbenchmark table
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
timeout 240 python -m pytest tests/test_ops.py -m gpu -q -x -p no:cacheprovider -k "flash or attention" > $O/r3m_test_flash.log 2>&1; tail -3 $O/r3m_test_flash.log
timeout 120 python tools/mb_flash.py merge > $O/r3m_mb_flash_merge.txt 2>&1; cat $O/r3m_mb_flash_merge.txt | cut -c1-170
echo done
