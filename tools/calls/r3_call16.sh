#!/bin/bash
# round 3, GPU call 16: 64-key forward experiment (flash_kt = 2): parity + microbenchmark
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
echo "== tests"; timeout 300 python -m pytest tests/test_ops.py -m gpu -q -p no:cacheprovider -k "flash" 2>&1 | tail -4
echo "== microbenchmark"; timeout 300 python tools/mb_flash.py > $O/r3l_mb_flash_kt.txt 2>&1; grep "flash kt" $O/r3l_mb_flash_kt.txt
echo done
