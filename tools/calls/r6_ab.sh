#!/bin/bash
# round 6: is bench.py healthy on this box?  (call r6_aa: every bench process died in its first host-to-device copy)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
COMAT_SECONDARY=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $O/r6ab_c2.log 2>&1; tail -c 600 $O/r6ab_c2.log
timeout 900 python -m pytest tests/test_fp8.py -m gpu -q -p no:cacheprovider -x -s > $O/r6ab_tests.log 2>&1; grep "fp8 UNet" $O/r6ab_tests.log | tail -4; tail -3 $O/r6ab_tests.log
COMAT_SECONDARY=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $O/r6ab_c2b.log 2>&1; tail -c 300 $O/r6ab_c2b.log
echo done
