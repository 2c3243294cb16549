#!/bin/bash
# round 3, GPU call 12: the GPU suite in two processes (which test failed before the segfault of call 11?)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
echo "== part 1 (everything but segments / full size)"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -rf --tb=short --deselect tests/test_segments.py --deselect tests/test_zz_fullsize_c1.py > $O/r3k_test_1.log 2>&1; tail -15 $O/r3k_test_1.log | cut -c1-300
echo "== part 2 (segments + full size)"; timeout 900 python -m pytest tests/test_segments.py tests/test_zz_fullsize_c1.py -m gpu -q -p no:cacheprovider -rf --tb=short > $O/r3k_test_2.log 2>&1; tail -8 $O/r3k_test_2.log | cut -c1-300
echo done
