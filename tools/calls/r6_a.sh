#!/bin/bash
# round 6, call 1: parity of the ping-pong shapes (g2_cfg 12 / 13) + microbench against the lockstep shapes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
echo "== parity cfg 12/13"
timeout 900 python -m pytest tests/test_ops.py -m gpu -q -p no:cacheprovider -x -k "(test_gemm2_matches_reference or test_gemm2_segments_and_conv) and (-12] or -13])" > $O/r6a_tests.log 2>&1; tail -5 $O/r6a_tests.log
echo "== mb_pp"
timeout 1200 python tools/mb_pp.py > $O/r6a_mb_pp.txt 2>&1; grep BEST $O/r6a_mb_pp.txt | head -40
echo done
