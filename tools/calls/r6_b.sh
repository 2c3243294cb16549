#!/bin/bash
# round 6, call 2: cycle timeline of the ping-pong loop + parity of the split-K tail fix
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
echo "== parity cfg 12/13"
timeout 900 python -m pytest tests/test_ops.py -m gpu -q -p no:cacheprovider -k "(test_gemm2_matches_reference or test_gemm2_segments_and_conv) and (-12] or -13])" > $O/r6b_tests.log 2>&1; tail -5 $O/r6b_tests.log
echo "== timeline"
COMAT_LIB_PATH=$PWD/comat_amd/lib/libcomat_hip_tl.so timeout 600 python tools/pp_timeline.py > $O/r6b_timeline.txt 2>&1; grep -c . $O/r6b_timeline.txt
echo done
