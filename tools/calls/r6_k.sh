#!/bin/bash
# round 6: deletion experiment on the pipelined GEMM's k-loop (make diag builds), block shapes 128x128 (cfg 1), 256x128 (3), 256x256 (12)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
: > $O/r6k_diag.txt
for lib in libcomat_hip.so libcomat_hip_d1.so libcomat_hip_d2.so libcomat_hip_d3.so libcomat_hip_d4.so libcomat_hip.so; do
  MB_CFGS=1,3,12 COMAT_LIB_PATH=comat_amd/lib/$lib timeout 300 python tools/mb_diag.py 2>&1 | grep -v amdgpu.ids >> $O/r6k_diag.txt
done
cat $O/r6k_diag.txt
echo "== parity cfg 12"
timeout 900 python -m pytest tests/test_ops.py -m gpu -q -p no:cacheprovider -x -k "(gemm2_matches_reference or gemm2_segments_and_conv) and 12" > $O/r6k_tests.log 2>&1; tail -4 $O/r6k_tests.log
