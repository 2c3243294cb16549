#!/bin/bash
# round 6: time against K at M = N = 4096 (slope = one k-tile, intercept = prologue + epilogue) for 128x128 and 256x256, with and without the epilogue
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
: > $O/r6r_kscan.txt
for lib in libcomat_hip.so libcomat_hip_d5.so; do
MB_ONLY=kscan MB_CFGS=1,12 COMAT_LIB_PATH=comat_amd/lib/$lib timeout 300 python tools/mb_diag.py 2>&1 | grep -v amdgpu.ids >> $O/r6r_kscan.txt
done
cat $O/r6r_kscan.txt
