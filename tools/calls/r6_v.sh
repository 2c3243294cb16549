#!/bin/bash
# round 6: intercept / slope of the pipelined kernel at the UNet's shapes, with and without the epilogue (libcomat_hip_d5.so)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
: > $O/r6v_unet.txt
for lib in libcomat_hip.so libcomat_hip_d5.so; do
MB_ONLY=unet MB_CFGS=1,2,6 COMAT_LIB_PATH=comat_amd/lib/$lib timeout 300 python tools/mb_diag.py 2>&1 | grep -v amdgpu.ids >> $O/r6v_unet.txt
done
cat $O/r6v_unet.txt
