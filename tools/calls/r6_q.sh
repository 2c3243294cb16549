#!/bin/bash
# round 6: 256x256 with a ring of 5 (all 160 KiB of LDS) and the grouped tile order (g2_order = 3: 4 x 8 patches per XCD) on the large problems
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
: > $O/r6q_diag.txt
for rep in 1 2; do
MB_CFGS=1,12 COMAT_LIB_PATH=comat_amd/lib/libcomat_hip.so timeout 300 python tools/mb_diag.py 2>&1 | grep -v amdgpu.ids >> $O/r6q_diag.txt
MB_ORDER=3 MB_CFGS=1,12 COMAT_LIB_PATH=comat_amd/lib/libcomat_hip.so timeout 300 python tools/mb_diag.py 2>&1 | grep -v amdgpu.ids >> $O/r6q_diag.txt
MB_CFGS=12 COMAT_LIB_PATH=comat_amd/lib/libcomat_hip_nst5.so timeout 300 python tools/mb_diag.py 2>&1 | grep -v amdgpu.ids >> $O/r6q_diag.txt
MB_ORDER=3 MB_CFGS=12 COMAT_LIB_PATH=comat_amd/lib/libcomat_hip_nst5.so timeout 300 python tools/mb_diag.py 2>&1 | grep -v amdgpu.ids >> $O/r6q_diag.txt
done
cat $O/r6q_diag.txt
