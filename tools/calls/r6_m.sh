#!/bin/bash
# round 6: does the fill rate depend on the row width of the LDS-DMA (64-byte rows = half cache lines vs 128-byte rows)?  128x128 with one
# block per CU either way: cfg 5 (64-byte k-tiles, ring of 6) vs cfg 11 (128-byte k-tiles, ring of 4), product build and the no-MFMA build
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
: > $O/r6m_diag.txt
for lib in libcomat_hip.so libcomat_hip_d1.so libcomat_hip_d3.so; do
  MB_CFGS=5,11,1,12 COMAT_LIB_PATH=comat_amd/lib/$lib timeout 300 python tools/mb_diag.py 2>&1 | grep -v amdgpu.ids >> $O/r6m_diag.txt
done
cat $O/r6m_diag.txt
