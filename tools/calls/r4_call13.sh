#!/bin/bash
# round 4, call 13: what each part of the 2-tile forward attention loop costs in place - builds of attention.hip with parts
# deleted (COMAT_FLASH_DIAG bits: 1 exp2, 2 P V products, 4 Q K^T products, 8 tile loads + LDS stores, 16 barrier, 32 running max)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out; mkdir -p $O
AB=$GRAFT_REPO_ROOT/comat_amd/lib/ab
for d in 0 1 2 4 8 16 24 32 6 7 63 0; do
  COMAT_LIB_PATH=$AB/libcomat_d$d.so timeout 120 python tools/mb_flash_diag.py 2>&1 | grep -v amdgpu.ids
done | tee $O/r4m_flash_diag.txt
echo done
