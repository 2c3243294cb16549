#!/bin/bash
# round 6: staged epilogue (libcomat_hip.so) against the lane = row epilogue in its new one-copy form (libcomat_hip_stage0.so)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
: > $O/r6w_scan.txt
for lib in libcomat_hip.so libcomat_hip_stage0.so; do
MB_ONLY=unet MB_CFGS=1,2,6 COMAT_LIB_PATH=comat_amd/lib/$lib timeout 300 python tools/mb_diag.py 2>&1 | grep -v amdgpu.ids >> $O/r6w_scan.txt
MB_ONLY=kscan MB_CFGS=1,12 COMAT_LIB_PATH=comat_amd/lib/$lib timeout 300 python tools/mb_diag.py 2>&1 | grep -v amdgpu.ids >> $O/r6w_scan.txt
done
for lib in libcomat_hip.so libcomat_hip_stage0.so libcomat_hip.so libcomat_hip_stage0.so; do
  echo "c2 $lib $(COMAT_LIB_PATH=comat_amd/lib/$lib COMAT_SECONDARY=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)" | tee -a $O/r6w_ab.txt
done
