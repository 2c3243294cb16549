#!/bin/bash
# round 3, call 19: packed softmax / dS arithmetic, max3 row maxima, block-uniform edge masks in the fused attention kernels -
# bit identity against the validated build (comat_amd/lib/ab/libcomat_hip_prev.so, swapped in on the box only), the flash
# tests, the kernel table
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
timeout 100 python tools/flash_bits.py dump /tmp/new.pt > $O/r3o_flash_bits.txt 2>&1
cp comat_amd/lib/libcomat_hip.so /tmp/new.so; cp comat_amd/lib/ab/libcomat_hip_prev.so comat_amd/lib/libcomat_hip.so
timeout 100 python tools/flash_bits.py dump /tmp/old.pt >> $O/r3o_flash_bits.txt 2>&1
cp /tmp/new.so comat_amd/lib/libcomat_hip.so
timeout 60 python tools/flash_bits.py compare /tmp/old.pt /tmp/new.pt >> $O/r3o_flash_bits.txt 2>&1; tail -8 $O/r3o_flash_bits.txt
timeout 100 python -m pytest tests/test_ops.py -m gpu -q -x -p no:cacheprovider -k "flash or attention" > $O/r3o_test_flash.log 2>&1; tail -2 $O/r3o_test_flash.log
timeout 60 python tools/mb_flash.py kt > $O/r3o_mb_flash_kt_packed.txt 2>&1; grep "flash kt" $O/r3o_mb_flash_kt_packed.txt | cut -c1-170
echo done
