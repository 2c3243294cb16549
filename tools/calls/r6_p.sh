#!/bin/bash
# round 6: the interleaved k-loop (gemm2.hip G2_ILV = 1: fragment reads and LDS-DMA pieces placed between the MFMAs) against the round-5 loop
# (libcomat_hip_ilv0.so): microbenchmark, parity, C2 step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
: > $O/r6p_diag.txt
for lib in libcomat_hip_ilv0.so libcomat_hip.so libcomat_hip_ilv0.so libcomat_hip.so; do
  MB_CFGS=1,3,12,11,2,6 COMAT_LIB_PATH=comat_amd/lib/$lib timeout 300 python tools/mb_diag.py 2>&1 | grep -v amdgpu.ids >> $O/r6p_diag.txt
done
cat $O/r6p_diag.txt
echo "== parity"
timeout 1500 python -m pytest tests/test_ops.py -m gpu -q -p no:cacheprovider -x -k "gemm or conv or lora or geglu" > $O/r6p_tests.log 2>&1; tail -4 $O/r6p_tests.log
echo "== C2 A/B"
for lib in libcomat_hip_ilv0.so libcomat_hip.so libcomat_hip_ilv0.so libcomat_hip.so; do
  echo "$lib $(COMAT_LIB_PATH=comat_amd/lib/$lib COMAT_SECONDARY=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)" | tee -a $O/r6p_c2_ab.txt
done
echo done
