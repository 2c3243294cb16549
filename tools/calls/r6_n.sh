#!/bin/bash
# round 6: hand-issued fragment reads with counted lgkmcnt waits (gemm2.hip G2_PIN = 3) against the compiler-scheduled reads: microbenchmark
# (with and without the LDS-DMA), parity of the pinned build, C2 step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
: > $O/r6n_diag.txt
for lib in libcomat_hip.so libcomat_hip_pin3.so libcomat_hip_d3.so libcomat_hip_pin3d3.so libcomat_hip.so libcomat_hip_pin3.so; do
  MB_CFGS=1,3,12,11 COMAT_LIB_PATH=comat_amd/lib/$lib timeout 300 python tools/mb_diag.py 2>&1 | grep -v amdgpu.ids >> $O/r6n_diag.txt
done
cat $O/r6n_diag.txt
echo "== parity (pinned build)"
COMAT_LIB_PATH=comat_amd/lib/libcomat_hip_pin3.so timeout 1500 python -m pytest tests/test_ops.py -m gpu -q -p no:cacheprovider -x -k "gemm or conv or lora or geglu" > $O/r6n_tests.log 2>&1; tail -4 $O/r6n_tests.log
echo "== C2 A/B"
for lib in libcomat_hip.so libcomat_hip_pin3.so libcomat_hip.so libcomat_hip_pin3.so; do
  echo "$lib $(COMAT_LIB_PATH=comat_amd/lib/$lib COMAT_SECONDARY=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)" | tee -a $O/r6n_c2_ab.txt
done
echo done
