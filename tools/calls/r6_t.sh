#!/bin/bash
# round 6: staged epilogue, one copy of the arithmetic in the instruction stream, against the round-5 epilogue: K scan + parity
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
: > $O/r6t_kscan.txt
for lib in libcomat_hip_stage0.so libcomat_hip.so; do
MB_ONLY=kscan MB_CFGS=1,12 COMAT_LIB_PATH=comat_amd/lib/$lib timeout 300 python tools/mb_diag.py 2>&1 | grep -v amdgpu.ids >> $O/r6t_kscan.txt
done
cat $O/r6t_kscan.txt
echo "== parity"
timeout 1500 python -m pytest tests/test_ops.py -m gpu -q -p no:cacheprovider -x -k "gemm or conv or lora or geglu" > $O/r6t_tests.log 2>&1; tail -4 $O/r6t_tests.log
