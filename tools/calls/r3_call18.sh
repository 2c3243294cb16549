#!/bin/bash
# round 3, call 18: MFMA results in VGPRs (-mllvm -amdgpu-mfma-vgpr-form=1).  A = attention.hip built with it (the tree's
# library), B = gemm2.hip as well (comat_amd/lib/ab/, swapped in on the box only).  Flash tests + kernel table on A, the C2
# step on A and on B, the GEMM / conv tests on B.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
timeout 200 python -m pytest tests/test_ops.py -m gpu -q -x -p no:cacheprovider -k "flash or attention" > $O/r3n_test_flash.log 2>&1; tail -2 $O/r3n_test_flash.log
timeout 120 python tools/mb_flash.py kt > $O/r3n_mb_flash_kt_vgpr.txt 2>&1; grep "flash kt" $O/r3n_mb_flash_kt_vgpr.txt | cut -c1-170
echo "== C2 step, A"; COMAT_SECONDARY=0 timeout 200 python bench.py --steps 10 --no-cpu-baseline --no-kernel-timing > $O/r3n_bench_A.log 2>&1; grep -o '"ms_per_step": [0-9.]*\|"probe_ms_per_step": {[^}]*}\|"eager_ms_per_step": [0-9.]*' $O/r3n_bench_A.log | head -3
cp comat_amd/lib/ab/libcomat_hip.so comat_amd/lib/libcomat_hip.so
echo "== C2 step, B"; COMAT_SECONDARY=0 timeout 200 python bench.py --steps 10 --no-cpu-baseline --no-kernel-timing > $O/r3n_bench_B.log 2>&1; grep -o '"ms_per_step": [0-9.]*\|"probe_ms_per_step": {[^}]*}\|"eager_ms_per_step": [0-9.]*' $O/r3n_bench_B.log | head -3
timeout 200 python -m pytest tests/test_ops.py tests/test_fp8.py -m gpu -q -x -p no:cacheprovider -k "gemm or conv or tt or fp8" > $O/r3n_test_gemm_B.log 2>&1; tail -2 $O/r3n_test_gemm_B.log
echo done
