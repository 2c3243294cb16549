#!/bin/bash
# round 4, call 10: C4 (SDXL 512^2) on one box: defaults / without merged no-grad weights / without any round-4 default
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
run() { env "$@" timeout 400 python bench.py --config c4 --steps 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"build_s": [0-9.]*' | head -2 | tr '\n' ' '; echo; }
echo "== C4 defaults"; run A=1
echo "== C4 COMAT_NOGRAD_MERGED=0"; run COMAT_NOGRAD_MERGED=0
echo "== C4 round-3 settings (NOGRAD_MERGED=0 GEGLU_FUSED=0 GEMM3=0 G2_ORDER=0 FLASH_XCD=0)"; run COMAT_NOGRAD_MERGED=0 COMAT_GEGLU_FUSED=0 COMAT_GEMM3=0 COMAT_G2_ORDER=0 COMAT_FLASH_XCD=0
echo done
