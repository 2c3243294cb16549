#!/bin/bash
# round 2, GPU call 16: GroupNorm with the finalisation in the apply kernel's prologue (norm_fused = 2): parity, then A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
echo "== tests"; timeout 300 python -m pytest tests/test_ops.py -m gpu -q -p no:cacheprovider -k "groupnorm" 2>&1 | tail -4
echo "== bench norm_fused=2"; COMAT_NORM_FUSED=2 COMAT_STEP_GRAPH=1 timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing > $O/r2p_bench_nf2.log 2>&1; grep -o '"ms_per_step": [0-9.]*' $O/r2p_bench_nf2.log
echo "== bench norm_fused=0"; COMAT_NORM_FUSED=0 COMAT_STEP_GRAPH=1 timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing > $O/r2p_bench_nf0.log 2>&1; grep -o '"ms_per_step": [0-9.]*' $O/r2p_bench_nf0.log
echo done
