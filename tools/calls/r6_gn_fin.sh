#!/bin/bash
# round 6: GroupNorm with the finalize folded into the apply kernel (norm_fused = 2: two launches) against the default (three), C2 step, same box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1 COMAT_SECONDARY=0
O=gpurun_out
mkdir -p $O
for nf in 3 2 3 2; do
  echo "== norm_fused=$nf"; COMAT_NORM_FUSED=$nf timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1
done
python tools/mb_gn.py 2>/dev/null | tail -16
echo done
