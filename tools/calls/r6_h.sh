#!/bin/bash
# round 6: C2 A/B of the tail-column LoRA products with tuned plans (same box, alternating)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
for t in 0 1 0 1; do
  echo "tail=$t $(COMAT_LORA_TAIL=$t COMAT_SECONDARY=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)"
done
echo done
