#!/bin/bash
# round 6: C4 (SDXL 512^2) and C3 A/B of the tail-column LoRA products (same box, alternating)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
for t in 0 1 0 1; do
  echo "c4 tail=$t $(COMAT_LORA_TAIL=$t timeout 900 python bench.py --config c4 --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)"
done
for t in 0 1; do
  echo "c3 tail=$t $(COMAT_LORA_TAIL=$t timeout 900 python bench.py --config c3 --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)"
done
echo done
