#!/bin/bash
# round 6: 256x256 lockstep block shape (g2_cfg 12: 8 waves of 128x64) - parity, microbenchmark against the other shapes; C2 A/B of the tail-column LoRA products
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
echo "== parity cfg 12"
timeout 900 python -m pytest tests/test_ops.py -m gpu -q -p no:cacheprovider -x -k "(gemm2_matches_reference or gemm2_segments_and_conv) and 12-" > $O/r6j_tests.log 2>&1; tail -4 $O/r6j_tests.log
echo "== mb_big"
timeout 900 python tools/mb_big.py > $O/r6j_mb_big.txt 2>&1; grep -c . $O/r6j_mb_big.txt; grep BEST $O/r6j_mb_big.txt | head -40
echo "== C2 A/B (COMAT_LORA_TAIL)"
for t in 0 1 0 1; do
  echo "tail=$t $(COMAT_LORA_TAIL=$t COMAT_SECONDARY=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)" | tee -a $O/r6j_tail_ab.txt
done
echo done
