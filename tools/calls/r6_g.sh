#!/bin/bash
# round 6: tail-column epilogue (epi2 = 4) + LoRA products riding in their neighbours' launches: parity, the SDXL full-size leg, C2 A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
echo "== parity"
COMAT_TEST_REPORT=$O/r6g_report.txt timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "tail_columns or lora or abi or c4_full or train_step or merged" > $O/r6g_tests.log 2>&1; tail -6 $O/r6g_tests.log; cat $O/r6g_report.txt 2>/dev/null | tail -5
echo "== C2 A/B (COMAT_LORA_TAIL)"
for t in 0 1 0 1; do
  echo "tail=$t $(COMAT_LORA_TAIL=$t COMAT_SECONDARY=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)"
done
echo done
