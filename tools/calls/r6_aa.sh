#!/bin/bash
# round 6: fp8 delayed scaling (ABI 8) on the GPU: its tests, then C5 with delayed against just-in-time scales (alternating), then C2 on the
# re-tuned plan table
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_fp8.py tests/test_abi.py -m gpu -q -p no:cacheprovider -x -s > $O/r6aa_tests.log 2>&1; grep "fp8 UNet" $O/r6aa_tests.log | tail -4; tail -3 $O/r6aa_tests.log
for mode in delayed jit delayed jit; do
  echo "c5 $mode $(COMAT_FP8_SCALING=$mode timeout 700 python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing 2>$O/r6aa_c5_$mode.err | grep -o '"ms_per_step": [0-9.]*' | head -1)" | tee -a $O/r6aa_c5_ab.txt
done
for i in 1 2; do
  echo "c2 $(COMAT_SECONDARY=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)" | tee -a $O/r6aa_c2.txt
done
echo done
