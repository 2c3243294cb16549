#!/bin/bash
# round 6: fused attention forward emitting the e4m3 bytes for its output projection (comat_flash_attn_fwd_q): tests, C5 A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_fp8.py tests/test_abi.py -m gpu -q -p no:cacheprovider > $O/r6ai_tests.log 2>&1; tail -3 $O/r6ai_tests.log
timeout 600 python -m pytest tests/test_ops.py -m gpu -q -p no:cacheprovider -k "flash or attention" > $O/r6ai_tests_ops.log 2>&1; tail -2 $O/r6ai_tests_ops.log
for v in 1 0 1 0; do
  echo "c5 flash_q8=$v $(COMAT_FP8_FLASH_Q8=$v timeout 700 python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing 2>$O/r6ai_c5_$v.err | grep -o '"ms_per_step": [0-9.]*' | head -1)" | tee -a $O/r6ai_c5_ab.txt
done
echo done
