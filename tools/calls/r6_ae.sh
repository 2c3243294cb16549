#!/bin/bash
# round 6: fp8 forward - tests on the final kernels, C5 A/B of the k-tail / batched groups and of the GEGLU bytes, plans for the C5 problems
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_fp8.py tests/test_abi.py -m gpu -q -p no:cacheprovider > $O/r6ae_tests.log 2>&1; tail -3 $O/r6ae_tests.log
timeout 600 python -m pytest tests/test_ops.py -m gpu -q -p no:cacheprovider -k "gemm2 or geglu" > $O/r6ae_tests_ops.log 2>&1; tail -2 $O/r6ae_tests_ops.log
for v in "1 1" "0 1" "1 0" "1 1" "0 1"; do
  set -- $v
  echo "c5 ktail=$1 geglu_q8=$2 $(COMAT_FP8_KTAIL=$1 COMAT_FP8_GEGLU_Q8=$2 timeout 700 python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing 2>$O/r6ae_c5_$1$2.err | grep -o '"ms_per_step": [0-9.]*' | head -1)" | tee -a $O/r6ae_c5_ab.txt
done
timeout 1500 python tools/tune_gemm2.py c5 > $O/r6ae_g2_tune_c5.jsonl 2> $O/r6ae_g2_tune_c5.err; tail -2 $O/r6ae_g2_tune_c5.err; wc -l $O/r6ae_g2_tune_c5.jsonl
echo done
