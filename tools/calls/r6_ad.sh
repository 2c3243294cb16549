#!/bin/bash
# round 6: the bf16 k-tail / batched q-k-v of the fp8 forward: tests, C5 A/B (COMAT_FP8_KTAIL 1 / 0), then plans for the C5 (fp8 kinds),
# C4 and batch-4 problems on this build
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_fp8.py tests/test_abi.py -m gpu -q -p no:cacheprovider > $O/r6ad_tests.log 2>&1; tail -3 $O/r6ad_tests.log
for kt in 1 0 1 0; do
  echo "c5 ktail=$kt $(COMAT_FP8_KTAIL=$kt timeout 700 python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing 2>$O/r6ad_c5_$kt.err | grep -o '"ms_per_step": [0-9.]*' | head -1)" | tee -a $O/r6ad_c5_ab.txt
done
for c in c5 c4 c2bs4; do
  timeout 1500 python tools/tune_gemm2.py $c > $O/r6ad_g2_tune_$c.jsonl 2> $O/r6ad_g2_tune_$c.err; tail -2 $O/r6ad_g2_tune_$c.err; wc -l $O/r6ad_g2_tune_$c.jsonl
done
echo done
