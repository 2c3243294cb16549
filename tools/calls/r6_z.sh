#!/bin/bash
# round 6: re-tune the plan table on the build with the staged epilogue (candidates now include g2_cfg 12 = 256x256 and 13 = 256x128 / 4 waves),
# after the parity tests of the new shape
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_ops.py -m gpu -q -p no:cacheprovider -k "gemm2" > $O/r6z_tests.log 2>&1; tail -3 $O/r6z_tests.log
for c in ${*:-c2}; do
  timeout 2400 python tools/tune_gemm2.py $c > $O/r6z_g2_tune_$c.jsonl 2> $O/r6z_g2_tune_$c.err; tail -2 $O/r6z_g2_tune_$c.err; wc -l $O/r6z_g2_tune_$c.jsonl
done
echo done
