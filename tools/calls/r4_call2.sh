#!/bin/bash
# round 4, call 2: the lean kernel (gemm3.hip) and chained launches: parity tests, microbenchmarks; the hip-fixture tests again
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
WHAT="${*:-tests mb fixtures}"
for w in $WHAT; do case $w in
tests)
  echo "== lean kernel / chain tests"
  timeout 600 python -m pytest tests/test_ops.py -m gpu -q -p no:cacheprovider -k "gemm3 or chain or tile_order or lora" > $O/r4b_tests.log 2>&1; tail -8 $O/r4b_tests.log ;;
mb)
  echo "== mb_gemm3"; timeout 600 python tools/mb_gemm3.py > $O/r4b_mb_gemm3.txt 2> $O/r4b_mb_gemm3.err; cat $O/r4b_mb_gemm3.txt; tail -3 $O/r4b_mb_gemm3.err ;;
fixtures)
  echo "== reference-code fixtures on hip"
  timeout 420 python -m pytest tests/test_models.py tests/test_losses.py tests/test_step.py -m gpu -q -p no:cacheprovider \
    -k "reference or third_party" > $O/r4b_fixtures.log 2>&1; tail -6 $O/r4b_fixtures.log ;;
esac; done
echo done
