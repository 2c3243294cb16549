#!/bin/bash
# round 4, call 5: GEGLU in the GEMM epilogue, the GroupNorm loops, the lean kernel's rule, flash_xcd = 1 by default:
# tests, GroupNorm microbenchmark, C2 against the tree of call 1 on the same box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
bench() { (cd $1 && shift && env "$@" COMAT_SECONDARY=0 COMAT_PROBE_EAGER=0 COMAT_STEP_MODE=graph timeout 200 python bench.py --steps 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1); }
WHAT="${*:-tests mbgn ab}"
for w in $WHAT; do case $w in
tests)
  echo "== op tests (geglu, norms, gemm3, attention options)"
  timeout 600 python -m pytest tests/test_ops.py -m gpu -q -p no:cacheprovider -k "geglu or norm or gemm3 or xcd or gemm_layouts" > $O/r4e_tests_ops.log 2>&1; tail -5 $O/r4e_tests_ops.log
  echo "== model / step tests"
  timeout 900 python -m pytest tests/test_models.py tests/test_step.py tests/test_blip.py tests/test_segments.py tests/test_fp8.py -m gpu -q -p no:cacheprovider -x > $O/r4e_tests_models.log 2>&1; tail -5 $O/r4e_tests_models.log ;;
mbgn)
  echo "== mb_gn"; timeout 300 python tools/mb_gn.py > $O/r4e_mb_gn.txt 2>&1; cat $O/r4e_mb_gn.txt ;;
ab)
  for i in 1 2; do
    echo "== old tree (128f112), COMAT_G2_ORDER=2"; bench _old COMAT_G2_ORDER=2
    echo "== HEAD"; bench . A=1
  done
  echo "== HEAD, COMAT_GEGLU_FUSED=0"; bench . COMAT_GEGLU_FUSED=0
  echo "== HEAD, COMAT_GEMM3=0"; bench . COMAT_GEMM3=0 ;;
esac; done
echo done
