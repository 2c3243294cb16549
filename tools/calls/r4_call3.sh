#!/bin/bash
# round 4, call 3: chained launches on the pipelined kernel (tests, microbenchmark, C2 A/B), then the plan table re-tuned under
# the current build (VGPR-form MFMA results, g2_order = 2) - tuned, regenerated and rebuilt ON the box, C2 before / after
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
bench() { env "$@" COMAT_SECONDARY=0 COMAT_PROBE_EAGER=0 COMAT_STEP_MODE=graph timeout 200 python bench.py --steps 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1; }
WHAT="${*:-tests mb ab tune ab2}"
for w in $WHAT; do case $w in
tests)
  echo "== chain tests"
  timeout 600 python -m pytest tests/test_ops.py -m gpu -q -p no:cacheprovider -k "chain or lora" > $O/r4c_tests.log 2>&1; tail -8 $O/r4c_tests.log ;;
mb)
  echo "== mb_gemm3"; timeout 600 python tools/mb_gemm3.py > $O/r4c_mb_gemm3.txt 2> $O/r4c_mb_gemm3.err; grep -A40 "LoRA pairs" $O/r4c_mb_gemm3.txt | cut -c1-110; tail -3 $O/r4c_mb_gemm3.err ;;
ab)
  echo "== C2, old plan table: defaults"; bench A=1
  echo "== C2, old plan table: COMAT_GEMM2_CHAIN=1"; bench COMAT_GEMM2_CHAIN=1
  echo "== C2, old plan table: COMAT_GEMM2_CHAIN=1 COMAT_FLASH_XCD=1"; bench COMAT_GEMM2_CHAIN=1 COMAT_FLASH_XCD=1 ;;
tune)
  echo "== plan tuning (C2 problems)"; timeout 560 python tools/tune_gemm2.py c2 > $O/r4c_g2_tune.jsonl 2> $O/r4c_g2_tune.err; wc -l $O/r4c_g2_tune.jsonl; tail -2 $O/r4c_g2_tune.err
  cp comat_amd/csrc/gemm2_plans.inc /tmp/plans_old.inc
  python tools/make_gemm2_plans.py profiles/r02_e_g2_tune.jsonl profiles/r02_j_g2_tune.jsonl $O/r4c_g2_tune.jsonl | tail -1
  cp comat_amd/csrc/gemm2_plans.inc $O/r4c_gemm2_plans.inc
  make -C comat_amd/csrc -j8 > $O/r4c_rebuild.log 2>&1; tail -1 $O/r4c_rebuild.log ;;
ab2)
  echo "== C2, NEW plan table: defaults"; bench A=1
  echo "== C2, NEW plan table: COMAT_GEMM2_CHAIN=1"; bench COMAT_GEMM2_CHAIN=1
  echo "== C2, NEW plan table: defaults again"; bench A=1 ;;
esac; done
echo done
