#!/bin/bash
# round 5, call 3: the whole GPU suite after the D-split fix, the chain removal and the GEGLU backward epilogue; same-box A/B of
# the GEGLU backward epilogue (C2) and of the D split (C3, alternating); C4 on its own.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out; mkdir -p $O
echo "== GPU suite"; rm -f $O/r5c_bf16_errors.txt; COMAT_TEST_REPORT=$O/r5c_bf16_errors.txt timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/r5c_test_all.log 2>&1; tail -6 $O/r5c_test_all.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
run() {  # label, env...
  local label=$1; shift
  echo "== C2 step: $label"
  env "$@" COMAT_SECONDARY=0 COMAT_PROBE_EAGER=0 timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"launch_mode": "[^"]*"' | tr '\n' ' '; echo
}
{
run "GEGLU backward as its own launch" COMAT_GEGLU_BWD_FUSED=0
run "GEGLU backward in the dgrad epilogue (default)" COMAT_GEGLU_BWD_FUSED=1
run "GEGLU backward as its own launch" COMAT_GEGLU_BWD_FUSED=0
run "GEGLU backward in the dgrad epilogue (default)" COMAT_GEGLU_BWD_FUSED=1
} 2>&1 | tee $O/r5c_c2_geglu_bwd_ab.txt
for d in 0 1 0 1; do
  echo "== C3, COMAT_D_SPLIT=$d"
  COMAT_D_SPLIT=$d timeout 400 python bench.py --config c3 --no-cpu-baseline --no-kernel-timing --steps 4 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1
done | tee $O/r5c_c3_dsplit.txt
echo "== C4"; timeout 500 python bench.py --config c4 --no-cpu-baseline --no-kernel-timing --steps 4 --warmup 1 > $O/r5c_bench_c4.log 2>&1; grep -o '"ms_per_step": [0-9.]*' $O/r5c_bench_c4.log | head -1
echo done
