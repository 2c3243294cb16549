#!/bin/bash
# round 6, final evidence on HEAD: the whole GPU suite + smoke, the counter passes behind the roofline block (recorded with the
# library's build id and copied to profiles/ on the box so that the bench line of the SAME call quotes them), the default bench
# line (cpu_baseline, C3 / C4 / batch-4 secondaries), the kernel trace of three eager C2 steps, C5.
#   bash tools/calls/r6_final.sh [tests|pmc|bench|trace|c5 ...]   (default: all)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
WHAT="${*:-tests pmc bench trace c5}"
for w in $WHAT; do case $w in
tests)
  echo "== GPU suite"; rm -f $O/r6z_bf16_errors.txt; COMAT_TEST_REPORT=$O/r6z_bf16_errors.txt timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/r6z_test_all.log 2>&1; tail -6 $O/r6z_test_all.log
  echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ;;
pmc)
  echo "== pmc step passes"
  for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    tag=$(echo $pass | cut -c1-5 | tr -d ' ')
    (cd /tmp && COMAT_STEP_MODE=eager COMAT_PROBE_EAGER=0 COMAT_SECONDARY=0 timeout 300 rocprofv3 --pmc $pass -d /tmp/pmc_step_$tag -o s -- python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing > "$GRAFT_REPO_ROOT/$O/r6z_pmc_step_$tag.log" 2>&1)
  done
  echo "== pmc targets"
  for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    tag=$(echo $pass | cut -c1-5 | tr -d ' ')
    (cd /tmp && timeout 150 rocprofv3 --pmc $pass -d /tmp/pmc_tgt_$tag -o t -- python "$GRAFT_REPO_ROOT/tools/pmc_targets.py" > "$GRAFT_REPO_ROOT/$O/r6z_pmc_tgt_$tag.log" 2>&1)
  done
  python tools/pmc_to_json.py $(find /tmp/pmc_tgt_* -name "*_results.db") --step $(find /tmp/pmc_step_* -name "*_results.db") > $O/r6z_pmc_kernels.json 2> $O/r6z_pmc_to_json.err; tail -c 900 $O/r6z_pmc_kernels.json; tail -3 $O/r6z_pmc_to_json.err
  # the bench line below quotes the newest profiles/rNN_pmc_kernels.json whose build id matches the loaded library
  python -c "import json; d = json.load(open('$O/r6z_pmc_kernels.json')); assert d.get('build_id')" && cp $O/r6z_pmc_kernels.json profiles/r06_pmc_kernels.json ;;
bench)
  echo "== bench default"; COMAT_BENCH_DUMP=$O/r6z_bench_shapes.txt timeout 1200 python bench.py > $O/r6z_bench_default.log 2>&1; tail -c 20000 $O/r6z_bench_default.log | grep -o '"ms_per_step": [0-9.]*\|"value": [0-9.e-]*\|"launch_mode": "[^"]*"\|"probe_ms_per_step": {[^}]*}\|"eager_ms_per_step": [0-9.]*\|"cores": [0-9]*\|"traffic": [0-9a-z.]*\|"frac": [0-9.]*' | head -24
  python tools/shape_gaps.py $O/r6z_bench_shapes.txt --top 40 > $O/r6z_gap_table.txt 2>&1; head -9 $O/r6z_gap_table.txt ;;
trace)
  echo "== kernel trace (eager C2, 3 steps)"
  (cd /tmp && COMAT_STEP_MODE=eager COMAT_PROBE_EAGER=0 COMAT_SECONDARY=0 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 0 --no-cpu-baseline --no-kernel-timing > "$GRAFT_REPO_ROOT/$O/r6z_bench_traced.log" 2>&1)
  python tools/rocpd_summary.py $(find /tmp/kt -name "*_results.db" | head -1) 3 > $O/r6z_kernel_trace_eager.txt 2>&1; head -24 $O/r6z_kernel_trace_eager.txt | cut -c1-150 ;;
c5)
  echo "== bench c5 (with the per-kernel roofline block)"; COMAT_BENCH_DUMP=$O/r6z_bench_c5_shapes.txt timeout 500 python bench.py --config c5 --no-cpu-baseline > $O/r6z_bench_c5.log 2>&1; grep -o '"ms_per_step": [0-9.]*\|"launch_mode": "[^"]*"' $O/r6z_bench_c5.log | head -3 ;;
esac; done
echo done
