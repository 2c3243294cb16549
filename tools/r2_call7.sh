#!/bin/bash
# round 2, GPU call 7: full GPU suite (bf16 error report, fp8 slice), default bench with back-to-back kernel replays,
# whole-step PMC passes (HBM traffic / MFMA busy of the dominant kernel), C4 / C3 benches, kernel trace, fp8 microbench, C5
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
rm -f $O/r2g_*
echo "== tests"; COMAT_TEST_REPORT=$PWD/$O/r2g_bf16_errors.txt timeout 1100 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -15 > $O/r2g_tests.log; tail -4 $O/r2g_tests.log
echo "== bench default"; COMAT_BENCH_DUMP=$O/r2g_bench_shapes.txt timeout 600 python bench.py > $O/r2g_bench_default.log 2>&1; tail -c 1500 $O/r2g_bench_default.log
echo "== pmc step"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -c1-5 | tr -d ' ')
  (cd /tmp && COMAT_STEP_GRAPH=0 timeout 420 rocprofv3 --pmc $pass -d /tmp/pmc_step_$tag -o s -- python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing > "$GRAFT_REPO_ROOT/$O/r2g_pmc_step_$tag.log" 2>&1)
  tail -c 300 $O/r2g_pmc_step_$tag.log
done
echo "== pmc targets"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -c1-5 | tr -d ' ')
  (cd /tmp && timeout 200 rocprofv3 --pmc $pass -d /tmp/pmc_tgt_$tag -o t -- python "$GRAFT_REPO_ROOT/tools/pmc_targets.py" > "$GRAFT_REPO_ROOT/$O/r2g_pmc_tgt_$tag.log" 2>&1)
done
python tools/pmc_to_json.py $(find /tmp/pmc_tgt_* -name "*_results.db") --step $(find /tmp/pmc_step_* -name "*_results.db") > $O/r2g_pmc_kernels.json 2> $O/r2g_pmc_to_json.err; tail -c 1200 $O/r2g_pmc_kernels.json; tail -3 $O/r2g_pmc_to_json.err
echo "== c4"; timeout 500 python bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline > $O/r2g_bench_c4.log 2>&1; tail -c 1200 $O/r2g_bench_c4.log
echo "== c3"; timeout 400 python bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline > $O/r2g_bench_c3.log 2>&1; tail -c 1200 $O/r2g_bench_c3.log
echo "== kernel trace"
(cd /tmp && COMAT_STEP_GRAPH=0 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$GRAFT_REPO_ROOT/$O/r2g_kt_bench.log" 2>&1)
tail -c 600 $O/r2g_kt_bench.log
f=$(find /tmp/kt -name "*_results.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py "$f" 4 | head -90 > $O/r2g_kernel_trace_eager.txt; head -12 $O/r2g_kernel_trace_eager.txt
echo "== mb fp8"; timeout 300 python tools/mb_fp8.py > $O/r2g_mb_fp8.txt 2>&1; cat $O/r2g_mb_fp8.txt
echo "== c5"; timeout 600 python bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/r2g_bench_c5.log 2>&1; tail -c 1500 $O/r2g_bench_c5.log
echo done
