#!/bin/bash
# round 2, GPU call 12: default bench with the event-pair calibration, and the counter passes on the final build
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
rm -f $O/r2l_*
echo "== bench default"; COMAT_BENCH_DUMP=$O/r2l_bench_shapes.txt timeout 600 python bench.py > $O/r2l_bench_default.log 2>&1; tail -c 2200 $O/r2l_bench_default.log | head -c 1500
echo "== pmc step"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -c1-5 | tr -d ' ')
  (cd /tmp && COMAT_STEP_GRAPH=0 timeout 300 rocprofv3 --pmc $pass -d /tmp/pmc_step_$tag -o s -- python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing > "$GRAFT_REPO_ROOT/$O/r2l_pmc_step_$tag.log" 2>&1)
done
echo "== pmc targets"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -c1-5 | tr -d ' ')
  (cd /tmp && timeout 150 rocprofv3 --pmc $pass -d /tmp/pmc_tgt_$tag -o t -- python "$GRAFT_REPO_ROOT/tools/pmc_targets.py" > "$GRAFT_REPO_ROOT/$O/r2l_pmc_tgt_$tag.log" 2>&1)
done
python tools/pmc_to_json.py $(find /tmp/pmc_tgt_* -name "*_results.db") --step $(find /tmp/pmc_step_* -name "*_results.db") > $O/r2l_pmc_kernels.json 2> $O/r2l_pmc_to_json.err; tail -c 900 $O/r2l_pmc_kernels.json; tail -3 $O/r2l_pmc_to_json.err
echo done
