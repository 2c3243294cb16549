#!/bin/bash
# round 6: where the C2 step's time is on the current tree - per-problem dump of the default line (no secondaries), gap table,
# kernel trace of three eager steps
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
COMAT_SECONDARY=0 COMAT_BENCH_DUMP=$O/r6x_bench_shapes.txt timeout 900 python bench.py --no-cpu-baseline > $O/r6x_bench.log 2>&1
tail -c 3000 $O/r6x_bench.log
python tools/shape_gaps.py $O/r6x_bench_shapes.txt --top 60 > $O/r6x_gap_table.txt 2>&1; head -9 $O/r6x_gap_table.txt
(cd /tmp && COMAT_STEP_MODE=eager COMAT_PROBE_EAGER=0 COMAT_SECONDARY=0 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 0 --no-cpu-baseline --no-kernel-timing > "$GRAFT_REPO_ROOT/$O/r6x_bench_traced.log" 2>&1)
python tools/rocpd_summary.py $(find /tmp/kt -name "*_results.db" | head -1) 3 > $O/r6x_kernel_trace_eager.txt 2>&1; head -40 $O/r6x_kernel_trace_eager.txt | cut -c1-160
echo done
