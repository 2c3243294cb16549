#!/bin/bash
# round 3, GPU call 1: where does the wall time of a GRAPH-REPLAYED C2 step go?  (kernel trace of replayed steps ->
# tools/rocpd_timeline.py: per-queue busy / gaps, concurrency, wall-time ownership by kernel family)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
(cd /tmp && COMAT_STEP_GRAPH=1 timeout 400 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python "$GRAFT_REPO_ROOT/bench.py" --steps 6 --warmup 1 --no-cpu-baseline --no-kernel-timing > "$GRAFT_REPO_ROOT/$O/r3a_bench_graph_traced.log" 2>&1)
tail -c 1500 $O/r3a_bench_graph_traced.log
DB=$(find /tmp/kt -name "*_results.db" | head -1)
# the timed region is the last 6 of 9 steps (capture step, first replay, warm-up): 0.45 leaves 4-5 whole replays in the window
python tools/rocpd_timeline.py $DB 0.5 > $O/r3a_timeline_graph_c2.txt 2>&1
head -70 $O/r3a_timeline_graph_c2.txt
echo "== untraced default bench"
COMAT_BENCH_DUMP=$O/r3a_bench_shapes.txt timeout 600 python bench.py --no-cpu-baseline > $O/r3a_bench_default.log 2>&1; tail -c 3000 $O/r3a_bench_default.log | head -c 1200
echo done
