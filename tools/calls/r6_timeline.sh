#!/bin/bash
# round 6: kernel families / launch counts of the GRAPH-REPLAYED C2 step (rocprofv3 kernel trace of the default launch mode, steady-state
# 40 % of a 30-step run; tools/rocpd_timeline.py).  Under the tracer the queues do not overlap: the file shows how the kernels' durations split
# over the executor's queues and families, not the overlap of a real step.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out; mkdir -p $O
(cd /tmp && COMAT_SECONDARY=0 COMAT_PROBE_EAGER=0 COMAT_STEP_MODE=graph timeout 500 rocprofv3 --kernel-trace -d /tmp/ktg -o ktg -- python "$GRAFT_REPO_ROOT/bench.py" --steps 30 --warmup 2 --no-cpu-baseline --no-kernel-timing > "$GRAFT_REPO_ROOT/$O/r6z_bench_graph_traced.log" 2>&1)
tail -1 $O/r6z_bench_graph_traced.log | grep -o '"ms_per_step": [0-9.]*'
python tools/rocpd_timeline.py $(find /tmp/ktg -name "*_results.db" | head -1) 0.6 > $O/r6z_timeline_graph_c2.txt 2>&1; sed -n 1,6p $O/r6z_timeline_graph_c2.txt | cut -c1-200; grep -n "own " $O/r6z_timeline_graph_c2.txt | head -30 | cut -c1-120
echo done
