#!/bin/bash
# round 5: where the graph-replayed C2 step's wall time goes - rocprofv3 kernel trace of the default launch mode (whole-step
# hipGraph), busy / idle / concurrency analysis over the steady-state half of the run (tools/rocpd_timeline.py)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out; mkdir -p $O
(cd /tmp && COMAT_SECONDARY=0 COMAT_PROBE_EAGER=0 COMAT_STEP_MODE=graph timeout 500 rocprofv3 --kernel-trace -d /tmp/ktg -o ktg -- python "$GRAFT_REPO_ROOT/bench.py" --steps 30 --warmup 2 --no-cpu-baseline --no-kernel-timing > "$GRAFT_REPO_ROOT/$O/r5z_bench_graph_traced.log" 2>&1)
tail -1 $O/r5z_bench_graph_traced.log | grep -o '"ms_per_step": [0-9.]*'
python tools/rocpd_timeline.py $(find /tmp/ktg -name "*_results.db" | head -1) 0.6 > $O/r5z_timeline_graph_c2.txt 2>&1; head -60 $O/r5z_timeline_graph_c2.txt | cut -c1-160
echo done
