#!/bin/bash
# round 5, after the final: (a) every MFMA problem of the bs-1 C2 step with its gap to its own roofline (the dump of the final call
# was overwritten by the batch-4 subprocess, which inherited COMAT_BENCH_DUMP: fixed), (b) busy / idle / concurrency analysis of the
# graph-replayed C2 step, (c) the default line once more, now quoting the counter passes committed as profiles/r05_pmc_kernels.json
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out; mkdir -p $O
echo "== shapes + gap table"; COMAT_SECONDARY=0 COMAT_PROBE_EAGER=0 COMAT_ATTN_MAP_PROBE=0 COMAT_BENCH_DUMP=$O/r5z_bench_shapes.txt timeout 400 python bench.py --no-cpu-baseline --steps 4 > $O/r5z_bench_shapes_run.log 2>&1
python tools/shape_gaps.py $O/r5z_bench_shapes.txt --top 40 > $O/r5z_gap_table.txt 2>&1; head -9 $O/r5z_gap_table.txt
echo "== timeline of the graph-replayed step"
(cd /tmp && COMAT_SECONDARY=0 COMAT_PROBE_EAGER=0 COMAT_STEP_MODE=graph timeout 500 rocprofv3 --kernel-trace -d /tmp/ktg -o ktg -- python "$GRAFT_REPO_ROOT/bench.py" --steps 30 --warmup 2 --no-cpu-baseline --no-kernel-timing > "$GRAFT_REPO_ROOT/$O/r5z_bench_graph_traced.log" 2>&1)
tail -1 $O/r5z_bench_graph_traced.log | grep -o '"ms_per_step": [0-9.]*'
python tools/rocpd_timeline.py $(find /tmp/ktg -name "*_results.db" | head -1) 0.6 > $O/r5z_timeline_graph_c2.txt 2>&1; head -45 $O/r5z_timeline_graph_c2.txt | cut -c1-160
echo "== bench default"; timeout 1200 python bench.py > $O/r5z_bench_default.log 2> $O/r5z_bench_default.err; echo rc=$?; grep "\[bench\]" $O/r5z_bench_default.err | tail -8
tail -c 30000 $O/r5z_bench_default.log | grep -o '"ms_per_step": [0-9.]*\|"value": [0-9.e-]*\|"traffic": [0-9a-z.]*\|"frac": [0-9.]*\|"mfma_busy_frac": [0-9.a-z]*' | head -12
echo done
