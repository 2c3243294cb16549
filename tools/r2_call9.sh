#!/bin/bash
# round 2, GPU call 9: fused-attention changes (D inside the dQ kernel, transposed images by head dim) - targeted tests,
# default bench (in-step + replayed kernel times), eager bench, kernel trace of the eager bench, C4 / C3 with roofline
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
rm -f $O/r2i_*
echo "== tests (attention, step, models, sdxl)"; timeout 600 python -m pytest tests/test_ops.py tests/test_step.py tests/test_models.py tests/test_sdxl.py -m gpu -q -p no:cacheprovider -k "flash or attention or step or unet or sampler or sdxl" 2>&1 | tail -6 > $O/r2i_tests.log; tail -3 $O/r2i_tests.log
echo "== bench default"; COMAT_BENCH_DUMP=$O/r2i_bench_shapes.txt timeout 600 python bench.py > $O/r2i_bench_default.log 2>&1; tail -c 3000 $O/r2i_bench_default.log
echo "== bench eager"; COMAT_STEP_GRAPH=0 timeout 400 python bench.py --no-cpu-baseline --no-kernel-timing > $O/r2i_bench_eager.log 2>&1; grep -o '"ms_per_step": [0-9.]*' $O/r2i_bench_eager.log
echo "== kernel trace (eager bench)"
(cd /tmp && COMAT_STEP_GRAPH=0 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > "$GRAFT_REPO_ROOT/$O/r2i_kt_bench.log" 2>&1)
f=$(find /tmp/kt -name "*_results.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py "$f" 3 > $O/r2i_kernel_trace_eager.txt; sed -n '/by kernel family/,$p' $O/r2i_kernel_trace_eager.txt | head -16
echo "== c4"; timeout 500 python bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline > $O/r2i_bench_c4.log 2>&1; grep -o '"ms_per_step": [0-9.]*' $O/r2i_bench_c4.log
echo "== c3"; timeout 400 python bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline > $O/r2i_bench_c3.log 2>&1; grep -o '"ms_per_step": [0-9.]*' $O/r2i_bench_c3.log
echo done
