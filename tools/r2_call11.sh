#!/bin/bash
# round 2, final GPU call: the whole GPU suite, smoke(), default bench, kernel trace of the eager bench, C3 / C4 / C5
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
rm -f $O/r2k_*
echo "== gate: block shapes of the pipelined kernel (incl. 128-byte k-tiles)"
if ! timeout 400 python -m pytest tests/test_ops.py -m gpu -q -x -p no:cacheprovider -k "gemm2 or split_k or k_major" > $O/r2k_gate.log 2>&1; then tail -15 $O/r2k_gate.log; echo "GATE FAILED"; exit 0; fi
tail -1 $O/r2k_gate.log
echo "== tests"; COMAT_TEST_REPORT=$PWD/$O/r2k_bf16_errors.txt timeout 1000 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -8 > $O/r2k_tests.log; tail -3 $O/r2k_tests.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== bench default"; COMAT_BENCH_DUMP=$O/r2k_bench_shapes.txt timeout 600 python bench.py > $O/r2k_bench_default.log 2>&1; tail -c 2600 $O/r2k_bench_default.log
echo "== kernel trace (eager bench)"
(cd /tmp && COMAT_STEP_GRAPH=0 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > "$GRAFT_REPO_ROOT/$O/r2k_kt_bench.log" 2>&1)
f=$(find /tmp/kt -name "*_results.db" | head -1); [ -n "$f" ] && python tools/rocpd_summary.py "$f" 3 > $O/r2k_kernel_trace_eager.txt; sed -n '/by kernel family/,$p' $O/r2k_kernel_trace_eager.txt | head -12
echo "== c4"; timeout 500 python bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline > $O/r2k_bench_c4.log 2>&1; grep -o '"ms_per_step": [0-9.]*' $O/r2k_bench_c4.log
echo "== c3"; timeout 400 python bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline > $O/r2k_bench_c3.log 2>&1; grep -o '"ms_per_step": [0-9.]*' $O/r2k_bench_c3.log
echo "== c5"; timeout 500 python bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/r2k_bench_c5.log 2>&1; grep -o '"ms_per_step": [0-9.]*' $O/r2k_bench_c5.log
echo done
