#!/bin/bash
# round 2, GPU call 8: pipelined split-K combine + new k-major split rule, double-buffered fused attention; tests, split
# sweeps, flash microbench, default bench (in-step + replayed kernel times), C4 with / without merged no-grad weights
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
rm -f $O/r2h_*
echo "== tests"; COMAT_TEST_REPORT=$PWD/$O/r2h_bf16_errors.txt timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -12 > $O/r2h_tests.log; tail -5 $O/r2h_tests.log
echo "== mb tt"; timeout 300 python tools/mb_tt.py > $O/r2h_mb_tt.txt 2>&1; cat $O/r2h_mb_tt.txt
echo "== mb flash"; timeout 200 python tools/mb_flash.py > $O/r2h_mb_flash.txt 2>&1; tail -12 $O/r2h_mb_flash.txt
echo "== bench default"; COMAT_BENCH_DUMP=$O/r2h_bench_shapes.txt timeout 600 python bench.py > $O/r2h_bench_default.log 2>&1; tail -c 2500 $O/r2h_bench_default.log
echo "== bench eager"; COMAT_STEP_GRAPH=0 timeout 400 python bench.py --no-cpu-baseline --no-kernel-timing > $O/r2h_bench_eager.log 2>&1; grep -o '"ms_per_step": [0-9.]*' $O/r2h_bench_eager.log
echo "== c4"; timeout 500 python bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/r2h_bench_c4.log 2>&1; grep -o '"ms_per_step": [0-9.]*' $O/r2h_bench_c4.log
echo "== c4 merged"; COMAT_NOGRAD_MERGED=1 timeout 500 python bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/r2h_bench_c4_merged.log 2>&1; grep -o '"ms_per_step": [0-9.]*' $O/r2h_bench_c4_merged.log
echo "== c3 merged"; COMAT_NOGRAD_MERGED=1 timeout 400 python bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/r2h_bench_c3_merged.log 2>&1; grep -o '"ms_per_step": [0-9.]*' $O/r2h_bench_c3_merged.log
echo done
