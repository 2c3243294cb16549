#!/bin/bash
# round 3, GPU call 2: grouped k-major weight gradients - op tests, microbenchmark, step tests, default bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
echo "== op tests"; timeout 300 python -m pytest tests/test_ops.py -m gpu -q -x -p no:cacheprovider -k "tt_grouped or weight_gradient_queue or lora_group" 2>&1 | tail -5
echo "== microbenchmark"; timeout 200 python tools/mb_tt_group.py > $O/r3b_mb_tt_group.txt 2>&1; cat $O/r3b_mb_tt_group.txt
echo "== step tests"; timeout 600 python -m pytest tests/test_step.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5
echo "== bench default"; COMAT_BENCH_DUMP=$O/r3b_bench_shapes.txt timeout 600 python bench.py --no-cpu-baseline > $O/r3b_bench_default.log 2>&1; tail -c 3000 $O/r3b_bench_default.log | head -c 1300
echo; echo "== bench ungrouped"; COMAT_TT_GROUPED=0 timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing > $O/r3b_bench_ungrouped.log 2>&1; grep -o '"ms_per_step": [0-9.]*' $O/r3b_bench_ungrouped.log
echo done
