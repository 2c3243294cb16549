#!/bin/bash
# round 2, GPU call 14: split step graph (forward + backward captured, exchange + optimizer eager: the data-parallel form)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
echo "== tests"; timeout 400 python -m pytest tests/test_step.py -m gpu -q -p no:cacheprovider 2>&1 | tail -4
echo "== bench split graph"; COMAT_GRAPH_SPLIT=1 COMAT_STEP_GRAPH=1 timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing > $O/r2n_bench_split.log 2>&1; grep -o '"ms_per_step": [0-9.]*\|"launch_mode": "[^"]*"' $O/r2n_bench_split.log
echo "== bench whole graph"; COMAT_STEP_GRAPH=1 timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing > $O/r2n_bench_whole.log 2>&1; grep -o '"ms_per_step": [0-9.]*\|"launch_mode": "[^"]*"' $O/r2n_bench_whole.log
echo done
