#!/bin/bash
# round 4, call 17: the new attention guard test, the default bench line on the final build WITH its counter figures
# (profiles/r04_pmc_kernels.json now belongs to this build), C4 standalone
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests/test_ops.py -m gpu -q -p no:cacheprovider -k "descriptor_range or flash_attention" 2>&1 | tail -3
echo "== bench default"; COMAT_BENCH_DUMP=$O/r4x_bench_shapes.txt timeout 900 python bench.py > $O/r4x_bench_default.log 2>&1; tail -c 16000 $O/r4x_bench_default.log | grep -o '"ms_per_step": [0-9.]*\|"value": [0-9.e-]*\|"traffic": [0-9a-z.]*\|"mfma_busy_frac": [0-9a-z.]*\|"frac": [0-9.]*' | head -12
echo "== bench c4"; timeout 500 python bench.py --config c4 --no-cpu-baseline > $O/r4x_bench_c4.log 2>&1; grep -o '"ms_per_step": [0-9.]*' $O/r4x_bench_c4.log | head -1
echo done
