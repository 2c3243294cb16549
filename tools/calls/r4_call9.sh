#!/bin/bash
# round 4, call 9: 3x3 convs with the input strip in LDS: parity tests, microbenchmark, C2 with / without
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
bench() { env "$@" COMAT_SECONDARY=0 COMAT_PROBE_EAGER=0 COMAT_STEP_MODE=graph timeout 200 python bench.py --steps 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1; }
echo "== strip conv tests"
timeout 600 python -m pytest tests/test_ops.py -m gpu -q -p no:cacheprovider -k "conv" > $O/r4g_tests.log 2>&1; tail -8 $O/r4g_tests.log
echo "== mb_conv_strip"; timeout 500 python tools/mb_conv_strip.py > $O/r4g_mb_conv_strip.txt 2>&1; cat $O/r4g_mb_conv_strip.txt
echo "== C2 defaults (g2_strip=0)"; bench A=1
echo "== C2 COMAT_G2_STRIP=1"; bench COMAT_G2_STRIP=1
echo "== C2 defaults again"; bench A=1
echo "== C2 COMAT_G2_STRIP=1 again"; bench COMAT_G2_STRIP=1
echo done
