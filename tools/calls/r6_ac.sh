#!/bin/bash
# round 6: fp8 delayed scaling on the GPU: tests, C5 delayed vs just-in-time scales (alternating), kernel trace of one C5 step
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_fp8.py tests/test_abi.py -m gpu -q -p no:cacheprovider -s > $O/r6ac_tests.log 2>&1; grep "fp8 UNet" $O/r6ac_tests.log | tail -4; tail -3 $O/r6ac_tests.log
for mode in delayed jit delayed jit; do
  echo "c5 $mode $(COMAT_FP8_SCALING=$mode timeout 700 python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing 2>$O/r6ac_c5_$mode.err | grep -o '"ms_per_step": [0-9.]*' | head -1)" | tee -a $O/r6ac_c5_ab.txt
done
(cd /tmp && COMAT_PROBE_EAGER=0 timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt5 -o kt -- python "$GRAFT_REPO_ROOT/bench.py" --config c5 --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing > "$GRAFT_REPO_ROOT/$O/r6ac_c5_traced.log" 2>&1)
python tools/rocpd_summary.py $(find /tmp/kt5 -name "*_results.db" | head -1) 1 > $O/r6ac_c5_kernel_trace.txt 2>&1; head -30 $O/r6ac_c5_kernel_trace.txt | cut -c1-160
echo done
