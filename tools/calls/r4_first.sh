#!/bin/bash
# First GPU call of the next round (prepared at the end of round 3, when the GPU budget was spent):
#   1. re-tune the plan table of the pipelined GEMM / conv kernel: since call 18 gemm2.hip is built with MFMA results in VGPRs,
#      which moved the 128x128 4-wave variants from one to two waves per SIMD - the table in gemm2_plans.inc was measured
#      before that (tools/tune_gemm2.py -> tools/make_gemm2_plans.py; rebuild, then `r3_final.sh tests bench`)
#   2. C4 / C5 on the current build (their last numbers predate the attention work of calls 16-19)
#   3. in-step A/B of the attention options that were chosen from microbenchmarks only
#   bash tools/calls/r4_first.sh [tune|c4|c5|flash ...]   (default: all; ~12 GPU-minutes)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
WHAT="${*:-tune c4 c5 flash}"
for w in $WHAT; do case $w in
tune)
  echo "== plan tuning (C2 problems)"; timeout 600 python tools/tune_gemm2.py c2 > $O/r4a_g2_tune.jsonl 2> $O/r4a_g2_tune.err; wc -l $O/r4a_g2_tune.jsonl; tail -2 $O/r4a_g2_tune.err ;;
c4)
  echo "== bench c4"; timeout 400 python bench.py --config c4 --no-cpu-baseline --no-kernel-timing > $O/r4a_bench_c4.log 2>&1; grep -o '"ms_per_step": [0-9.]*\|"launch_mode": "[^"]*"' $O/r4a_bench_c4.log | head -2 ;;
c5)
  echo "== bench c5"; COMAT_BENCH_DUMP=$O/r4a_bench_c5_shapes.txt timeout 500 python bench.py --config c5 --no-cpu-baseline > $O/r4a_bench_c5.log 2>&1; grep -o '"ms_per_step": [0-9.]*' $O/r4a_bench_c5.log | head -1 ;;
flash)
  for v in "COMAT_FLASH_MERGE=0" "COMAT_FLASH_MERGE=2" "COMAT_FLASH_KT=3" "COMAT_FLASH_KT=5" ""; do
    echo "== C2 step, ${v:-defaults}"
    env $v COMAT_SECONDARY=0 COMAT_PROBE_EAGER=0 COMAT_STEP_MODE=graph timeout 120 python bench.py --steps 8 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1
  done ;;
esac; done
echo done
