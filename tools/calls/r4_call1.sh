#!/bin/bash
# round 4, call 1: land and measure what round 3 left unmeasured, and look INSIDE the pipelined kernel
#   fixtures : reference-code fixtures on the hip backend (merged next/hip-fixtures) + XCD-order bit-identity tests + batch capture test
#   diag     : timing table of the step's dominant GEMM / conv problems under g2_order and every block shape
#   pmc      : SQ / cache counter passes over the same problems (what a 128x128 block waits for)
#   ab       : C2 step with defaults / g2_order=2 / flash_xcd=1 / both
#   fetch    : FETCH_SIZE of an eager step, defaults vs both XCD options
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
WHAT="${*:-fixtures diag pmc ab fetch}"
for w in $WHAT; do case $w in
fixtures)
  echo "== reference-code fixtures on hip + xcd identity + batch capture"
  timeout 420 python -m pytest tests/test_models.py tests/test_losses.py tests/test_step.py tests/test_ops.py -m gpu -q -p no:cacheprovider \
    -k "reference or third_party or tile_order or xcd or capture_with_a_real_batch" > $O/r4a_fixtures.log 2>&1; tail -6 $O/r4a_fixtures.log ;;
diag)
  echo "== g2 diag timing"; timeout 300 python tools/probes/g2_diag.py > $O/r4a_g2_diag.txt 2> $O/r4a_g2_diag.err; cat $O/r4a_g2_diag.txt; tail -2 $O/r4a_g2_diag.err ;;
pmc)
  echo "== g2 diag counters"
  i=0
  for pass in "SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
              "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LEVEL_WAVES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE" \
              "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" \
              "FETCH_SIZE"; do
    i=$((i+1))
    (cd /tmp && G2_DIAG_PMC=1 timeout 200 rocprofv3 --pmc $pass -d /tmp/pmc_diag_$i -o d -- python "$GRAFT_REPO_ROOT/tools/probes/g2_diag.py" > "$GRAFT_REPO_ROOT/$O/r4a_pmc_diag_$i.log" 2>&1)
    tail -1 $O/r4a_pmc_diag_$i.log | cut -c1-200
  done
  python tools/pmc_dump.py $(find /tmp/pmc_diag_* -name "*_results.db") --match gemm2 > $O/r4a_pmc_diag.txt 2> $O/r4a_pmc_diag.err; wc -l $O/r4a_pmc_diag.txt; tail -2 $O/r4a_pmc_diag.err ;;
ab)
  for v in "" "COMAT_G2_ORDER=2" "COMAT_FLASH_XCD=1" "COMAT_G2_ORDER=2 COMAT_FLASH_XCD=1" "COMAT_G2_ORDER=1"; do
    echo "== C2 step, ${v:-defaults}"
    env $v COMAT_SECONDARY=0 COMAT_PROBE_EAGER=0 COMAT_STEP_MODE=graph timeout 150 python bench.py --steps 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1
  done ;;
fetch)
  for v in "" "COMAT_G2_ORDER=2 COMAT_FLASH_XCD=1"; do
    tag=$([ -z "$v" ] && echo base || echo xcd)
    (cd /tmp && env $v COMAT_STEP_MODE=eager COMAT_PROBE_EAGER=0 COMAT_SECONDARY=0 timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_$tag -o s -- python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing > "$GRAFT_REPO_ROOT/$O/r4a_pmc_$tag.log" 2>&1)
    python tools/pmc_to_json.py --step $(find /tmp/pmc_$tag -name "*_results.db") > $O/r4a_pmc_$tag.json 2>/dev/null
    python - <<PY
import json
d = json.load(open("$O/r4a_pmc_$tag.json")).get("step_families", {})
for k in sorted(d):
    if "gemm2" in k or "flash" in k: print("$tag", k, d[k].get("launches"), "launches", round(d[k].get("hbm_read_bytes_per_launch", 0) / 1e6, 1), "MB read / launch", d[k].get("avg_us"), "us")
PY
  done ;;
esac; done
echo done
