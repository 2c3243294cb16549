#!/bin/bash
# (branch next/xcd-orders) XCD-aware orders, both bit-identical to the defaults: option g2_order (pipelined GEMM / conv: tile order
# inside an XCD chunk chosen by the L2-miss model) and flash_xcd (fused attention: one (batch, head) per XCD).
#   1. the bit-identity tests and the op tests they sit next to
#   2. C2 step with defaults / g2_order=2 / flash_xcd=1 / both
#   3. FETCH_SIZE of an eager step with and without them (separate --pmc passes)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
timeout 300 python -m pytest tests/test_ops.py -m gpu -q -x -p no:cacheprovider -k "tile_order or xcd or gemm2 or flash" > $O/r4x_tests.log 2>&1; tail -3 $O/r4x_tests.log
for v in "" "COMAT_G2_ORDER=2" "COMAT_FLASH_XCD=1" "COMAT_G2_ORDER=2 COMAT_FLASH_XCD=1"; do
  echo "== C2 step, ${v:-defaults}"
  env $v COMAT_SECONDARY=0 COMAT_PROBE_EAGER=0 COMAT_STEP_MODE=graph timeout 120 python bench.py --steps 8 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1
done
for v in "" "COMAT_G2_ORDER=2 COMAT_FLASH_XCD=1"; do
  tag=$([ -z "$v" ] && echo base || echo xcd)
  (cd /tmp && env $v COMAT_STEP_MODE=eager COMAT_PROBE_EAGER=0 timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_$tag -o s -- python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing > "$GRAFT_REPO_ROOT/$O/r4x_pmc_$tag.log" 2>&1)
  python tools/pmc_to_json.py --step $(find /tmp/pmc_$tag -name "*_results.db") > $O/r4x_pmc_$tag.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("$O/r4x_pmc_$tag.json")).get("step_families", {})
for k in ("gemm2_kernel", "flash_fwd2_kernel", "flash_dq2_kernel", "flash_dkdv2_kernel", "flash_dkdv_kernel"):
    if k in d: print("$tag", k, d[k].get("launches"), "launches", round(d[k].get("hbm_read_bytes_per_launch", 0) / 1e6, 1), "MB read / launch")
PY
done
echo done
