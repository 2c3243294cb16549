#!/bin/bash
# round 4, call 4: is the current tree slower than the tree of call 1?  Same box, alternating runs: _old (commit 128f112, COMAT_G2_ORDER=2) vs HEAD
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
bench() { (cd $1 && shift && env "$@" COMAT_SECONDARY=0 COMAT_PROBE_EAGER=0 COMAT_STEP_MODE=graph timeout 200 python bench.py --steps 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1); }
for i in 1 2; do
  echo "== old tree (128f112), COMAT_G2_ORDER=2"; bench _old COMAT_G2_ORDER=2
  echo "== HEAD"; bench . A=1
done
echo "== HEAD, COMAT_LORA_CHAIN=1 (library falls back to two launches)"; bench . COMAT_LORA_CHAIN=1
echo "== HEAD, COMAT_FLASH_XCD=1"; bench . COMAT_FLASH_XCD=1
echo done
