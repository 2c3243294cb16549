#!/bin/bash
# First GPU call of round 5 (prepared at the end of round 4, nothing in it has run on a GPU yet): the in-block key split of the
# 2-tile forward attention (branch next/flash-splitk: flash_fwd2s_kernel, option flash_ks, off by default).
#   1. its parity test (split against unsplit against an fp32 reference; even / odd pair counts, ragged edges, reproducibility)
#      and the attention tests of the suite;
#   2. the step's attention problems with flash_ks = 0 / 1 / 2 on one box (tools/mb_flash_ab.py, one process per setting);
#   3. the C2 step with flash_ks = 0 and 1, alternating.
# Expectation from profiles/r04_q_flash_occupancy.txt: forward 2 x 8 x 4096^2 d=40 ~101 -> ~85 us (four waves per SIMD instead
# of two), 1 x 8: ~65 -> ~48; small grids (BLIP 577^2, 1024^2 at d=64) gain the most.  If it holds: flash_ks = 1 by default.
# Branch next/dkdv-load-order (dK/dV: lse / D loads before the tile loads, measured -1 % / -4 %) merges independently.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out; mkdir -p $O
echo "== parity"; timeout 400 python -m pytest tests/test_ops.py -m gpu -q -p no:cacheprovider -k "key_split or flash or attention" 2>&1 | tail -4
for ks in 0 1 2; do
  COMAT_FLASH_KS=$ks timeout 240 python tools/mb_flash_ab.py > $O/r5a_mb_flash_ks$ks.txt 2>&1
  echo "== flash_ks=$ks"; tail -1 $O/r5a_mb_flash_ks$ks.txt; grep "Nq=4096 Nk=4096\|Nq=1024 Nk=1024\|d= 64" $O/r5a_mb_flash_ks$ks.txt | cut -c1-100
done
for ks in 0 1 0 1; do
  echo "== C2 step, flash_ks=$ks"
  COMAT_FLASH_KS=$ks COMAT_SECONDARY=0 timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 2>&1 | grep -o '"ms_per_step": [0-9.]*' | head -1
done | tee $O/r5a_c2_ks_ab.txt
echo done
