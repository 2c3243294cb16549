#!/bin/bash
# round 5, call 1: the three structural changes of the round on a GPU for the first time -
#   merged-weight trained LoRA calls (comat_lora_merge, COMAT_TRAIN_MERGED), the cooperative one-launch GroupNorm (norm_fused = 4),
#   the in-block key split of the forward attention (flash_ks, written in round 4) - parity first, then same-box A/B of the C2 step.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out; mkdir -p $O
echo "== parity (ops touched this round)"
timeout 500 python -m pytest tests/test_ops.py -m gpu -q -p no:cacheprovider -x \
  -k "lora or groupnorm or geglu or key_split or merged or abi" 2>&1 | tail -5
echo "== parity (step / models / segments)"
timeout 600 python -m pytest tests/test_step.py tests/test_models.py tests/test_segments.py tests/test_sdxl.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -5
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== GroupNorm forms"; timeout 300 python tools/mb_gn.py > $O/r5a_mb_gn.txt 2>&1; tail -17 $O/r5a_mb_gn.txt
for ks in 0 2; do
  COMAT_FLASH_KS=$ks timeout 240 python tools/mb_flash_ab.py > $O/r5a_mb_flash_ks$ks.txt 2>&1
  echo "== flash_ks=$ks"; tail -1 $O/r5a_mb_flash_ks$ks.txt; grep "Nq=4096 Nk=4096\|Nq=1024 Nk=1024\|d= 64" $O/r5a_mb_flash_ks$ks.txt | cut -c1-110
done
run() {  # label, env...
  local label=$1; shift
  echo "== C2 step: $label"
  env "$@" COMAT_SECONDARY=0 timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"launch_mode": "[^"]*"' | tr '\n' ' '; echo
}
{
run "round-4 forms (low-rank trained calls, three-launch GroupNorm)" COMAT_TRAIN_MERGED=0 COMAT_NORM_FUSED=3
run "merged trained calls only" COMAT_TRAIN_MERGED=1 COMAT_NORM_FUSED=3
run "cooperative GroupNorm only" COMAT_TRAIN_MERGED=0 COMAT_NORM_FUSED=4
run "both (default)" COMAT_TRAIN_MERGED=1 COMAT_NORM_FUSED=4
run "both + flash_ks=1" COMAT_TRAIN_MERGED=1 COMAT_NORM_FUSED=4 COMAT_FLASH_KS=1
run "round-4 forms again" COMAT_TRAIN_MERGED=0 COMAT_NORM_FUSED=3
run "both (default) again" COMAT_TRAIN_MERGED=1 COMAT_NORM_FUSED=4
} 2>&1 | tee $O/r5a_c2_ab.txt
echo done
