#!/bin/bash
# round 4, call 15: attention, the policy of attention.hip (F1 = the defaults) against its alternatives on one box:
#   F2 = early transposed reads in every 2-tile forward, F3 = nowhere, F4 = also in the 2-tile backward kernels,
#   F5 = dS with the softmax scale inside (SCALE_OUT off), base = main before this work;
# then the attention parity tests on F1 and the C2 step, base against F1.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out; mkdir -p $O
AB=$GRAFT_REPO_ROOT/comat_amd/lib/ab
for v in base F1 F2 F3 F4 F5; do
  c=""; [ $v = F1 ] && c="--check"
  COMAT_LIB_PATH=$AB/libcomat_$v.so timeout 240 python tools/mb_flash_ab.py $c > $O/r4o_mb_flash_$v.txt 2>&1
  echo "== $v"; tail -1 $O/r4o_mb_flash_$v.txt; grep "Nq=4096 Nk=4096 d= 40\|Nq=1024 Nk=1024 d= 80\|d= 64" $O/r4o_mb_flash_$v.txt | cut -c1-200
done
echo "== attention tests on F1 (the default library)"
timeout 400 python -m pytest tests/test_ops.py tests/test_fullsize.py -m gpu -q -p no:cacheprovider -k "flash or attention" > $O/r4o_test_F1.log 2>&1; tail -4 $O/r4o_test_F1.log
for v in base F1 base F1; do
  echo "== C2 step on $v"
  COMAT_LIB_PATH=$AB/libcomat_$v.so COMAT_SECONDARY=0 timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 2>&1 | grep -o '"ms_per_step": [0-9.]*' | head -1
done | tee $O/r4o_c2_ab.txt
echo done
