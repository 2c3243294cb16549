#!/bin/bash
# round 4, call 14: attention, VALU diet on top of `w2` (call 12), same-box A/B of builds of attention.hip:
#   ft   = w2 + COMAT_FLASH_FULL_TILES (buffer loads with scalar tile offsets for full tiles: no per-iteration bounds / zeroing)
#   ftso = ft + COMAT_FLASH_SCALE_OUT (dS without the softmax scale; dQ / dK scaled when stored)
#   all  = ftso + COMAT_FLASH_SUM_MFMA (forward: denominator from a ones column of V through the P V product)
#   alle = all + COMAT_FLASH_EARLY_TR
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out; mkdir -p $O
AB=$GRAFT_REPO_ROOT/comat_amd/lib/ab
for v in base w2 ft ftso all alle; do
  c=""; [ $v = all -o $v = base ] && c="--check"
  COMAT_LIB_PATH=$AB/libcomat_$v.so timeout 240 python tools/mb_flash_ab.py $c > $O/r4n_mb_flash_$v.txt 2>&1
  echo "== $v"; tail -1 $O/r4n_mb_flash_$v.txt; grep "Nq=4096 Nk=4096 d= 40\|Nq=1024 Nk=1024 d= 80\|Nq= 256 Nk= 256\|d= 64" $O/r4n_mb_flash_$v.txt | cut -c1-200
done
echo "== attention tests on all"
COMAT_LIB_PATH=$AB/libcomat_all.so timeout 400 python -m pytest tests/test_ops.py tests/test_fullsize.py -m gpu -q -p no:cacheprovider -k "flash or attention" > $O/r4n_test_all.log 2>&1; tail -12 $O/r4n_test_all.log
echo done
