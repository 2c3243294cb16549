#!/bin/bash
# round 4, call 16: dK/dV kernel with the lse / D loads issued BEFORE the tile loads (H) against the default library (F1)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out; mkdir -p $O
AB=$GRAFT_REPO_ROOT/comat_amd/lib/ab
for v in F1 H F1 H; do
  COMAT_LIB_PATH=$AB/libcomat_$v.so timeout 240 python tools/mb_flash_ab.py > $O/r4p_mb_flash_$v.txt 2>&1
  echo "== $v"; tail -1 $O/r4p_mb_flash_$v.txt; grep "Nq=4096 Nk=4096\|Nq=1024 Nk=1024\|d= 64" $O/r4p_mb_flash_$v.txt | cut -c1-100
done
COMAT_LIB_PATH=$AB/libcomat_H.so timeout 400 python -m pytest tests/test_ops.py tests/test_fullsize.py -m gpu -q -p no:cacheprovider -k "flash or attention" 2>&1 | tail -2
echo done
