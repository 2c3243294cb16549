#!/bin/bash
# round 6: the query split of the cross-attention backward (option flash_qs) x flash_merge, microbench; then C2 under the best settings
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_fp8.py tests/test_ops.py -m gpu -q -p no:cacheprovider -k 'groupnorm or flash or attention' > $O/r6ah_tests.log 2>&1; tail -3 $O/r6ah_tests.log
timeout 300 python tools/flash_bits.py > $O/r6ah_flash_bits.txt 2>&1; tail -3 $O/r6ah_flash_bits.txt
timeout 600 python tools/mb_flash.py qs 2>&1 | grep -v amdgpu.ids > $O/r6ah_mb_flash_qs.txt; cat $O/r6ah_mb_flash_qs.txt
for v in "512 1" "128 1" "256 1" "512 1" "128 1" "256 1"; do
  set -- $v
  echo "c2 flash_qs=$1 flash_merge=$2 $(COMAT_FLASH_QS=$1 COMAT_FLASH_MERGE=$2 COMAT_SECONDARY=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)" | tee -a $O/r6ah_c2_ab.txt
done
echo done
