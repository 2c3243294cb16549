#!/bin/bash
# round 6: GroupNorm finalize-in-apply with a burst-load prologue (norm_fused = 5) against the default (3): parity, microbench, C2 / C3 A/B;
# then the merged-vs-low-rank training trajectory from U = 0
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_ops.py -m gpu -q -p no:cacheprovider -k "groupnorm" > $O/r6ag_tests.log 2>&1; tail -3 $O/r6ag_tests.log
timeout 300 python tools/mb_gn.py 2>&1 | grep -v amdgpu.ids > $O/r6ag_mb_gn.txt; cat $O/r6ag_mb_gn.txt
for nf in 3 5 3 5; do
  echo "c2 norm_fused=$nf $(COMAT_NORM_FUSED=$nf COMAT_SECONDARY=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)" | tee -a $O/r6ag_ab.txt
done
for nf in 3 5; do
  echo "c3 norm_fused=$nf $(COMAT_NORM_FUSED=$nf timeout 600 python bench.py --config c3 --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)" | tee -a $O/r6ag_ab.txt
done
timeout 600 python tools/merged_trajectory.py 8 2>&1 | grep -v amdgpu.ids > $O/r6ag_merged_trajectory.txt; cat $O/r6ag_merged_trajectory.txt
echo done
bash tools/calls/r6_final.sh c5
