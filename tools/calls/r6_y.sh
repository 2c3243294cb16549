#!/bin/bash
# round 6: ring depth of the 128-byte-k-tile shapes (g2_cfg 13 - 16) on the UNet's latency-bound problems
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
timeout 900 python tools/mb_ring.py 2>&1 | grep -v amdgpu.ids > $O/r6y_mb_ring.txt
tail -5 $O/r6y_mb_ring.txt
echo done
