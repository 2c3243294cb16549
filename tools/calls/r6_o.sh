#!/bin/bash
# round 6: what a CU can overlap - fragment reads / MFMAs / barrier in the k-loop's shapes, no global traffic (tools/probes/loop_probe.hip)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
mkdir -p $O
timeout 300 tools/probes/loop_probe.bin > $O/r6o_loop_probe.txt 2>&1
cat $O/r6o_loop_probe.txt
