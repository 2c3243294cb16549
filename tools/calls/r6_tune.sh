#!/bin/bash
# round 6: re-tune the plan table of the pipelined kernel on the current build (C2 shapes; C4 with "c4" as argument)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
for c in ${*:-c2}; do
  timeout 1500 python tools/tune_gemm2.py $c > $O/r6_g2_tune_$c.jsonl 2> $O/r6_g2_tune_$c.err; tail -2 $O/r6_g2_tune_$c.err; wc -l $O/r6_g2_tune_$c.jsonl
done
echo done
