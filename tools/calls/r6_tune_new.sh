#!/bin/bash
# round 6: plans for the problems that are new in the step (tail-column products: N + 128 columns)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
TUNE_NEW_ONLY=1 timeout 1500 python tools/tune_gemm2.py ${*:-c2} > $O/r6_g2_tune_new.jsonl 2> $O/r6_g2_tune_new.err; tail -2 $O/r6_g2_tune_new.err; wc -l $O/r6_g2_tune_new.jsonl
echo done
