#!/bin/bash
# round 6: C2 / C3 / C4 step A/B of the staged epilogue (libcomat_hip.so) against the round-5 epilogue (libcomat_hip_stage0.so); GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
: > $O/r6u_ab.txt
for lib in libcomat_hip_stage0.so libcomat_hip.so libcomat_hip_stage0.so libcomat_hip.so; do
  echo "c2 $lib $(COMAT_LIB_PATH=comat_amd/lib/$lib COMAT_SECONDARY=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)" | tee -a $O/r6u_ab.txt
done
for lib in libcomat_hip_stage0.so libcomat_hip.so; do
  echo "c4 $lib $(COMAT_LIB_PATH=comat_amd/lib/$lib timeout 900 python bench.py --config c4 --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)" | tee -a $O/r6u_ab.txt
  echo "c3 $lib $(COMAT_LIB_PATH=comat_amd/lib/$lib timeout 900 python bench.py --config c3 --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)" | tee -a $O/r6u_ab.txt
done
echo "== GPU suite"
COMAT_TEST_REPORT=$O/r6u_report.txt timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/r6u_tests.log 2>&1; tail -6 $O/r6u_tests.log
echo done
