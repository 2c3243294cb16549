#!/bin/bash
# round 4, call 12: fused attention, same-box A/B of five builds of attention.hip (comat_amd/lib/ab/):
#   base   = main (4ba517fb8216a0fa)
#   fix    = + loads_landed() at the end of every prologue (no vmcnt(0) between a loop's tile loads and its first MFMA) and the
#            lse scaling moved from the load to the LDS store (dK/dV)
#   w2     = fix + COMAT_FLASH_W2 (amdgpu_waves_per_eu(2,2) on the bf16 head-dim <= 64 kernels)
#   e1     = w2 + COMAT_FLASH_EARLY_TR (transposed LDS reads of the second product issued ahead of the softmax arithmetic)
#   e1only = fix + COMAT_FLASH_EARLY_TR
# then the attention parity tests on the two candidates, then SQ counter passes on `fix` (where the wave cycles go).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
AB=$GRAFT_REPO_ROOT/comat_amd/lib/ab
for v in base fix w2 e1 e1only; do
  c=""; [ $v = fix -o $v = e1 ] && c="--check"
  COMAT_LIB_PATH=$AB/libcomat_$v.so timeout 240 python tools/mb_flash_ab.py $c > $O/r4l_mb_flash_$v.txt 2>&1
  echo "== $v"; tail -1 $O/r4l_mb_flash_$v.txt; grep "Nq=4096 Nk=4096 d= 40\|Nq=1024 Nk=1024 d= 80\|Nq= 256 Nk= 256" $O/r4l_mb_flash_$v.txt | head -4
done
for v in e1 fix; do
  echo "== attention tests on $v"
  COMAT_LIB_PATH=$AB/libcomat_$v.so timeout 400 python -m pytest tests/test_ops.py tests/test_fullsize.py -m gpu -q -p no:cacheprovider -k "flash or attention" > $O/r4l_test_$v.log 2>&1; tail -3 $O/r4l_test_$v.log
done
echo "== SQ counters (fix)"
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC"
n=0
for pass in "$P1" "$P2" "$P3"; do
  n=$((n+1))
  (cd /tmp && COMAT_LIB_PATH=$AB/libcomat_fix.so timeout 200 rocprofv3 --pmc $pass -d /tmp/fd$n -o f -- python "$GRAFT_REPO_ROOT/tools/pmc_flash_diag.py" > "$GRAFT_REPO_ROOT/$O/r4l_pmc_$n.log" 2>&1)
done
python tools/pmc_dump.py $(find /tmp/fd1 /tmp/fd2 /tmp/fd3 -name "*_results.db") --match flash > $O/r4l_pmc_flash.txt 2>&1; head -60 $O/r4l_pmc_flash.txt
echo done
