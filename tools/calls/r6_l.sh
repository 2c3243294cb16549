#!/bin/bash
# round 6: counters of the k-loop deletion builds (effective clock, waits, L2 hit rate, request latency) on conv 1x256x256 512->256 and gemm 4096^3
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
PA="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES"
PB="SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE"
PC="TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCC_EA0_RDREQ_sum"
: > $O/r6l_pmc.txt
for lib in libcomat_hip.so libcomat_hip_d1.so libcomat_hip_d3.so; do
  dbs=""
  i=0
  for pass in "$PA" "$PB" "$PC"; do
    i=$((i+1))
    d=/tmp/pmc_${lib}_$i
    (cd /tmp && MB_ONLY=big MB_CFGS=1,12 COMAT_LIB_PATH="$GRAFT_REPO_ROOT/comat_amd/lib/$lib" timeout 200 rocprofv3 --pmc $pass -d $d -o p -- python "$GRAFT_REPO_ROOT/tools/mb_diag.py" > $GRAFT_REPO_ROOT/$O/r6l_run_${lib}_$i.log 2>&1)
    dbs="$dbs $(find $d -name '*_results.db' | head -1)"
  done
  echo "#### $lib" >> $O/r6l_pmc.txt
  python tools/pmc_dump.py $dbs --match gemm2 >> $O/r6l_pmc.txt 2>&1
done
tail -60 $O/r6l_pmc.txt
