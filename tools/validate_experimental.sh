#!/bin/bash
# First GPU call of the next round: parity-test every prepared-but-unvalidated variant, then A/B each one on the default
# bench (C2) and the SDXL one (C4).  Run on the GPU box from the repo root:  bash tools/validate_experimental.sh
# Results land in gpurun_out/exp_*.log (one JSON line per bench run; grep ms_per_step).
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
ms() { grep -o '"ms_per_step": [0-9.]*' "$1" | tail -1; }

COMAT_TEST_EXPERIMENTAL=1 timeout 600 python -m pytest tests -m gpu -q \
    -k "tile or inblock or sdxl_graph or fused_qkv or flash_trim or merged" > gpurun_out/exp_tests.log 2>&1 < /dev/null
tail -3 gpurun_out/exp_tests.log

B="python bench.py --no-cpu-baseline --no-kernel-timing --steps 4 --warmup 1"
timeout 200 $B > gpurun_out/exp_base.log 2>&1 < /dev/null;                              echo "baseline         $(ms gpurun_out/exp_base.log)"
COMAT_TILE_AUTO=1 timeout 200 $B > gpurun_out/exp_tile_auto.log 2>&1 < /dev/null;       echo "TILE_AUTO=1      $(ms gpurun_out/exp_tile_auto.log)"
COMAT_KSPLIT=2 timeout 200 $B > gpurun_out/exp_ks2.log 2>&1 < /dev/null;                echo "KSPLIT=2         $(ms gpurun_out/exp_ks2.log)"
COMAT_KSPLIT=4 timeout 200 $B > gpurun_out/exp_ks4.log 2>&1 < /dev/null;                echo "KSPLIT=4         $(ms gpurun_out/exp_ks4.log)"
COMAT_FLASH_TRIM=1 timeout 200 $B > gpurun_out/exp_trim.log 2>&1 < /dev/null;            echo "FLASH_TRIM=1     $(ms gpurun_out/exp_trim.log)"
COMAT_FLASH_TR=1 timeout 200 $B > gpurun_out/exp_tr.log 2>&1 < /dev/null;                echo "FLASH_TR=1       $(ms gpurun_out/exp_tr.log)"
COMAT_FLASH_TR=1 COMAT_FLASH_TRIM=1 timeout 200 $B > gpurun_out/exp_tr_trim.log 2>&1 < /dev/null; echo "FLASH_TR+TRIM    $(ms gpurun_out/exp_tr_trim.log)"
COMAT_BLIP_FUSED_QKV=1 timeout 200 $B > gpurun_out/exp_qkv.log 2>&1 < /dev/null;        echo "BLIP_FUSED_QKV=1 $(ms gpurun_out/exp_qkv.log)"
COMAT_TILE_AUTO=1 COMAT_KSPLIT=2 COMAT_BLIP_FUSED_QKV=1 COMAT_FLASH_TRIM=1 COMAT_FLASH_TR=1 timeout 200 $B > gpurun_out/exp_all.log 2>&1 < /dev/null; echo "all five         $(ms gpurun_out/exp_all.log)"
C3="python bench.py --config c3 --no-cpu-baseline --no-kernel-timing --steps 3 --warmup 2"
timeout 300 $C3 > gpurun_out/exp_c3_base.log 2>&1 < /dev/null;                          echo "c3 baseline      $(ms gpurun_out/exp_c3_base.log)"
COMAT_NOGRAD_MERGED=1 timeout 300 $C3 > gpurun_out/exp_c3_merged.log 2>&1 < /dev/null;  echo "c3 NOGRAD_MERGED $(ms gpurun_out/exp_c3_merged.log)"
C4="python bench.py --config c4 --no-cpu-baseline --no-kernel-timing --steps 2 --warmup 1"
timeout 300 $C4 > gpurun_out/exp_c4_base.log 2>&1 < /dev/null;                          echo "c4 baseline      $(ms gpurun_out/exp_c4_base.log)"
COMAT_SDXL_GRAPHS=1 timeout 300 $C4 > gpurun_out/exp_c4_graphs.log 2>&1 < /dev/null;    echo "c4 SDXL_GRAPHS=1 $(ms gpurun_out/exp_c4_graphs.log)"

COMAT_TEST_FULLSIZE=1 timeout 600 python -m pytest tests/test_zz_fullsize_c1.py -m gpu -q > gpurun_out/exp_c1_golden.log 2>&1 < /dev/null; tail -2 gpurun_out/exp_c1_golden.log
# full-size C1 parity (north-star acceptance numbers): a few minutes of host time for the CPU oracle
timeout 1500 python tools/parity_c1.py > gpurun_out/parity_c1.log 2>&1 < /dev/null; tail -1 gpurun_out/parity_c1.log
