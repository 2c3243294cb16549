#!/bin/bash
# round 2, GPU call 1: acceptance test (full-size C1 golden), parity tests of every prepared variant, A/B benches
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
ms() { grep -o '"ms_per_step": [0-9.]*' "$1" | tail -1; }
COMAT_TEST_FULLSIZE=1 timeout 600 python -m pytest tests/test_zz_fullsize_c1.py -m gpu -q -x > gpurun_out/r2_c1_golden.log 2>&1 < /dev/null; tail -5 gpurun_out/r2_c1_golden.log
COMAT_TEST_EXPERIMENTAL=1 timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_zz_fullsize_c1.py > gpurun_out/r2_exp_tests.log 2>&1 < /dev/null
tail -15 gpurun_out/r2_exp_tests.log
B="python bench.py --no-cpu-baseline --no-kernel-timing --steps 6 --warmup 2"
timeout 200 $B > gpurun_out/r2_base.log 2>&1 < /dev/null;                              echo "baseline         $(ms gpurun_out/r2_base.log)"
COMAT_TILE_AUTO=1 timeout 200 $B > gpurun_out/r2_tile_auto.log 2>&1 < /dev/null;       echo "TILE_AUTO=1      $(ms gpurun_out/r2_tile_auto.log)"
COMAT_KSPLIT=4 timeout 200 $B > gpurun_out/r2_ks4.log 2>&1 < /dev/null;                echo "KSPLIT=4         $(ms gpurun_out/r2_ks4.log)"
COMAT_FLASH_TR=1 COMAT_FLASH_TRIM=1 timeout 200 $B > gpurun_out/r2_tr_trim.log 2>&1 < /dev/null; echo "FLASH_TR+TRIM    $(ms gpurun_out/r2_tr_trim.log)"
COMAT_BLIP_FUSED_QKV=1 timeout 200 $B > gpurun_out/r2_qkv.log 2>&1 < /dev/null;        echo "BLIP_FUSED_QKV=1 $(ms gpurun_out/r2_qkv.log)"
COMAT_TILE_AUTO=1 COMAT_KSPLIT=4 COMAT_BLIP_FUSED_QKV=1 COMAT_FLASH_TRIM=1 COMAT_FLASH_TR=1 timeout 200 $B > gpurun_out/r2_all.log 2>&1 < /dev/null; echo "all five         $(ms gpurun_out/r2_all.log)"
C4="python bench.py --config c4 --no-cpu-baseline --no-kernel-timing --steps 2 --warmup 1"
timeout 300 $C4 > gpurun_out/r2_c4_base.log 2>&1 < /dev/null;                          echo "c4 baseline      $(ms gpurun_out/r2_c4_base.log)"
COMAT_SDXL_GRAPHS=1 timeout 300 $C4 > gpurun_out/r2_c4_graphs.log 2>&1 < /dev/null;    echo "c4 SDXL_GRAPHS=1 $(ms gpurun_out/r2_c4_graphs.log)"
