#!/bin/bash
# round 2, GPU call 5: ds_read_b64_tr_b16 semantics probe, parity with the 8-wave tile + attention VALU diet, re-tune with
# the 8-wave tile, flash timings, default bench (as the driver runs it), kernel trace of graph-mode replay
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
ms() { grep -o '"ms_per_step": [0-9.]*' "$1" | tail -1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/probes/tr_read_probe.hip -o /tmp/tr_probe 2>/dev/null && timeout 60 /tmp/tr_probe > gpurun_out/r2e_tr_probe.txt 2>&1; head -3 gpurun_out/r2e_tr_probe.txt
timeout 1500 python -X faulthandler -m pytest tests -m gpu -q > gpurun_out/r2e_tests_all.log 2>&1 < /dev/null; echo "all gpu tests: $(tail -1 gpurun_out/r2e_tests_all.log)"
timeout 300 python tools/mb_flash.py > gpurun_out/r2e_mb_flash.txt 2>&1 < /dev/null; grep "Nq=4096 Nk=4096 d= 40 trim=1 tr=1" gpurun_out/r2e_mb_flash.txt | cut -c1-160
timeout 900 python tools/tune_gemm2.py c2 > gpurun_out/r2e_g2_tune.jsonl 2> gpurun_out/r2e_g2_tune.err < /dev/null; wc -l gpurun_out/r2e_g2_tune.jsonl
timeout 600 python bench.py > gpurun_out/r2e_bench_default.log 2>&1 < /dev/null; echo "default bench $(ms gpurun_out/r2e_bench_default.log)"; tail -1 gpurun_out/r2e_bench_default.log | cut -c1-2500
cd /tmp && COMAT_STEP_GRAPH=1 timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2e_prof -o c2 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/r2e_prof.log 2>&1 < /dev/null
cd $R
DB=$(ls gpurun_out/r2e_prof/*results.db gpurun_out/r2e_prof/*/*results.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/rocpd_summary.py $DB 1 > gpurun_out/r2e_kernel_trace_graph.txt; python tools/rocpd_timeline.py $DB 0.6 > gpurun_out/r2e_timeline_graph.txt; rm -rf gpurun_out/r2e_prof; head -12 gpurun_out/r2e_kernel_trace_graph.txt | cut -c1-160; cat gpurun_out/r2e_timeline_graph.txt; fi
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r2e_prof.log | tail -1
