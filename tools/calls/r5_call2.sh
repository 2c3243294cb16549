#!/bin/bash
# round 5, call 2: the whole GPU suite on the current tree (merged trained calls, D step split over the head's two graphs, bs-4
# graphs, GEGLU fallback), the merged-vs-low-rank gradient A/B at full size, runtime knobs of the hipGraph executor on the C2
# step, C3 with / without the D split, the batch-4 line.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out; mkdir -p $O
echo "== GPU suite"; timeout 800 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/r5b_test_all.log 2>&1; tail -4 $O/r5b_test_all.log
echo "== merged vs low-rank trained call, full size"; timeout 400 python tools/train_merged_check.py > $O/r5b_train_merged.txt 2>&1; tail -6 $O/r5b_train_merged.txt
for m in 0 1; do
  rm -f $O/r5b_c1_bf16_m$m.txt
  COMAT_TRAIN_MERGED=$m COMAT_TEST_REPORT=$O/r5b_c1_bf16_m$m.txt timeout 300 python -m pytest tests/test_zz_fullsize_c1.py -m gpu -q -p no:cacheprovider -k bf16 2>&1 | tail -1
  echo "COMAT_TRAIN_MERGED=$m: $(cat $O/r5b_c1_bf16_m$m.txt)"
done
run() {  # label, env...
  local label=$1; shift
  echo "== C2 step: $label"
  env "$@" COMAT_SECONDARY=0 COMAT_PROBE_EAGER=0 timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --steps 20 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"launch_mode": "[^"]*"' | tr '\n' ' '; echo
}
{
run "default" A=1
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run "DEBUG_HIP_FORCE_GRAPH_QUEUES=1" DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run "DEBUG_HIP_FORCE_GRAPH_QUEUES=2" DEBUG_HIP_FORCE_GRAPH_QUEUES=2
run "DEBUG_HIP_GRAPH_BATCH_SIZE=1" DEBUG_HIP_GRAPH_BATCH_SIZE=1
} 2>&1 | tee $O/r5b_c2_runtime_knobs.txt
for d in 1 0; do
  echo "== C3, COMAT_D_SPLIT=$d"
  COMAT_D_SPLIT=$d timeout 400 python bench.py --config c3 --no-cpu-baseline --no-kernel-timing --steps 4 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1
done | tee $O/r5b_c3_dsplit.txt
echo "== C2 at a per-GPU batch of 4"; COMAT_SECONDARY=0 COMAT_PROBE_EAGER=0 COMAT_ATTN_MAP_PROBE=0 timeout 400 python bench.py --bs 4 --no-cpu-baseline --steps 2 > $O/r5b_bench_bs4.log 2>&1
tail -1 $O/r5b_bench_bs4.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['metric'], d['ms_per_step'], 'img/s', d['value'], 'step_frac', r['step_frac'], d['config']['launch_mode'])
for k,v in r['families'].items(): print('  ', k, v['ms'], v['launches'], v['frac_of_peak'])
" 2>&1 | tail -12
echo done
