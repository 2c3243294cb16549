#!/bin/bash
# round 5, call 4: plans for the GEMM problems that are new in the step (merged trained / no-grad calls: batched q/k/v products,
# data-gradients without the low-rank segment) - tools/tune_gemm2.py on the problems gemm2_plans.inc does not hold; C3 with its
# per-piece GPU times, D step split over the head's two graphs or whole inside its backward graph.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out; mkdir -p $O
echo "== tune (new problems of C2 / C3 / C4)"; TUNE_NEW_ONLY=1 COMAT_PRECAPTURE=0 timeout 900 python tools/tune_gemm2.py c2 c3 c4 > $O/r5d_g2_tune.jsonl 2> $O/r5d_g2_tune.err; tail -2 $O/r5d_g2_tune.err; wc -l $O/r5d_g2_tune.jsonl
python - <<'PY'
import json
rows=[json.loads(l) for l in open('gpurun_out/r5d_g2_tune.jsonl') if l.startswith('{')]
tot_auto=sum(r['us']['auto']*r['calls'] for r in rows); tot_best=sum(r['best_us']*r['calls'] for r in rows)
print(f"{len(rows)} problems: auto {tot_auto/1e3:.2f} ms, best {tot_best/1e3:.2f} ms (per step, summed over the configurations' call counts)")
for r in sorted(rows, key=lambda r: -(r['us']['auto']-r['best_us'])*r['calls'])[:12]:
    print(r['kind'], r['M'], r['N'], r['nkt'], r['batch'], 'calls', r['calls'], 'auto', r['us']['auto'], 'best', r['best'], r['best_us'])
PY
for d in 1 0; do
  echo "== C3, COMAT_D_SPLIT=$d"
  COMAT_D_SPLIT=$d COMAT_ATTN_MAP_PROBE=0 timeout 500 python bench.py --config c3 --no-cpu-baseline --steps 4 > $O/r5d_bench_c3_dsplit$d.log 2>&1
  tail -1 $O/r5d_bench_c3_dsplit$d.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], json.dumps(d['config']['gpu_ms_per_step_by_piece']))"
done
echo done
