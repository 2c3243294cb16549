#!/bin/bash
# round 5, call 4b: plans for the GEMM problems that are new in the step (merged trained / no-grad calls: batched q/k/v products,
# data-gradients without the low-rank segment) - tools/tune_gemm2.py on the problems gemm2_plans.inc does not hold yet.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out; mkdir -p $O
echo "== tune (new problems of C2 / C3 / C4)"; TUNE_NEW_ONLY=1 COMAT_PRECAPTURE=0 timeout 1200 python tools/tune_gemm2.py c2 c3 c4 > $O/r5d_g2_tune.jsonl 2> $O/r5d_g2_tune.err; tail -3 $O/r5d_g2_tune.err; wc -l $O/r5d_g2_tune.jsonl
python - <<'PY'
import json
rows=[json.loads(l) for l in open('gpurun_out/r5d_g2_tune.jsonl') if l.startswith('{')]
tot_auto=sum(r['us']['auto']*r['calls'] for r in rows); tot_best=sum(r['best_us']*r['calls'] for r in rows)
print(f"{len(rows)} problems: auto {tot_auto/1e3:.2f} ms, best {tot_best/1e3:.2f} ms (per step, summed over the configurations' call counts)")
for r in sorted(rows, key=lambda r: -(r['us']['auto']-r['best_us'])*r['calls'])[:12]:
    print(r['kind'], r['M'], r['N'], r['nkt'], r['batch'], 'calls', r['calls'], 'auto', r['us']['auto'], 'best', r['best'], r['best_us'])
PY
echo done
