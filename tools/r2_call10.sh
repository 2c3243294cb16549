#!/bin/bash
# round 2, GPU call 10: 128-byte k-tile variants of the pipelined kernel - parity tests of every block shape, then the
# per-shape plan sweep (graph-timed) over the problems of the C2 and C4 steps
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
rm -f $O/r2j_*
echo "== tests (gemm2 block shapes)"; timeout 600 python -m pytest tests/test_ops.py -m gpu -q -p no:cacheprovider -k "gemm2 or split_k or k_major" 2>&1 | tail -8 > $O/r2j_tests.log; tail -4 $O/r2j_tests.log
echo "== tune"; TUNE_TOP=${TUNE_TOP:-280} timeout 900 python tools/tune_gemm2.py c2 c4 > $O/r2j_g2_tune.jsonl 2> $O/r2j_g2_tune.err; wc -l $O/r2j_g2_tune.jsonl; tail -2 $O/r2j_g2_tune.err
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r2j_g2_tune.jsonl") if l.startswith("{")]
tot_best=sum(r["best_us"]*r["calls"] for r in rows)/1e3
def best_k2(r): return min(v for k,v in r["us"].items() if ":" in k and int(k.split(":")[0])<8)
tot_k2=sum(best_k2(r)*r["calls"] for r in rows)/1e3
tot_auto=sum(r["us"]["auto"]*r["calls"] for r in rows)/1e3
n4=sum(1 for r in rows if int(r["best"].split(":")[0])>=8)
print(f"{len(rows)} problems: current table {tot_auto:.1f} ms, best 64-byte-tile plan {tot_k2:.1f} ms, best plan incl. 128-byte tiles {tot_best:.1f} ms ({n4} problems pick a 128-byte tile)")
PY
echo done
