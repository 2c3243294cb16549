#!/bin/bash
# round 2, GPU call 13: where the host spends a step (no-op kernels, real allocator) and what one C-ABI call costs
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
timeout 300 python tools/host_overhead.py --config c2 --steps 3 --device cuda --profile > $O/r2m_host_profile.txt 2>&1; head -60 $O/r2m_host_profile.txt
timeout 120 python - <<'PY' 2>&1 | tail -8
import time, torch, sys
sys.path.insert(0, ".")
from comat_amd import _hip
k = _hip.HipKernels()
dev = torch.device("cuda:0")
x = torch.randn(64, 64, device=dev).bfloat16(); w = torch.randn(64, 64, device=dev).bfloat16(); y = torch.empty(64, 64, device=dev, dtype=torch.bfloat16)
g = torch.ones(64, device=dev); b = torch.zeros(64, device=dev); st = torch.empty(64, 2, device=dev)
def t(fn, n=20000):
    for _ in range(200): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    dt = time.perf_counter() - t0; torch.cuda.synchronize(); return dt / n * 1e6
print("gemm call       %.2f us" % t(lambda: k.gemm(x, w, y, 64, 64, 64, 64, 64, 64)))
print("layernorm call  %.2f us" % t(lambda: k.layernorm_fwd(x, g, b, y, st, 64, 64, 1e-5)))
print("unary call      %.2f us" % t(lambda: k.unary(0, x, y, 4096)))
print("new_empty       %.2f us" % t(lambda: x.new_empty((64, 64))))
print("Event pair      %.2f us" % t(lambda: torch.cuda.Event(enable_timing=True).record(), 5000))
PY
echo done
