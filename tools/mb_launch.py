"""Floor of a chain of dependent tiny kernels on this system: eager launches vs hipGraph replay (us per kernel).
A CoMat step is ~12 k dependent kernels; this is the part of its time that no kernel tuning removes."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from comat_amd import _hip, ops  # noqa: E402

if __name__ == "__main__":
    K = _hip.HipKernels()
    dev = torch.device("cuda:0")
    for numel in (256, 1 << 20):
        x = torch.randn(numel, device=dev).to(torch.bfloat16)
        y = torch.empty_like(x)
        n = 2000

        def chain():
            for i in range(n // 2):
                K.unary(ops.UN_SILU, x, y, numel)
                K.unary(ops.UN_SILU, y, x, numel)

        chain()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        s.record()
        chain()
        e.record()
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        eager = s.elapsed_time(e) * 1e3 / n
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            chain()
        g.replay()
        torch.cuda.synchronize()
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        graph = s.elapsed_time(e) * 1e3 / n
        print(f"numel {numel:8d}: eager {eager:6.2f} us/kernel (host enqueue {host * 1e6 / n:5.2f} us/launch), "
              f"graph replay {graph:6.2f} us/kernel", flush=True)
