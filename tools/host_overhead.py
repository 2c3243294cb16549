"""Host-side cost of one CoMat step, measured without a GPU: tensors live on the `meta` device and every kernel entry
point is a no-op, so what remains is exactly the Python + autograd + allocator-bookkeeping work the host must finish
per step to keep the GPU fed.  Usage: python tools/host_overhead.py [--config c2] [--steps 3] [--profile]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from comat_amd import _hip, ops  # noqa: E402


class NullKernels:
    name = "null"

    def __init__(self):
        self.calls = 0

    def __getattr__(self, name):
        if name not in _hip.HipKernels.__dict__:
            raise AttributeError(name)

        def nop(*a, **k):
            self.calls += 1
        return nop


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--device", default="meta", help="meta (anywhere) or cuda (real allocator / torch ops, no-op kernels)")
    args = ap.parse_args()
    import bench
    nk = NullKernels()
    ops.set_kernel_backend(nk)
    dev = torch.device(args.device)
    trainer, batch, fixed, scfg, _, _ = bench.build_world(dev, torch.bfloat16, 1, args.config)
    trainer.train_step(batch, **fixed)
    prof = cProfile.Profile() if args.profile else None
    nk.calls = 0
    t0 = time.perf_counter()
    if prof:
        prof.enable()
    for _ in range(args.steps):
        trainer.train_step(batch, **fixed)
    if prof:
        prof.disable()
    if dev.type == "cuda":
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print(f"host ms/step {dt * 1e3:.1f}   kernel API calls/step {nk.calls / args.steps:.0f}   "
          f"us/call {dt * 1e6 / (nk.calls / args.steps):.1f}")
    if prof:
        pstats.Stats(prof).sort_stats("tottime").print_stats(35)


if __name__ == "__main__":
    main()
