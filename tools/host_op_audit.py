"""Which lines of the host code issue torch's own kernels (copies, adds, casts) inside a training step?

    python tools/host_op_audit.py [--attrcon] [--bf16] [--top 40]

The step's arithmetic goes through libcomat_hip.so; anything torch launches by itself (strided `copy_`, `.contiguous()`,
gradient-accumulation adds, `.to(dtype)`) is glue the library does not control.  The host code is the same on the CPU
simulator of the C ABI (tests/sim_backend.py), so this runs one tiny eager step there under a TorchDispatchMode, drops
every aten op issued from INSIDE the simulator (on the GPU those are kernels of the library) and counts the rest per
`comat_amd/` source line.  Counts scale with the number of blocks, not with tensor sizes.

Round-3 reading: the only torch kernels left inside a step are autograd's own gradient-accumulation adds where a tensor
feeds two consumers (45 in the tiny G backward, 5 in D: the ~290 `CUDAFunctor_add<bf16>` per C2 step of
profiles/r03_z_kernel_trace_eager.txt, 1.3 ms) and a handful of scalar loss ops; no `.contiguous()` / `copy_` / cast is
issued by the host code.  The `elementwise_kernel_manual_unroll<direct_copy>` and `bfloat16_copy_kernel` rows of that
trace (864 + 1454 launches in a 3-step run) are the weight set-up (`w.t().contiguous()`, `.to(bf16)` once per layer at
construction), not step work - which also means the trace's "166 ms of kernels per step" overstates the step by them."""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

WATCH = ("aten::copy_", "aten::add", "aten::add_", "aten::mul", "aten::cat", "aten::clone", "aten::_to_copy", "aten::sum",
         "aten::zero_", "aten::fill_", "aten::index_select", "aten::where", "aten::div", "aten::sub", "aten::neg",
         "aten::zeros", "aten::zeros_like", "aten::empty_strided")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--attrcon", action="store_true")
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--bf16", action="store_true")
    args = ap.parse_args()
    from comat_amd import ops
    from sim_backend import SimKernels
    from test_step import make_world
    ops.set_kernel_backend(SimKernels())
    dtype = torch.bfloat16 if args.bf16 else torch.float32
    cfg, batch, W, tr = make_world(dtype, torch.device("cpu"), args.attrcon)
    fixed = dict(training_steps=[1, 2], crop=(0, 0, 63, 63))
    tr.train_step(batch, **fixed)  # memoised tables, optimizer state
    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode
    sites = collections.Counter()
    shapes = {}

    class Audit(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, a=(), kw=None):
            name = "aten::" + func.__name__.split(".")[0]
            if name in WATCH:
                frames = traceback.extract_stack()
                if not any(f.filename.endswith("sim_backend.py") for f in frames):  # else: a kernel of the library
                    mine = [f for f in frames if "/comat_amd/" in f.filename]
                    if mine:
                        f = mine[-1]
                        key = (f"{f.filename.split('/comat_amd/')[1]}:{f.lineno} {f.name}", name)
                        sites[key] += 1
                        shapes.setdefault(key, " ".join(f"{tuple(x.shape)}{'' if x.is_contiguous() else '*'}"
                                                        for x in a if torch.is_tensor(x))[:60])
            return func(*a, **(kw or {}))

    with Audit():
        tr.train_step(batch, **fixed)
    print(f"# torch-issued ops per comat_amd source line in one eager step (tiny world, attrcon={args.attrcon}, {dtype})")
    tot = collections.Counter()
    for (site, name), n in sites.items():
        tot[name] += n
    print("# totals: " + ", ".join(f"{k} {v}" for k, v in tot.most_common()))
    for (site, name), n in sites.most_common(args.top):
        print(f"{n:5d}  {name:20s} {site:60s} {shapes[(site, name)]}")


if __name__ == "__main__":
    main()
