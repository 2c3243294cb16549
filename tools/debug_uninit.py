"""Uninitialised-read hunt: poison the caching allocator's free blocks with NaNs before a step; results must not change."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from comat_amd import ops, _hip
from test_step import make_world
ops.set_kernel_backend(_hip.HipKernels())
dev = torch.device("cuda:0")

def poison():
    blocks = [torch.full((n,), float("nan"), device=dev) for n in (1 << 12, 1 << 16, 1 << 20, 1 << 22, 1 << 24, 1 << 26)] * 3
    blocks += [torch.full((n,), float("nan"), device=dev, dtype=torch.bfloat16) for n in (1 << 10, 1 << 14, 1 << 18, 3 << 20)] * 8
    torch.cuda.synchronize(); del blocks

res = []
for mode in ("clean", "poisoned"):
    for attrcon in (False, True):
        cfg, batch, W, tr = make_world(torch.bfloat16, dev, attrcon)
        if mode == "poisoned": poison()
        logs = tr.train_step(batch, training_steps=[1, 2], crop=(0, 0, 63, 63), **({"attrcon_steps": [2]} if attrcon else {}))
        if mode == "poisoned": poison()
        logs2 = tr.train_step(batch, training_steps=[1, 2], crop=(0, 0, 63, 63), **({"attrcon_steps": [2]} if attrcon else {}))
        torch.cuda.synchronize()
        res.append((mode, attrcon, float(logs["step_loss"]), float(logs2["step_loss"]), tr.bank.flat.clone(), tr.D.bank.flat.clone()))
        print(mode, attrcon, res[-1][2], res[-1][3], bool(torch.isfinite(tr.bank.flat).all()), flush=True)
for a, b in ((0, 2), (1, 3)):
    print("attrcon", res[a][1], "identical:", res[a][2:4] == res[b][2:4], torch.equal(res[a][4], res[b][4]), torch.equal(res[a][5], res[b][5]))
