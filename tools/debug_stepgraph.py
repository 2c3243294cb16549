"""Locate what breaks hipGraph capture of the whole step: capture progressively larger parts of a tiny-config step,
one stage per process (run each with `python -X faulthandler tools/debug_stepgraph.py <stage>`).
  fwd      no-grad forward of the generator losses
  bwd0     G forward + backward, no side streams, no discriminator
  bwd1     ... with the LoRA weight-gradient side stream
  opt      ... + clip / AdamW
  gan0     full step with the discriminator, D step serial (COMAT_D_STREAM=0)
  gan1     full step, D step on its own stream
  gan2     full step, D step on its own stream but its weight gradients NOT forked again (no nested fork)
  full     GraphedStep itself (eager step, capture, replay);  full2: with COMAT_GRAPH_D=fork"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
stage = sys.argv[1]
if stage == "gan0":
    os.environ["COMAT_D_STREAM"] = "0"
if stage in ("gan2", "full2"):
    os.environ["COMAT_GRAPH_D"] = "fork"
from comat_amd import _hip, ops  # noqa: E402
from test_step import make_world  # noqa: E402

dev = torch.device("cuda:0")
ops.set_kernel_backend(_hip.HipKernels())
cfg, batch, W, tr = make_world(torch.bfloat16, dev, False)
ts, crop = [1, 2], (1, 0, 63, 63)
if stage in ("fwd", "bwd0", "bwd1", "opt"):
    tr.cfg.gan_loss = False
if stage == "bwd0":
    ops.set_side_stream_enabled(False)
sb = {k: (v.to(dev) if torch.is_tensor(v) else ([n.to(dev) for n in v] if k == "noises" else v)) for k, v in batch.items()}


def body():
    if stage == "fwd":
        with torch.no_grad():
            return tr.compute_losses(sb, training_steps=ts, crop=crop)["loss"]
    if stage in ("bwd0", "bwd1"):
        return tr._forward_backward(sb, dict(training_steps=ts, crop=crop))["step_loss"]
    return tr.train_step(sb, training_steps=ts, crop=crop)["step_loss"]


if stage == "gan2":
    tr.flat_d = True
if stage in ("full", "full2"):
    from comat_amd.step import GraphedStep
    gs = GraphedStep(tr)
    for i in range(3):
        out = gs(batch, training_steps=ts, crop=crop)
        torch.cuda.synchronize()
        print(f"[{stage}] call {i}: loss {float(out['step_loss']):.6f}", flush=True)
    print(f"[{stage}] OK", flush=True)
    sys.exit(0)

print(f"[{stage}] eager: {float(body()):.6f}", flush=True)
tr.blip.static_tables = tr.blip.tables(64, 64, crop).static_copy()
tr.bank.mark_updated()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
print(f"[{stage}] capturing", flush=True)
with torch.cuda.graph(g):
    out = body()
print(f"[{stage}] captured; replaying", flush=True)
for i in range(2):
    g.replay()
    torch.cuda.synchronize()
    print(f"[{stage}] replay {i}: {float(out):.6f}", flush=True)
print(f"[{stage}] OK", flush=True)
