"""Does the merged-weight TRAINED call (COMAT_TRAIN_MERGED=1, round 5: y = x (W + s U D)^T with the sum rounded to bf16 once per
optimizer step; dx through its transpose; factor gradients from the unmerged activations) keep the LoRA signal and its
gradients?  ("a bf16 A/B of gradient error merged vs low-rank"; ADVICE r4: "validate at init-scale magnitudes")

SD1.5 UNet at full size, CFG batch 2, 64x64 latents, ONE trained call (forward + backward with a fixed cotangent) per variant,
LoRA up factors scaled by s in {1, 0.1, 0.01, 0.001} (make_lora_weights draws |U| ~ 0.02: s = 0.001 is the size of U after a
handful of AdamW steps at lr 5e-5 from U = 0).  Reference: the low-rank form in fp32 storage (exact-f32 MFMA).  Per scale:
    effect            |eps(LoRA) - eps(U = 0)| / |eps|            what the factors change in the output (fp32)
    eps error         |eps_bf16 - eps_fp32| / |eps_fp32|          low-rank | merged
    effect lost       |(eps_bf16 - eps_bf16(U = 0)) - (eps_fp32 - eps_fp32(U = 0))| / |eps_fp32 - eps_fp32(U = 0)|   low-rank | merged
    grad error        |g_bf16 - g_fp32| / |g_fp32| over the flat LoRA gradient, and separately over its down / up halves
and the fp32-storage merged call against the fp32 low-rank call (the parity mode: must stay far below 1e-3).

    python tools/train_merged_check.py > gpurun_out/train_merged.txt"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from comat_amd import _hip, config, ops, weights  # noqa: E402
from comat_amd.unet import LoRABank, UNet  # noqa: E402

dev = torch.device("cuda:0")
ops.set_kernel_backend(_hip.HipKernels())
cfg = config.SD15_UNET
usd = weights.make_unet_weights(cfg, seed=1234)
lsd0 = weights.make_lora_weights(cfg, seed=4321)
g = torch.Generator().manual_seed(0)
B, H, W, L = 2, 64, 64, 77
x32 = torch.randn(B * H * W, 4, generator=g)
ctx32 = torch.randn(B * L, cfg.cross_attention_dim, generator=g)
go32 = torch.randn(B * H * W, 4, generator=g)

worlds = {}
for dtype in (torch.float32, torch.bfloat16):
    bank = LoRABank(cfg, lsd0, dtype, dev)
    worlds[dtype] = (bank, UNet(cfg, usd, dtype, dev, bank))
up_mask = torch.cat([torch.full((worlds[torch.float32][0].params[n].numel(),), float(n.endswith("up.weight")))
                     for n in worlds[torch.float32][0].names]).bool()


def run(dtype, scale, merged):
    bank, unet = worlds[dtype]
    ops.set_train_merged(merged)
    flat = torch.cat([(lsd0[n] * (scale if n.endswith("up.weight") else 1.0)).reshape(-1).float() for n in bank.names])
    bank.flat.copy_(flat.to(dev))
    bank.mark_updated()
    bank.set_requires_grad(True)
    bank.zero_grad()
    x = x32.to(dev, dtype).requires_grad_(True)
    eps, _ = unet(x, B, H, W, 500, ctx32.to(dev, dtype), L)
    eps.backward(go32.to(dev, dtype))
    ops.join_side_streams()
    torch.cuda.synchronize()
    return eps.detach().float().cpu(), bank.flat_grad.detach().double().cpu().clone(), x.grad.detach().float().cpu()


rel = lambda a, b: float((a - b).double().norm() / b.double().norm())
e0_32, _, _ = run(torch.float32, 0.0, False)
e0_lr, _, _ = run(torch.bfloat16, 0.0, False)
e0_mg, _, _ = run(torch.bfloat16, 0.0, True)
print("# scale of U | LoRA effect on eps (fp32) | eps error bf16: low-rank, merged | LoRA effect lost: low-rank, merged | "
      "LoRA gradient error bf16 (all / down / up): low-rank ; merged | dx error: low-rank, merged | fp32 storage: merged vs low-rank (eps, grad)")
for s in (1.0, 0.1, 0.01, 0.001):
    e32, g32, dx32 = run(torch.float32, s, False)
    em32, gm32, _ = run(torch.float32, s, True)
    elr, glr, dxlr = run(torch.bfloat16, s, False)
    emg, gmg, dxmg = run(torch.bfloat16, s, True)
    d = e32 - e0_32
    lost = lambda e, e0: float(((e - e0) - d).double().norm() / d.double().norm())
    gerr = lambda a: (rel(a, g32), rel(a[~up_mask], g32[~up_mask]), rel(a[up_mask], g32[up_mask]))
    print(f"  {s:6.3f}  {rel(e32, e0_32):9.2e} | {rel(elr, e32):9.2e} {rel(emg, e32):9.2e} | {lost(elr, e0_lr):9.2e} {lost(emg, e0_mg):9.2e} | "
          + " ".join(f"{v:9.2e}" for v in gerr(glr)) + " ; " + " ".join(f"{v:9.2e}" for v in gerr(gmg))
          + f" | {rel(dxlr, dx32):9.2e} {rel(dxmg, dx32):9.2e} | {rel(em32, e32):9.2e} {rel(gm32, g32):9.2e}", flush=True)
ops.set_train_merged(os.environ.get("COMAT_TRAIN_MERGED", "1") != "0")
