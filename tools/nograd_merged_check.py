"""Does the merged-weight no-grad forward (COMAT_NOGRAD_MERGED=1: W + s U D re-rounded to bf16 once per optimizer step, plain GEMMs
instead of the low-rank products) keep the LoRA signal?  ("bound the error against the unmerged forward")

SD1.5 UNet at full size, CFG batch 2, 64x64 latents, one no-grad call per variant, LoRA up factors scaled by s in {1, 0.1, 0.01}
(real training starts from U = 0 and moves slowly: the small scales are the relevant ones).  Per scale:
    effect   = |eps(LoRA) - eps(no LoRA)| / |eps(LoRA)|          in fp32 storage (exact-f32 MFMA): what the factors change
    unmerged = |eps_bf16 - eps_fp32| / |eps_fp32|                 the product's bf16 forward (separate low-rank products)
    merged   = the same with merged weights
    lost     = |(eps_bf16(LoRA) - eps_bf16(no LoRA)) - (eps_fp32(LoRA) - eps_fp32(no LoRA))| / |eps_fp32(LoRA) - eps_fp32(no LoRA)|
               how much of the LoRA's EFFECT each bf16 variant gets wrong (same frozen-weight rounding on both sides cancels)
and the time of one graph-replayed no-grad forward for both variants.

    python tools/nograd_merged_check.py > gpurun_out/nograd_merged.txt"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from comat_amd import _hip, config, ops, weights  # noqa: E402
from comat_amd.unet import GraphedUNetForward, LoRABank, UNet  # noqa: E402

dev = torch.device("cuda:0")
ops.set_kernel_backend(_hip.HipKernels())
cfg = config.SD15_UNET
usd = weights.make_unet_weights(cfg, seed=1234)
lsd0 = weights.make_lora_weights(cfg, seed=4321)
g = torch.Generator().manual_seed(0)
B, H, W, L = 2, 64, 64, 77
x32 = torch.randn(B * H * W, 4, generator=g)
ctx32 = torch.randn(B * L, cfg.cross_attention_dim, generator=g)


def forward(dtype, lsd, merged=False, time_it=False):
    os.environ["COMAT_NOGRAD_MERGED"] = "1" if merged else "0"
    bank = LoRABank(cfg, lsd, dtype, dev)
    unet = UNet(cfg, usd, dtype, dev, bank)
    x, ctx = x32.to(dev, dtype), ctx32.to(dev, dtype)
    with torch.no_grad():
        eps, _ = unet(x, B, H, W, 500, ctx, L)
        torch.cuda.synchronize()
        ms = None
        if time_it:
            gf = GraphedUNetForward(unet)
            gf(x, B, H, W, 500, ctx, L)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5):
                gf(x, B, H, W, 500, ctx, L)
            e.record()
            torch.cuda.synchronize()
            ms = s.elapsed_time(e) / 5
    out = eps.float().cpu()
    del unet, bank
    torch.cuda.empty_cache()
    return out, ms


rel = lambda a, b: float((a - b).double().norm() / b.double().norm())
zero = {k: (v * 0 if k.endswith("up.weight") else v) for k, v in lsd0.items()}
ref0, _ = forward(torch.float32, zero)
b0, _ = forward(torch.bfloat16, zero)
print("# scale of U | effect of the LoRA (fp32) | bf16 error unmerged | bf16 error merged | LoRA effect lost: unmerged | merged | ms per no-grad forward unmerged | merged")
for s in (1.0, 0.1, 0.01):
    lsd = {k: (v * s if k.endswith("up.weight") else v) for k, v in lsd0.items()}
    ref, _ = forward(torch.float32, lsd)
    bu, tu = forward(torch.bfloat16, lsd, merged=False, time_it=(s == 1.0))
    bm, tm = forward(torch.bfloat16, lsd, merged=True, time_it=(s == 1.0))
    d_ref = ref - ref0
    print(f"  {s:5.2f}   {rel(ref, ref0):10.3e}   {rel(bu, ref):10.3e}   {rel(bm, ref):10.3e}   "
          f"{float(((bu - b0) - d_ref).double().norm() / d_ref.double().norm()):10.3e}   "
          f"{float(((bm - b0) - d_ref).double().norm() / d_ref.double().norm()):10.3e}   "
          + (f"{tu:7.2f}   {tm:7.2f}" if tu else ""), flush=True)
os.environ["COMAT_NOGRAD_MERGED"] = "0"
