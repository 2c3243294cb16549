import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from comat_amd import config, weights, ops, _hip
from comat_amd.unet import UNet, LoRABank, GraphedUNetForward
ops.set_kernel_backend(_hip.HipKernels())
dev = torch.device("cuda:0"); dtype = torch.bfloat16
size = sys.argv[1] if len(sys.argv) > 1 else "tiny"
cfg = config.TINY_UNET if size == "tiny" else config.SD15_UNET
print("size", size, flush=True)
usd = weights.make_unet_weights(cfg); lsd = weights.make_lora_weights(cfg)
bank = LoRABank(cfg, lsd, dtype, dev); unet = UNet(cfg, usd, dtype, dev, bank)
B, h, w, L = (2, 8, 8, 7) if size == "tiny" else (2, 64, 64, 77)
g = torch.Generator().manual_seed(0)
x = torch.randn(B*h*w, 4, generator=g).to(dev, dtype); ctx = torch.randn(B*L, cfg.cross_attention_dim, generator=g).to(dev, dtype)
with torch.no_grad():
    a, _ = unet(x, B, h, w, 334, ctx, L); b, _ = unet(x, B, h, w, 334, ctx, L)
torch.cuda.synchronize()
print("eager determinism max diff", float((a.float()-b.float()).abs().max()), "scale", float(a.float().abs().max()), flush=True)
gu = GraphedUNetForward(unet)
for it in range(3):
    with torch.no_grad():
        o = gu(x, B, h, w, 334, ctx, L)
    torch.cuda.synchronize()
    d = (o.float()-a.float()).abs()
    print("replay", it, "max diff vs eager", float(d.max()), "rows differing", int((d.max(1).values > 0).sum()), "of", d.shape[0], flush=True)
ts = (1, 21, 41) if size != "many" else [i * 20 + 1 for i in range(50)]
for t in ts:
    with torch.no_grad():
        o = gu(x, B, h, w, t, ctx, L)
    torch.cuda.synchronize()
    print("graph t", t, "ok", float(o.float().abs().max()), "mem GB", torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30, flush=True)

for rep in range(2):
    for t in ts:
        with torch.no_grad():
            o = gu(x, B, h, w, t, ctx, L)
torch.cuda.synchronize()
print("replayed all twice ok", flush=True)
