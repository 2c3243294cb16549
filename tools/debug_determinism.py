"""Finds the first kernel call whose outputs differ between two identical eager runs (race / uninitialised read hunt)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from comat_amd import config, weights, ops, _hip
from comat_amd.unet import UNet, LoRABank

class Recorder:
    def __init__(self, inner):
        self.inner, self.log = inner, []
    def __getattr__(self, name):
        fn = getattr(self.inner, name)
        if not callable(fn):
            return fn
        def wrapped(*a, **kw):
            r = fn(*a, **kw)
            ts = [t for t in list(a) + list(kw.values()) if isinstance(t, torch.Tensor) and t.is_floating_point() and t.numel() > 20000]
            sig = tuple(float(t.float().double().sum()) for t in ts)
            shapes = [tuple(t.shape) for t in ts]
            self.log.append((name, sig, shapes, [x for x in a if isinstance(x, int)][:8]))
            return r
        return wrapped

dev = torch.device("cuda:0"); dtype = torch.bfloat16
cfg = config.SD15_UNET
usd = weights.make_unet_weights(cfg); lsd = weights.make_lora_weights(cfg)
hipk = _hip.HipKernels()
ops.set_kernel_backend(hipk)
bank = LoRABank(cfg, lsd, dtype, dev); unet = UNet(cfg, usd, dtype, dev, bank)
B, h, w, L = 2, 64, 64, 77
g = torch.Generator().manual_seed(0)
x = torch.randn(B*h*w, 4, generator=g).to(dev, dtype); ctx = torch.randn(B*L, cfg.cross_attention_dim, generator=g).to(dev, dtype)
with torch.no_grad():
    unet(x, B, h, w, 334, ctx, L)
logs = []
for run in range(2):
    rec = Recorder(hipk); ops.set_kernel_backend(rec)
    with torch.no_grad():
        unet(x, B, h, w, 334, ctx, L)
    torch.cuda.synchronize(); logs.append(rec.log)
a, b = logs
print("calls", len(a), len(b))
nd = 0
for i, (u, v) in enumerate(zip(a, b)):
    if u[1] != v[1]:
        print("DIFF at call", i, u[0], u[2], u[3], [x - y for x, y in zip(u[1], v[1])])
        nd += 1
        if nd >= 6: break
print("done, diffs:", nd)
