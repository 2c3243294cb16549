"""Generates tests/golden/attn_store.npz and tests/golden/grounding_loss.npz from the reference's OWN files
(attn_utils/tc_attn_utils.py and attn_utils/tc_loss_utils.py), imported from /root/reference in the build container.
    python tests/golden/make_attn_golden.py
The fixtures hold inputs + expected outputs only.

tc_attn_utils patches `.forward` of modules whose class is named `Attention`; diffusers is not installed, so a
stand-in `Attention` with the attributes the patched forward touches is used (to_q/to_k/to_v/to_out, heads,
head_to_batch_dim, batch_to_head_dim, get_attention_scores, ...), arranged in a toy net with down/mid/up children.
tc_loss_utils imports torchvision.transforms.Resize, which is absent: a local shim with the semantics of torchvision
0.15 `Resize(antialias=True)` on tensors (bilinear, align_corners=False, antialias; non-float inputs are
interpolated in fp32 and cast back — a bool mask becomes `interp != 0`) is registered in sys.modules first."""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

# ---- torchvision shim (generator only) ---------------------------------------------------------------------------
tv = types.ModuleType("torchvision")
tvt = types.ModuleType("torchvision.transforms")


class Resize:
    def __init__(self, size, antialias=None):
        self.size, self.antialias = size, antialias

    def __call__(self, x):
        dt = x.dtype
        y = F.interpolate(x.float(), size=self.size, mode="bilinear", align_corners=False, antialias=bool(self.antialias))
        if dt in (torch.uint8, torch.int8, torch.int16, torch.int32, torch.int64):
            y = y.round()
        return y.to(dt)


tvt.Resize = Resize
tv.transforms = tvt
sys.modules["torchvision"] = tv
sys.modules["torchvision.transforms"] = tvt
sys.path.insert(0, REF)
from attn_utils import tc_attn_utils as ref_attn  # noqa: E402
from attn_utils import tc_loss_utils as ref_loss  # noqa: E402


class Attention(nn.Module):  # stand-in for diffusers.models.attention_processor.Attention
    def __init__(self, dim, ctx_dim, heads):
        super().__init__()
        self.heads = heads
        self.to_q = nn.Linear(dim, dim, bias=False)
        self.to_k = nn.Linear(ctx_dim, dim, bias=False)
        self.to_v = nn.Linear(ctx_dim, dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(dim, dim), nn.Dropout(0.0)])
        self.spatial_norm = None
        self.group_norm = None
        self.norm_cross = False
        self.residual_connection = False
        self.rescale_output_factor = 1.0
        self.scale = (dim // heads) ** -0.5

    def prepare_attention_mask(self, m, *a):
        return m

    def head_to_batch_dim(self, t):
        b, n, c = t.shape
        return t.reshape(b, n, self.heads, c // self.heads).permute(0, 2, 1, 3).reshape(b * self.heads, n, c // self.heads)

    def batch_to_head_dim(self, t):
        bh, n, d = t.shape
        b = bh // self.heads
        return t.reshape(b, self.heads, n, d).permute(0, 2, 1, 3).reshape(b, n, self.heads * d)

    def get_attention_scores(self, q, k, mask=None):
        s = torch.baddbmm(torch.empty(q.shape[0], q.shape[1], k.shape[1]), q, k.transpose(-1, -2), beta=0, alpha=self.scale)
        return s.softmax(dim=-1)


class Block(nn.Module):
    def __init__(self, dim, ctx_dim, heads, n):
        super().__init__()
        self.attn = nn.ModuleList()
        for _ in range(n):
            self.attn.append(Attention(dim, dim, heads))      # self
            self.attn.append(Attention(dim, ctx_dim, heads))  # cross

    def forward(self, x, ctx):
        for i, a in enumerate(self.attn):
            x = x + a(x, encoder_hidden_states=None if i % 2 == 0 else ctx)
        return x


class ToyUNet(nn.Module):
    def __init__(self, dim=16, ctx_dim=12, heads=2):
        super().__init__()
        self.down_blocks = Block(dim, ctx_dim, heads, 1)
        self.mid_block = Block(dim, ctx_dim, heads, 1)
        self.up_blocks = Block(dim, ctx_dim, heads, 2)

    def forward(self, x16, x4, ctx):
        a = self.down_blocks(x16, ctx)
        b = self.mid_block(x4, ctx)
        c = self.up_blocks(a, ctx)
        return a.sum() + b.sum() + c.sum()


def attn_golden():
    torch.manual_seed(0)
    net = ToyUNet()
    store = ref_attn.AttentionStore(["mid_2", "up_4"])
    ref_attn.register_attention_control(net, store)
    B, L = 2, 7
    x16, x4, ctx = torch.randn(B, 16, 16), torch.randn(B, 4, 16), torch.randn(B, L, 12)
    out = {"num_att_layers": np.array(store.num_att_layers), "x16": x16.numpy(), "x4": x4.numpy(), "ctx": ctx.numpy()}
    for k, v in net.state_dict().items():
        out["w:" + k] = v.numpy()
    # (1) with grad: probs require grad -> stored
    store.reset()
    net(x16, x4, ctx)
    maps = ref_attn.get_cross_attn_map_from_unet(store, False, reses=[4, 2])
    out["keys"] = np.array(sorted(maps.keys()))
    for k, lst in maps.items():
        out[f"n:{k}"] = np.array(len(lst))
        for i, m in enumerate(lst):
            out[f"map:{k}:{i}"] = m.detach().numpy()
    # gradient of a loss on a captured map w.r.t. a projection weight (clone() keeps the graph)
    loss = (maps["up_4"][1] ** 2).sum()
    g = torch.autograd.grad(loss, net.up_blocks.attn[3].to_q.weight)[0]
    out["grad_up_attn3_to_q"] = g.numpy()
    # (2) no grad: the hook is gated on attention_probs.requires_grad -> nothing stored
    store.reset()
    with torch.no_grad():
        net(x16, x4, ctx)
    out["nograd_stored"] = np.array(sum(len(v) for v in store.step_store.values()) +
                                    sum(len(v) for v in store.attention_store.values()) if store.attention_store else
                                    sum(len(v) for v in store.step_store.values()))
    np.savez_compressed(os.path.join(HERE, "attn_store.npz"), **out)
    print("attn_store.npz keys:", list(out["keys"]), "layers", int(out["num_att_layers"]), "nograd", int(out["nograd_stored"]))


def loss_golden():
    g = torch.Generator().manual_seed(1)
    out = {}
    for case, (res, heads, n_maps) in enumerate([(8, 4, 3), (16, 2, 1), (4, 8, 2)]):
        L = 77
        maps = [torch.softmax(torch.randn(heads, res, res, L, generator=g) * 2, dim=-1) for _ in range(n_maps)]
        masks = []
        for r in ((10, 30, 5, 40), (28, 60, 20, 64)):
            m = torch.zeros(1, 1, 64, 64, dtype=torch.bool)
            m[..., r[0]:r[1], r[2]:r[3]] = True
            masks.append(m)
        words = [[2, 3], [6, 7, 9]]
        d = ref_loss.get_grounding_loss_by_layer(masks, words, res, maps, False)
        out[f"c{case}:res"] = np.array(res)
        out[f"c{case}:maps"] = torch.stack(maps).numpy()
        out[f"c{case}:masks"] = torch.cat(masks).numpy()
        out[f"c{case}:token_loss"] = np.array(float(d["token_loss"]))
        out[f"c{case}:pixel_loss"] = np.array(float(d["pixel_loss"]))
    out["words"] = np.array([2, 3, -1, 6, 7, 9])  # -1 separates objects
    np.savez_compressed(os.path.join(HERE, "grounding_loss.npz"), **out)
    print("grounding_loss.npz", {k: float(v) for k, v in out.items() if k.endswith("_loss")})


if __name__ == "__main__":
    attn_golden()
    loss_golden()
