"""Golden vectors for the attribute-concentration branch of the sampler from the reference's OWN
`AttrConcenTrainableSDPipeline.forward` + `_attrcon_forward`, run in the build container.

    python tests/golden/make_attrcon_sampler_golden.py      # writes tests/golden/attrcon_sampler.npz

The two methods (AttrConcenTrainableSDPipeline.py:34-235 and :239-279; the module imports diffusers and spaCy at the top) are
pulled out of the source with `ast` and executed as they are.  Stand-ins: a toy UNet over latents whose blocks are built from
stand-in `Attention` modules (tests/golden/make_attn_golden.py) patched by the reference's own `register_attention_control`,
with the reference's own `AttentionStore` as controller and its own `get_cross_attn_map_from_unet`; the scheduler / VAE /
pass-through helpers of make_sampler_golden.py; a parser that returns nothing.  What the vectors pin: on which steps the
capturing branch runs (trained AND in `attrcon_train_steps`), that the conditional half goes through the UNet on its own and
only ITS cross-attention probabilities are kept, `[uncond; cond]` order of the reassembled prediction, `attn_dict[str(t)]`
keyed `{place}_{res}` with maps of shape (bs * heads, res, res, L), and the gradient that a loss on the captured maps sends
into the UNet's weights."""
import ast
import os
import sys
import textwrap
import types
from typing import Any, Callable, Dict, List, Optional, Tuple, Union

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_attn_golden as toy  # noqa: E402 - torchvision shim, stand-in Attention / Block, the reference's tc_attn_utils
import make_sampler_golden as base  # noqa: E402 - StubScheduler


class ToyLatentUNet(nn.Module):
    """latents [B,4,h,w] -> eps [B,4,h,w]: tokens at h*w ('down', 'up') and at (h/2)*(w/2) ('mid'), self + cross attention"""

    def __init__(self, dim=16, ctx_dim=8, heads=2):
        super().__init__()
        self.inp, self.outp = nn.Linear(4, dim), nn.Linear(dim, 4)
        self.down_blocks = toy.Block(dim, ctx_dim, heads, 1)
        self.mid_block = toy.Block(dim, ctx_dim, heads, 1)
        self.up_blocks = toy.Block(dim, ctx_dim, heads, 1)

    def forward(self, latents, t, encoder_hidden_states=None, cross_attention_kwargs=None, added_cond_kwargs=None,
                return_dict=False):
        B, C, h, w = latents.shape
        x = self.inp(latents.permute(0, 2, 3, 1).reshape(B, h * w, C)) * (1.0 + 1e-3 * float(t))
        if added_cond_kwargs is not None:  # SDXL: a per-sample shift from the pooled embedding and the size / crop ids
            x = x + (added_cond_kwargs["text_embeds"].mean(dim=1) + 1e-3 * added_cond_kwargs["time_ids"].float().sum(dim=1)).reshape(B, 1, 1)
        a = self.down_blocks(x, encoder_hidden_states)
        m = a.reshape(B, h // 2, 2, w // 2, 2, -1).mean(dim=(2, 4)).reshape(B, (h // 2) * (w // 2), -1)
        m = self.mid_block(m, encoder_hidden_states)
        up = m.reshape(B, h // 2, 1, w // 2, 1, -1).expand(-1, -1, 2, -1, 2, -1).reshape(B, h * w, -1)
        c = self.up_blocks(a + up, encoder_hidden_states)
        return (self.outp(c).reshape(B, h, w, C).permute(0, 3, 1, 2),)


def reference_methods(file="AttrConcenTrainableSDPipeline.py", cls_name="AttrConcenTrainableSDPipeline"):
    src = open(os.path.join(REF, file)).read()
    cls = next(n for n in ast.parse(src).body if isinstance(n, ast.ClassDef) and n.name == cls_name)
    ns = {"torch": torch, "Union": Union, "List": List, "Optional": Optional, "Callable": Callable, "Dict": Dict, "Any": Any,
          "Tuple": Tuple, "get_cross_attn_map_from_unet": toy.ref_attn.get_cross_attn_map_from_unet}
    for name in ("forward", "_attrcon_forward"):
        fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == name)
        exec(compile(textwrap.dedent(ast.get_source_segment(src, fn)), f"{file}:{name}", "exec"), ns)
    return ns["forward"], ns["_attrcon_forward"]


def main():
    forward, attrcon_forward = reference_methods()
    torch.manual_seed(2)
    g = torch.Generator().manual_seed(44)
    bs, h, w, L, C, N = 2, 16, 16, 5, 8, 5   # 16 x 16 and 8 x 8 tokens: resolutions `get_cross_attn_map_from_unet` looks for
    layers = ["mid_8", "up_16"]
    net = ToyLatentUNet(ctx_dim=C)
    with torch.no_grad():  # a well-conditioned sampler: the toy's prediction stays small against the latents (with CFG 7.5 a
        net.outp.weight.mul_(0.05)  # large one makes the 5-step chain amplify rounding differences by orders of magnitude)
        net.outp.bias.mul_(0.05)
    store = toy.ref_attn.AttentionStore(layers)
    toy.ref_attn.register_attention_control(net, store)
    V = torch.randn(3, 4, generator=g) * 0.5
    lat0 = torch.randn(bs, 4, h, w, generator=g)
    noises = [torch.randn(bs, 4, h, w, generator=g) for _ in range(N)]
    cond, uncond = torch.randn(bs, L, C, generator=g), torch.randn(bs, L, C, generator=g)
    gimg, glat = torch.randn(bs, 3, h, w, generator=g), torch.randn(bs, 4, h, w, generator=g)
    out = dict(V=V, latents=lat0, noises=torch.stack(noises), cond=cond, uncond=uncond, gimg=gimg, glat=glat, n_steps=np.int64(N),
               layers=np.array(layers), heads=np.int64(2), scaling_factor=np.float64(0.18215))
    for k, v in net.state_dict().items():
        out["w:" + k] = v.clone()
    watch = ["up_blocks.attn.1.to_q.weight", "mid_block.attn.1.to_k.weight", "down_blocks.attn.0.to_v.weight", "inp.weight"]
    for name, train, attr in (("a", [1, 3], [3]), ("b", [0, 2, 4], [2, 0]), ("c", [1, 2], [4])):  # c: the drawn step is not trained
        net.zero_grad()
        x0 = lat0.clone().requires_grad_(True)
        self = types.SimpleNamespace(_execution_device=torch.device("cpu"), unet=net, scheduler=base.StubScheduler(noises),
                                     controller=store, attn_dict={}, parser=lambda p: None)
        self._attrcon_forward = lambda *a, **k: attrcon_forward(self, *a, **k)
        self.encode_prompt = lambda prompt, device, n, cfg, neg, prompt_embeds=None, negative_prompt_embeds=None, lora_scale=None: \
            (prompt_embeds, negative_prompt_embeds)
        self.prepare_latents = lambda b, c, hh, ww, dtype, device, generator, latents: latents
        self.prepare_extra_step_kwargs = lambda generator, eta: {}
        self.vae = types.SimpleNamespace(dtype=torch.float32, config=types.SimpleNamespace(scaling_factor=0.18215),
                                         decode=lambda z, return_dict=False: (torch.einsum("oc,bchw->bohw", V, z),))
        prev = torch.is_grad_enabled()
        image, latents = forward(self, prompt=["p0", "p1"], height=8 * h, width=8 * w, training_timesteps=list(train),
                                 detach_gradient=True, bp_on_trained=True, num_inference_steps=N, guidance_scale=7.5,
                                 latents=x0 * 1.0, prompt_embeds=cond, negative_prompt_embeds=uncond, output_type="image",
                                 return_latents=True, attrcon_train_steps=list(attr))
        torch.set_grad_enabled(prev)
        loss = (image * gimg).sum() + (latents * glat).sum()
        keys = []
        for ts in sorted(self.attn_dict):
            for place in sorted(self.attn_dict[ts]):
                for i, m in enumerate(self.attn_dict[ts][place]):
                    keys.append(f"{ts}:{place}:{i}")
                    out[f"{name}:map:{ts}:{place}:{i}"] = m.detach().clone()
                    loss = loss + 3.0 * (m ** 2).sum()
        loss.backward()
        out[f"{name}:train"], out[f"{name}:attr"], out[f"{name}:map_keys"] = np.array(train), np.array(attr), np.array(keys)
        out[f"{name}:image"], out[f"{name}:latents"] = image.detach(), latents.detach()
        out[f"{name}:dx0"] = x0.grad.clone() if x0.grad is not None else torch.zeros_like(x0)
        for wname in watch:
            out[f"{name}:d:{wname}"] = dict(net.named_parameters())[wname].grad.clone()
        print(name, "trained", train, "drawn", attr, "-> captured", keys)
    # ---- SDXL (AttrConcenTrainableSDXLPipeline.py:234-496): the added conditioning is cut in halves with the batch, the UNet
    # input is detached on every step, latents.half() are decoded and returned raw
    forward_xl, attrcon_forward_xl = reference_methods("AttrConcenTrainableSDXLPipeline.py", "AttrConcenTrainableSDXLPipeline")
    pooled, npooled = torch.randn(bs, 5, generator=g), torch.randn(bs, 5, generator=g)
    out.update(pooled=pooled, npooled=npooled, xl_scaling_factor=np.float64(0.13025))
    for name, train, attr in (("xa", [1, 3], [3]), ("xb", [0, 2, 4], [2, 0])):
        net.zero_grad()
        x0 = lat0.clone().requires_grad_(True)
        self = types.SimpleNamespace(_execution_device=torch.device("cpu"), unet=net, scheduler=base.StubScheduler(noises),
                                     controller=store, attn_dict={}, parser=lambda p: None)
        self._attrcon_forward = lambda *a, **k: attrcon_forward_xl(self, *a, **k)
        self.encode_prompt = lambda **kw: (kw["prompt_embeds"], kw["negative_prompt_embeds"], kw["pooled_prompt_embeds"],
                                           kw["negative_pooled_prompt_embeds"])
        self.prepare_latents = lambda b, c, hh, ww, dtype, device, generator, latents: latents
        self.prepare_extra_step_kwargs = lambda generator, eta: {}
        self._get_add_time_ids = lambda osz, crop, tsz, dtype=None: torch.tensor([list(osz) + list(crop) + list(tsz)], dtype=dtype)
        self.vae = types.SimpleNamespace(config=types.SimpleNamespace(scaling_factor=0.13025),
                                         decode=lambda z, return_dict=False: (torch.einsum("oc,bchw->bohw", V.to(z.dtype), z),))
        prev = torch.is_grad_enabled()
        image, latents = forward_xl(self, prompt=["p0", "p1"], height=8 * h, width=8 * w, training_timesteps=list(train),
                                    detach_gradient=True, num_inference_steps=N, guidance_scale=7.5, latents=x0 * 1.0,
                                    prompt_embeds=cond, negative_prompt_embeds=uncond, pooled_prompt_embeds=pooled,
                                    negative_pooled_prompt_embeds=npooled, return_latents=True, attrcon_train_steps=list(attr))
        torch.set_grad_enabled(prev)
        loss = (image.float() * gimg).sum() + (latents.float() * glat).sum()
        keys = []
        for ts in sorted(self.attn_dict):
            for place in sorted(self.attn_dict[ts]):
                for i, m in enumerate(self.attn_dict[ts][place]):
                    keys.append(f"{ts}:{place}:{i}")
                    out[f"{name}:map:{ts}:{place}:{i}"] = m.detach().clone()
                    loss = loss + 3.0 * (m ** 2).sum()
        loss.backward()
        out[f"{name}:train"], out[f"{name}:attr"], out[f"{name}:map_keys"] = np.array(train), np.array(attr), np.array(keys)
        out[f"{name}:image"], out[f"{name}:latents"] = image.detach().float(), latents.detach().float()
        out[f"{name}:dx0"] = x0.grad.clone() if x0.grad is not None else torch.zeros_like(x0)
        for wname in watch:
            out[f"{name}:d:{wname}"] = dict(net.named_parameters())[wname].grad.clone()
        print(name, "trained", train, "drawn", attr, "-> captured", keys)
    np.savez_compressed(os.path.join(HERE, "attrcon_sampler.npz"), **{k: (v.numpy() if torch.is_tensor(v) else v) for k, v in out.items()})


if __name__ == "__main__":
    main()
