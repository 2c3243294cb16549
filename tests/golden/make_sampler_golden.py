"""Golden vectors for the K-of-N differentiable sampler LOOP from the reference's OWN `TrainableSDPipeline.forward`, run in
the build container.

    python tests/golden/make_sampler_golden.py      # writes tests/golden/sampler_loop.npz

`TrainableSDPipeline.py` subclasses a diffusers pipeline (absent here), so the method's definition (TrainableSDPipeline.py:
20-225) is pulled out of the source with `ast` and executed as it is on a stand-in object - nothing of it is stored in this
repository, only its outputs.  The stand-ins are the objects a diffusers pipeline would hold: `unet` (a small differentiable
function of latents, timestep and text condition with ONE trainable matrix standing in for the LoRA factors), `vae.decode` (a
fixed 1x1 channel map), `scheduler` (timesteps of the SD1.5 DDPM configuration and its ancestral step, from oracle/sd.py's
`DDPM` - itself pinned by the closed-form known answers of SURVEY.md section 8c - with the step noise handed in instead of
drawn), `encode_prompt` / `prepare_latents` / `prepare_extra_step_kwargs` (pass-through).  Flag set of the trainer
(training_script.py:558-567): detach_gradient=True, bp_on_trained=True, guidance 7.5, return_latents=True.
What the vectors pin: which denoise steps run with gradient (the three `torch.set_grad_enabled` gates), that the UNet input
is NOT detached on trained steps, the order of the CFG halves and the guidance formula, the latents chain, `image / 2 + 0.5`,
the returned latents - through the image, the latents, and the gradients of a loss with respect to the stand-in's trainable
matrix and to the initial latents, for three choices of trained steps."""
import ast
import os
import sys
import textwrap
import types
from typing import Any, Callable, Dict, List, Optional, Tuple, Union

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import sd as O  # noqa: E402


def reference_forward(cls_name="TrainableSDPipeline"):
    src = open(os.path.join(REF, "TrainableSDPipeline.py")).read()
    tree = ast.parse(src)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls_name)
    fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "forward")
    ns = {"torch": torch, "Union": Union, "List": List, "Optional": Optional, "Callable": Callable, "Dict": Dict, "Any": Any,
          "Tuple": Tuple}
    exec(compile(textwrap.dedent(ast.get_source_segment(src, fn)), f"TrainableSDPipeline.py:{cls_name}.forward", "exec"), ns)
    return ns["forward"]


def stub_unet(W, x, t, ctx):
    """[B,4,h,w], int t, [B,L,C] -> [B,4,h,w]: nonlinear in x (so the gradient through the INPUT matters), depends on the
    per-sample condition (so the order of the CFG halves matters) and on the timestep"""
    shift = ctx.mean(dim=(1, 2)).reshape(-1, 1, 1, 1)
    return torch.tanh(torch.einsum("oc,bchw->bohw", W, x)) * (1.0 + 1e-3 * float(t)) + 0.3 * shift + 0.1 * x.roll(1, dims=3)


class StubScheduler:
    order = 1

    def __init__(self, noises):
        self.ddpm, self.noises, self.timesteps, self.i = O.DDPM(), noises, None, 0

    def set_timesteps(self, n, device=None):
        self.timesteps = torch.tensor(self.ddpm.set_timesteps(n))
        self.i = 0

    def scale_model_input(self, x, t):
        return x

    def step(self, eps, t, x, return_dict=True, **kw):
        z = self.noises[self.i]
        self.i += 1
        prev = self.ddpm.step(eps, int(t), x, z)
        return types.SimpleNamespace(prev_sample=prev) if return_dict else (prev,)


def main():
    forward = reference_forward()
    g = torch.Generator().manual_seed(33)
    bs, h, w, L, C, N = 2, 4, 5, 6, 8, 5
    W0 = torch.randn(4, 4, generator=g) * 0.6
    V = torch.randn(3, 4, generator=g) * 0.5
    lat0 = torch.randn(bs, 4, h, w, generator=g)
    noises = [torch.randn(bs, 4, h, w, generator=g) for _ in range(N)]
    cond, uncond = torch.randn(bs, L, C, generator=g), torch.randn(bs, L, C, generator=g)
    gimg, glat = torch.randn(bs, 3, h, w, generator=g), torch.randn(bs, 4, h, w, generator=g)
    out = dict(W=W0, V=V, latents=lat0, noises=torch.stack(noises), cond=cond, uncond=uncond, gimg=gimg, glat=glat,
               n_steps=np.int64(N), scaling_factor=np.float64(0.18215))
    cases = {"a": [1, 3], "b": [0, 1, 2, 3, 4], "c": [4], "d": []}  # d: no trained step (sampling only: validation, GT latents)
    for name, train in cases.items():
        Wp = W0.clone().requires_grad_(True)
        x0 = lat0.clone().requires_grad_(True)
        calls = []

        def unet(x, t, encoder_hidden_states=None, cross_attention_kwargs=None, return_dict=False):
            calls.append((int(t), bool(torch.is_grad_enabled()), bool(x.requires_grad)))
            return (stub_unet(Wp, x, int(t), encoder_hidden_states),)
        self = types.SimpleNamespace(_execution_device=torch.device("cpu"), unet=unet, scheduler=StubScheduler(noises))
        self.encode_prompt = lambda prompt, device, n, cfg, neg, prompt_embeds=None, negative_prompt_embeds=None, lora_scale=None: \
            (prompt_embeds, negative_prompt_embeds)
        self.prepare_latents = lambda b, c, hh, ww, dtype, device, generator, latents: latents
        self.prepare_extra_step_kwargs = lambda generator, eta: {}
        self.vae = types.SimpleNamespace(dtype=torch.float32, config=types.SimpleNamespace(scaling_factor=0.18215),
                                         decode=lambda z, return_dict=False: (torch.einsum("oc,bchw->bohw", V, z),))
        prev = torch.is_grad_enabled()
        image, latents = forward(self, height=8 * h, width=8 * w, training_timesteps=list(train), detach_gradient=True,
                                 bp_on_trained=True, num_inference_steps=N, guidance_scale=7.5, latents=x0 * 1.0,
                                 prompt_embeds=cond, negative_prompt_embeds=uncond, output_type="image", return_latents=True)
        torch.set_grad_enabled(prev)  # the reference leaves the global grad mode wherever its last gate put it
        loss = (image * gimg).sum() + (latents * glat).sum()
        loss.backward()
        out[f"{name}:train"] = np.array(train)
        out[f"{name}:image"] = image.detach()
        out[f"{name}:latents"] = latents.detach()
        out[f"{name}:dW"] = Wp.grad.clone() if Wp.grad is not None else torch.zeros_like(Wp)
        out[f"{name}:dx0"] = x0.grad.clone() if x0.grad is not None else torch.zeros_like(x0)
        out[f"{name}:unet_grad_mode"] = np.array([c[1] for c in calls])
        out[f"{name}:unet_input_requires_grad"] = np.array([c[2] for c in calls])
        out[f"{name}:t"] = np.array([c[0] for c in calls])
        print(name, train, "t", [c[0] for c in calls], "grad mode", [int(c[1]) for c in calls], "input grad", [int(c[2]) for c in calls],
              "|dW|", float(out[f"{name}:dW"].norm()), "|dx0|", float(out[f"{name}:dx0"].norm()))
    # ---- SDXL (TrainableSDPipeline.py:657-846): pooled text embedding + size / crop ids as added conditioning, the UNet
    # input detached on EVERY step, latents.half() decoded and returned raw (no / 2 + 0.5) with return_latents
    forward_xl = reference_forward("TrainableSDXLPipeline")
    pooled, npooled = torch.randn(bs, 5, generator=g), torch.randn(bs, 5, generator=g)
    out.update(pooled=pooled, npooled=npooled)

    def stub_unet_xl(W, x, t, ctx, text_embeds, time_ids):
        extra = (text_embeds.mean(dim=1) + 1e-3 * time_ids.float().sum(dim=1)).reshape(-1, 1, 1, 1)
        return stub_unet(W, x, t, ctx) + 0.2 * extra
    for name, train in (("xa", [1, 3]), ("xb", [0, 1, 2, 3, 4])):
        Wp = W0.clone().requires_grad_(True)
        x0 = lat0.clone().requires_grad_(True)
        calls = []

        def unet(x, t, encoder_hidden_states=None, cross_attention_kwargs=None, added_cond_kwargs=None, return_dict=False):
            calls.append((int(t), bool(torch.is_grad_enabled()), bool(x.requires_grad)))
            return (stub_unet_xl(Wp, x, int(t), encoder_hidden_states, added_cond_kwargs["text_embeds"], added_cond_kwargs["time_ids"]),)
        self = types.SimpleNamespace(_execution_device=torch.device("cpu"), unet=unet, scheduler=StubScheduler(noises))
        self.encode_prompt = lambda **kw: (kw["prompt_embeds"], kw["negative_prompt_embeds"], kw["pooled_prompt_embeds"],
                                           kw["negative_pooled_prompt_embeds"])
        self.prepare_latents = lambda b, c, hh, ww, dtype, device, generator, latents: latents
        self.prepare_extra_step_kwargs = lambda generator, eta: {}
        self._get_add_time_ids = lambda osz, crop, tsz, dtype=None: torch.tensor([list(osz) + list(crop) + list(tsz)], dtype=dtype)
        self.vae = types.SimpleNamespace(config=types.SimpleNamespace(scaling_factor=0.13025),
                                         decode=lambda z, return_dict=False: (torch.einsum("oc,bchw->bohw", V.to(z.dtype), z),))
        prev = torch.is_grad_enabled()
        image, latents = forward_xl(self, height=8 * h, width=8 * w, training_timesteps=list(train), detach_gradient=True,
                                    num_inference_steps=N, guidance_scale=7.5, latents=x0 * 1.0, prompt_embeds=cond,
                                    negative_prompt_embeds=uncond, pooled_prompt_embeds=pooled,
                                    negative_pooled_prompt_embeds=npooled, return_latents=True)
        torch.set_grad_enabled(prev)
        ((image.float() * gimg).sum() + (latents.float() * glat).sum()).backward()
        out[f"{name}:train"] = np.array(train)
        out[f"{name}:image"] = image.detach().float()
        out[f"{name}:latents"] = latents.detach().float()
        out[f"{name}:dW"] = Wp.grad.clone()
        out[f"{name}:dx0"] = x0.grad.clone() if x0.grad is not None else torch.zeros_like(x0)
        out[f"{name}:unet_grad_mode"] = np.array([c[1] for c in calls])
        out[f"{name}:unet_input_requires_grad"] = np.array([c[2] for c in calls])
        out[f"{name}:t"] = np.array([c[0] for c in calls])
        print(name, train, "grad mode", [int(c[1]) for c in calls], "input grad", [int(c[2]) for c in calls], "image dtype", image.dtype,
              "|dW|", float(Wp.grad.norm()), "|dx0|", float(out[f"{name}:dx0"].norm()))
    out["xl_scaling_factor"] = np.float64(0.13025)
    np.savez_compressed(os.path.join(HERE, "sampler_loop.npz"), **{k: (v.numpy() if torch.is_tensor(v) else v) for k, v in out.items()})


if __name__ == "__main__":
    main()
