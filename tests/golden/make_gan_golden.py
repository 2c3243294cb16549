"""Golden vectors for the GAN loss assembly from the reference's OWN `D_sd.D_sd_pipeline_forward`, run in the build container.

    python tests/golden/make_gan_golden.py      # writes tests/golden/gan_losses.npz

`training_utils/gan_sdxl.py` imports diffusers at the top (absent here), so the method's definition is pulled out of the
source with `ast` (with `set_D_sd_pipeline_lora` and `get_D_gt_noise`, which it calls) and executed as it is on a stand-in object - nothing of it is stored in this repository, only its outputs.
The stand-ins are the things the method receives from elsewhere: the discriminator UNet (here a small deterministic function
of latents, timestep and text condition whose result depends on the order of the batch and on every argument), the scheduler
(`set_timesteps` / `timesteps` / the identity `scale_model_input` of DDPM), the 4 -> 1 linear head with fixed weights, and
the argument namespace (`condition_discriminator=False`, `gan_unet_lastlayer_cls=False`: scripts/sd15.sh, scripts/sdxl.sh).
What the vectors pin: which timestep the discriminator is evaluated at (the LAST of a fresh N-step schedule), the text
condition (the null embedding; twice for the D side), the batch order [generated; real], the targets (1 for the generator
side; 0 for the generated half and 1 for the real half on the discriminator side), NHWC head, mean-reduced BCE with logits."""
import ast
import os
import sys
import textwrap
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def reference_methods(*names):
    src = open(os.path.join(REF, "training_utils", "gan_sdxl.py")).read()
    tree = ast.parse(src)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "D_sd")
    ns = {"torch": torch, "nn": nn}
    for name in names:
        fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == name)
        exec(compile(textwrap.dedent(ast.get_source_segment(src, fn)), f"gan_sdxl.py:{name}", "exec"), ns)
    return [ns[name] for name in names]


def stub_unet_fn(mix, latents, t, cond):
    """[B,4,h,w], scalar t, [B,L,C] -> [B,4,h,w]: channel mix + a per-sample shift from the condition + a timestep term"""
    shift = cond.float().mean(dim=(1, 2)).reshape(-1, 1, 1, 1)
    return torch.einsum("oc,bchw->bohw", mix, latents.float()) + shift + 0.01 * float(t) * latents.float().flip(1)


class StubUNet(nn.Module):
    def __init__(self, mix):
        super().__init__()
        self.mix = nn.Parameter(mix.clone())
        self.calls = []

    def forward(self, latents, t, encoder_hidden_states=None, cross_attention_kwargs=None, return_dict=False):
        self.calls.append(dict(t=int(t), training=self.training, batch=latents.shape[0], cond_batch=encoder_hidden_states.shape[0]))
        return (stub_unet_fn(self.mix, latents, t, encoder_hidden_states),)


class StubScheduler:
    def __init__(self, schedule):
        self.schedule = schedule
        self.timesteps = None

    def set_timesteps(self, n, device=None):
        self.timesteps = torch.tensor(self.schedule(n))

    def scale_model_input(self, x, t):  # "identity in ddpm" (the reference's own comment)
        return x


def main():
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import sd as O
    fwd, set_lora, get_gt = reference_methods("D_sd_pipeline_forward", "set_D_sd_pipeline_lora", "get_D_gt_noise")
    g = torch.Generator().manual_seed(21)
    bs, h, w, L, C, N = 2, 5, 6, 7, 12, 5
    mix = torch.randn(4, 4, generator=g) * 0.7
    head_w, head_b = torch.randn(1, 4, generator=g) * 0.8, torch.randn(1, generator=g) * 0.3
    fake, real = torch.randn(bs, 4, h, w, generator=g), torch.randn(bs, 4, h, w, generator=g)
    null = torch.randn(bs, L, C, generator=g)
    self = types.SimpleNamespace()
    self.unet = StubUNet(mix)
    self.mlp = nn.Sequential(nn.Linear(4, 1))
    with torch.no_grad():
        self.mlp[0].weight.copy_(head_w)
        self.mlp[0].bias.copy_(head_b)
    self.cls_loss_fn = nn.BCEWithLogitsLoss()
    self.D_args = types.SimpleNamespace(condition_discriminator=False, gan_unet_lastlayer_cls=False)
    self.ori_scheduler = StubScheduler(lambda n: O.DDPM().set_timesteps(n))
    self.weight_dtype = torch.float32
    # the discriminator's trainable set (get_trainable_parameters: its LoRA factors - here the stand-in's `mix` - plus the head)
    self.D_parameters = [self.unet.mix] + list(self.mlp.parameters())
    flags = []

    def set_and_record(requires_grad=True):  # the reference's own method, with the flag it was called with written down
        flags.append(bool(requires_grad))
        set_lora(self, requires_grad=requires_grad)
    self.set_D_sd_pipeline_lora = set_and_record
    self.get_D_gt_noise = lambda device, **kw: get_gt(self, device, **kw)
    kw = dict(negative_prompt_embeds=null, num_inference_steps=N, batch=dict(latents=real))
    fake_g = fake.clone().requires_grad_(True)
    g_loss = fwd(self, fake_g, side="G", **kw)
    g_loss.backward()
    d_loss = fwd(self, fake.clone().detach(), side="D", **kw)
    d_loss.backward()
    out = dict(mix=mix, head_w=head_w, head_b=head_b, fake=fake, real=real, null=null, n_steps=np.int64(N),
               g_loss=g_loss.detach(), d_loss=d_loss.detach(), g_dfake=fake_g.grad,
               d_dhead_w=self.mlp[0].weight.grad.clone(), d_dmix=self.unet.mix.grad.clone(),
               t_used=np.array([c["t"] for c in self.unet.calls]), unet_training=np.array([c["training"] for c in self.unet.calls]),
               unet_batch=np.array([c["batch"] for c in self.unet.calls]), cond_batch=np.array([c["cond_batch"] for c in self.unet.calls]),
               lora_flags=np.array(flags))
    np.savez(os.path.join(HERE, "gan_losses.npz"), **{k: (v.numpy() if torch.is_tensor(v) else v) for k, v in out.items()})
    print({k: (tuple(v.shape) if hasattr(v, "shape") else v) for k, v in out.items()})
    print("G loss", float(g_loss), "D loss", float(d_loss), "t", out["t_used"], "lora flags", flags)


if __name__ == "__main__":
    main()
