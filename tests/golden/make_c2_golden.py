"""Generates tests/golden/c2_full.npz: the GAN leg at FULL size - BASELINE config C2's loss set (concept matching + GAN fidelity:
generator-side discriminator loss, then the discriminator step on [fake.detach(); real]) on the SD1.5 generator AND the SD1.5
discriminator, 1 prompt, 2 trained denoise steps, fp32 - evaluated by the CPU oracle (oracle/step.py train_step, no optimizer)
on seeded weights and inputs.  Complements c1_full.npz (concept matching alone): VERDICT r4 "full-size parity exists for C1's loss
set only".  Stored: the scalars, and the generator's / discriminator's LoRA gradients as per-tensor norms + 8 Rademacher inner
products each (make_c1_golden.rademacher), the discriminator head's 5 gradient values in full.  Run in the build container
(about 10 minutes on 8 cores, ~50 GB):
    python tests/golden/make_c2_golden.py
"""
import dataclasses
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from comat_amd import config, weights  # noqa: E402
from comat_amd.step import StepConfig  # noqa: E402
from make_c1_golden import c1_inputs, rademacher  # noqa: E402
from oracle import blip as OB  # noqa: E402
from oracle import sd as O  # noqa: E402
from oracle import step as OS  # noqa: E402


def c2_inputs():
    """C1's seeded world plus the discriminator (its own SD1.5 UNet weights, LoRA and Linear(4, 1) head), the null-prompt
    embedding and a real latent - shared with tests/test_zz_fullsize_c1.py"""
    (ucfg, vcfg, bcfg), sd, batch, _, ts, crop = c1_inputs()
    sd = dict(sd, d_unet=weights.make_unet_weights(ucfg, seed=1235), d_lora=weights.make_lora_weights(ucfg, seed=4322))
    g = torch.Generator().manual_seed(99)
    sd["head_w"], sd["head_b"] = torch.randn(4, generator=g) * 0.5, torch.randn(1, generator=g) * 0.1
    g = torch.Generator().manual_seed(2000)
    batch = dict(batch, gan_null_embeds=torch.randn(1, 77, ucfg.cross_attention_dim, generator=g),
                 real_latents=torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(7)) * (0.2 / 0.18215))
    scfg = StepConfig(resolution=512, total_step=2, K=2, gan_loss=True, attrcon=False)
    return (ucfg, vcfg, bcfg), sd, batch, scfg, ts, crop


def functionals(grads, names):
    norms = np.array([float(grads[n].double().norm()) for n in names])
    proj = np.stack([(rademacher(n, grads[n].numel()).double() @ grads[n].double().reshape(-1)).numpy() for n in names])
    return norms, proj


def main():
    (ucfg, vcfg, bcfg), sd, batch, scfg, ts, crop = c2_inputs()
    W = dict(unet=sd["unet"], vae=sd["vae"], blip=sd["blip"], d_unet=sd["d_unet"], ucfg=O.UNetConfig(**dataclasses.asdict(ucfg)),
             vcfg=O.VAEConfig(**dataclasses.asdict(vcfg)), bcfg=OB.BlipConfig(**dataclasses.asdict(bcfg)),
             lora={k: v.clone().requires_grad_(True) for k, v in sd["lora"].items()},
             d_lora={k: v.clone().requires_grad_(True) for k, v in sd["d_lora"].items()},
             head_w=sd["head_w"].clone().requires_grad_(True), head_b=sd["head_b"].clone().requires_grad_(True))
    ref = OS.train_step(W, batch, scfg, ts, crop)
    names, dnames = sorted(W["lora"]), sorted(W["d_lora"])
    gn, gp = functionals(ref["g_grads"], names)
    dn, dp = functionals(ref["d_grads"], dnames)
    out = dict(names=np.array(names), grad_norm=gn, grad_proj=gp, d_names=np.array(dnames), d_grad_norm=dn, d_grad_proj=dp,
               head_grad=np.concatenate([ref["head_grads"][0].double().numpy().reshape(-1), ref["head_grads"][1].double().numpy().reshape(-1)]),
               loss=np.float64(float(ref["loss"])), blip_reward=np.float64(float(ref["Blip"])),
               G_loss=np.float64(float(ref["G_loss"])), D_loss=np.float64(float(ref["D_loss"])),
               latents_norm=np.float64(float(ref["latents"].double().norm())))
    path = os.path.join(HERE, "c2_full.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "loss", float(ref["loss"]), "G_loss", float(ref["G_loss"]), "D_loss", float(ref["D_loss"]),
          "|g|", float(np.sqrt((gn ** 2).sum())), "|d|", float(np.sqrt((dn ** 2).sum())))


if __name__ == "__main__":
    main()
