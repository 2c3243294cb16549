"""Generates tests/golden/c1_full.npz: BASELINE config C1 at FULL size (SD1.5, 1 prompt, 2 trained denoise steps,
concept-matching loss only, fp32) evaluated by the CPU oracle (oracle/step.py) on seeded weights and inputs
(comat_amd.weights factories, seeds below).  The LoRA gradient (25.5 M values) is stored as functionals: per tensor
its L2 norm and 8 inner products with Rademacher (+-1) vectors — E[<d, r>^2] = |d|^2, so the differences of the
products estimate the gradient error norm per tensor.  Run in the build container (about 2 minutes, 34 GB):
    python tests/golden/make_c1_golden.py
"""
import dataclasses
import os
import sys
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from comat_amd import config, weights  # noqa: E402
from comat_amd.step import StepConfig  # noqa: E402
from oracle import blip as OB  # noqa: E402
from oracle import sd as O  # noqa: E402
from oracle import step as OS  # noqa: E402

NPROJ = 8


def c1_inputs():
    """(configs, weights, batch, step config, training steps, crop) — shared with tests/test_zz_fullsize_c1.py"""
    ucfg, vcfg, bcfg = config.SD15_UNET, config.SD15_VAE, config.BLIP_LARGE
    sd = dict(unet=weights.make_unet_weights(ucfg, seed=1234), vae=weights.make_vae_weights(vcfg, seed=2345),
              blip=weights.make_blip_weights(bcfg, seed=3456), lora=weights.make_lora_weights(ucfg, seed=4321))
    scfg = StepConfig(resolution=512, total_step=2, K=2, gan_loss=False, attrcon=False)
    g = torch.Generator().manual_seed(1000)
    ids = torch.randint(1000, bcfg.vocab_size - 2, (1, 16), generator=g)
    batch = dict(prompt_embeds=torch.randn(1, 77, ucfg.cross_attention_dim, generator=g),
                 negative_prompt_embeds=torch.randn(1, 77, ucfg.cross_attention_dim, generator=g),
                 latents=torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(42)),
                 noises=[torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(100 + i)) for i in range(2)],
                 blip_input_ids=ids, blip_attention_mask=torch.ones_like(ids))
    return (ucfg, vcfg, bcfg), sd, batch, scfg, [0, 1], (1, 1, 510, 510)


def rademacher(name, numel):
    """+-1 vectors [NPROJ, numel] seeded by the parameter's name (independent of any parameter ordering)"""
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    return torch.randint(0, 2, (NPROJ, numel), generator=g, dtype=torch.int8).float() * 2 - 1


def main():
    (ucfg, vcfg, bcfg), sd, batch, scfg, ts, crop = c1_inputs()
    W = dict(unet=sd["unet"], vae=sd["vae"], blip=sd["blip"], ucfg=O.UNetConfig(**dataclasses.asdict(ucfg)),
             vcfg=O.VAEConfig(**dataclasses.asdict(vcfg)), bcfg=OB.BlipConfig(**dataclasses.asdict(bcfg)),
             lora={k: v.clone().requires_grad_(True) for k, v in sd["lora"].items()})
    ref = OS.g_loss_terms(W, batch, scfg, ts, crop)
    ref["loss"].backward()
    names = sorted(W["lora"])
    norms = np.array([float(W["lora"][n].grad.double().norm()) for n in names])
    proj = np.stack([(rademacher(n, W["lora"][n].numel()).double() @ W["lora"][n].grad.double().reshape(-1)).numpy()
                     for n in names])
    img = ref["image"].detach()
    out = dict(names=np.array(names), grad_norm=norms, grad_proj=proj, loss=np.float64(float(ref["loss"])),
               blip_reward=np.float64(float(ref["Blip"])), token_logp=ref["token_logp"].detach().numpy(),
               latents_norm=np.float64(float(ref["latents"].double().norm())),
               image_mean=np.float64(float(img.double().mean())), image_norm=np.float64(float(img.double().norm())),
               image_samples=img[0, :, ::64, ::64].numpy())
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c1_full.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "loss", float(ref["loss"]), "grad norm", float(np.sqrt((norms ** 2).sum())))


if __name__ == "__main__":
    main()
