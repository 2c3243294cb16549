"""Pin the UNet / VAE / scheduler half of the oracle against diffusers itself - the day a box has diffusers.

`oracle/sd.py` restates `UNet2DConditionModel`, `AutoencoderKL.decode`, `DDPMScheduler` and the LoRA-patched attention
from the published architecture: diffusers is a dependency of the reference (`requirements.txt:6`,
`diffusers>=0.22.1`) that is NOT installed in the build container and cannot be installed (no network), so that half of
the oracle is "parity unpinned" (DESIGN.md section 2).  This script closes it in one command wherever diffusers can be
imported.  It is development tooling: it is never imported by the product, by tests/ or by bench.py, it does not travel
to the GPU box as anything but text, and it copies nothing from diffusers - it only CALLS it.

    python tools/compare_with_diffusers.py [--write-golden] [--size tiny|sd15]

What it does, for a seeded random-weight model of the chosen size (weights keyed by the upstream state-dict names,
comat_amd/weights.py, so they load into diffusers unchanged):
  1. UNet: `UNet2DConditionModel(**cfg).load_state_dict(sd)` vs `oracle.sd.unet_forward` on the same (sample, t, ctx):
     the call of `TrainableSDPipeline.py:144-150`; with rank-r LoRA layers on to_q/to_k/to_v/to_out.0 added the way
     `training_utils/pipeline.py:95-114` adds them (LoRALinearLayer on every Attention) and the gradients of those
     factors for a fixed cotangent;
  2. VAE: `AutoencoderKL.decode(z / scaling_factor)` vs `oracle.sd.vae_decode` (`TrainableSDPipeline.py:220`);
  3. scheduler: `DDPMScheduler.set_timesteps / step` with the reference's config (`training_utils/pipeline.py:51-59`:
     fixed_small variance) vs `oracle.sd.DDPM` on every step of a 5-step and a 50-step schedule
     (`TrainableSDPipeline.py:95,136,166`);
  4. --write-golden: stores inputs + diffusers' outputs as tests/golden/sd_tiny.npz, which tests/test_oracle.py picks up
     when present (`test_oracle_matches_diffusers_golden`), turning "parity unpinned" into a committed fixture.
Exit status 0 = every comparison within 2e-5 relative (fp32, CPU)."""
from __future__ import annotations

import argparse
import dataclasses
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TOL = 2e-5


def rel(a, b):
    a, b = a.detach().double().reshape(-1), b.detach().double().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


def diffusers_unet_kwargs(cfg):
    """UNetConfig (comat_amd/config.py) -> constructor arguments of diffusers.UNet2DConditionModel"""
    nb = len(cfg.block_out_channels)
    down = tuple("CrossAttnDownBlock2D" if a else "DownBlock2D" for a in cfg.down_attn)
    up = tuple("CrossAttnUpBlock2D" if a else "UpBlock2D" for a in cfg.up_attn)
    kw = dict(sample_size=8, in_channels=cfg.in_channels, out_channels=cfg.out_channels, down_block_types=down,
              up_block_types=up, block_out_channels=tuple(cfg.block_out_channels), layers_per_block=cfg.layers_per_block,
              cross_attention_dim=cfg.cross_attention_dim, norm_num_groups=cfg.norm_groups, flip_sin_to_cos=True,
              freq_shift=0, use_linear_projection=cfg.linear_projection)
    if cfg.heads_per_level:  # SDXL: attention_head_dim is the number of heads per level in diffusers' (mis)naming
        kw["attention_head_dim"] = tuple(cfg.heads_per_level)
        kw["transformer_layers_per_block"] = tuple(cfg.transformer_layers)
    else:
        kw["attention_head_dim"] = cfg.num_heads
    if cfg.addition_embed:
        kw.update(addition_embed_type="text_time", addition_time_embed_dim=cfg.addition_time_embed_dim,
                  projection_class_embeddings_input_dim=cfg.pooled_dim + 6 * cfg.addition_time_embed_dim)
    assert nb == len(down)
    return kw


def add_lora(unet, lora_sd, rank):
    """what training_utils/pipeline.py:95-114 does: a LoRALinearLayer (scale 1, network_alpha None) on to_q / to_k /
    to_v / to_out[0] of every Attention; weights from `lora_sd` ('<attn>.to_q.lora.down.weight', ...)"""
    from diffusers.models.lora import LoRALinearLayer
    params = {}
    for name, mod in unet.named_modules():
        if not (name.endswith("attn1") or name.endswith("attn2")):
            continue
        for proj, target in (("to_q", mod.to_q), ("to_k", mod.to_k), ("to_v", mod.to_v), ("to_out.0", mod.to_out[0])):
            layer = LoRALinearLayer(target.in_features, target.out_features, rank=rank)
            layer.down.weight.data.copy_(lora_sd[f"{name}.{proj}.lora.down.weight"])
            layer.up.weight.data.copy_(lora_sd[f"{name}.{proj}.lora.up.weight"])
            target.set_lora_layer(layer)
            params[f"{name}.{proj}.lora.down.weight"] = layer.down.weight
            params[f"{name}.{proj}.lora.up.weight"] = layer.up.weight
    return params


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="tiny", choices=["tiny", "sd15"])
    ap.add_argument("--write-golden", action="store_true")
    args = ap.parse_args()
    try:
        import diffusers
        from diffusers import AutoencoderKL, DDPMScheduler, UNet2DConditionModel
    except ImportError:
        print("diffusers is not importable here: nothing compared (the oracle's UNet / VAE / scheduler stay 'parity "
              "unpinned'; run this where `pip install 'diffusers>=0.22.1'` is possible)")
        return 2
    from comat_amd import config, weights
    from oracle import sd as O

    torch.manual_seed(0)
    ucfg = config.TINY_UNET if args.size == "tiny" else config.SD15_UNET
    vcfg = config.TINY_VAE if args.size == "tiny" else config.SD15_VAE
    usd = weights.make_unet_weights(ucfg, seed=1234, perturb_norms=True)
    vsd = weights.make_vae_weights(vcfg, seed=2345, perturb_norms=True)
    lsd = {k: (v * 5 if k.endswith("up.weight") else v) for k, v in weights.make_lora_weights(ucfg, seed=4321).items()}
    ocfg, ovcfg = O.UNetConfig(**dataclasses.asdict(ucfg)), O.VAEConfig(**dataclasses.asdict(vcfg))
    g = torch.Generator().manual_seed(7)
    B, hw, L = 2, 8 if args.size == "tiny" else 64, 7 if args.size == "tiny" else 77
    x = torch.randn(B, 4, hw, hw, generator=g)
    ctx = torch.randn(B, L, ucfg.cross_attention_dim, generator=g)
    gout = torch.randn(B, 4, hw, hw, generator=g)
    t = 417
    report, golden = [], {}

    # ---- 1. UNet (+ LoRA) ------------------------------------------------------------------------------------------------
    unet = UNet2DConditionModel(**diffusers_unet_kwargs(ucfg)).eval()
    missing, unexpected = unet.load_state_dict(usd, strict=False)
    assert not unexpected and not [m for m in missing if "lora" not in m], (missing, unexpected)
    with torch.no_grad():
        e_d = unet(x, t, encoder_hidden_states=ctx, return_dict=False)[0]
        e_o = O.unet_forward(usd, ocfg, x, t, ctx)
    report.append(("unet eps (no LoRA)", rel(e_o, e_d)))
    lp = add_lora(unet, lsd, ucfg.lora_rank)
    for p in unet.parameters():
        p.requires_grad_(False)
    for p in lp.values():
        p.requires_grad_(True)
    xd = x.clone().requires_grad_(True)
    e_d = unet(xd, t, encoder_hidden_states=ctx, return_dict=False)[0]
    (e_d * gout).sum().backward()
    lo = {k: v.clone().requires_grad_(True) for k, v in lsd.items()}
    xo = x.clone().requires_grad_(True)
    e_o = O.unet_forward(usd, ocfg, xo, t, ctx, lo)
    (e_o * gout).sum().backward()
    report.append(("unet eps (LoRA)", rel(e_o, e_d)))
    report.append(("unet d eps / d sample", rel(xo.grad, xd.grad)))
    gd = torch.cat([lp[k].grad.reshape(-1) for k in sorted(lp)])
    go = torch.cat([lo[k].grad.reshape(-1) for k in sorted(lp)])
    report.append(("LoRA gradients (all factors)", rel(go, gd)))
    golden.update(unet_x=x.numpy(), unet_ctx=ctx.numpy(), unet_t=np.int64(t), unet_gout=gout.numpy(),
                  unet_eps=e_d.detach().numpy(), unet_dx=xd.grad.numpy(), lora_grad=gd.numpy(),
                  lora_names=np.array(sorted(lp)))

    # ---- 2. VAE decode -------------------------------------------------------------------------------------------------------
    vae = AutoencoderKL(in_channels=3, out_channels=vcfg.out_channels, latent_channels=vcfg.latent_channels,
                        block_out_channels=tuple(vcfg.block_out_channels), layers_per_block=vcfg.layers_per_block,
                        norm_num_groups=vcfg.norm_groups, down_block_types=("DownEncoderBlock2D",) * len(vcfg.block_out_channels),
                        up_block_types=("UpDecoderBlock2D",) * len(vcfg.block_out_channels),
                        scaling_factor=vcfg.scaling_factor).eval()
    missing, unexpected = vae.load_state_dict(vsd, strict=False)  # the oracle holds the decoder half only
    assert not unexpected and all(m.startswith(("encoder.", "quant_conv.")) for m in missing), (missing, unexpected)
    z = torch.randn(1, 4, hw, hw, generator=g)
    with torch.no_grad():
        i_d = vae.decode(z / vcfg.scaling_factor, return_dict=False)[0]
        i_o = O.vae_decode(vsd, ovcfg, z / vcfg.scaling_factor)
    report.append(("vae decode", rel(i_o, i_d)))
    golden.update(vae_z=z.numpy(), vae_image=i_d.numpy())

    # ---- 3. scheduler ----------------------------------------------------------------------------------------------------------
    sch = DDPMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", num_train_timesteps=1000,
                        steps_offset=1, timestep_spacing="leading", clip_sample=False, variance_type="fixed_small")
    worst = 0.0
    for n in (5, 50):
        sch.set_timesteps(n)
        mine = O.DDPM()
        ts = mine.set_timesteps(n)
        assert [int(v) for v in sch.timesteps] == [int(v) for v in ts], "timestep tables differ"
        lat = torch.randn(1, 4, 8, 8, generator=g)
        for tt in ts:
            eps, noise = torch.randn(1, 4, 8, 8, generator=g), torch.randn(1, 4, 8, 8, generator=g)
            torch.manual_seed(123)
            ref = sch.step(eps, int(tt), lat, generator=None, return_dict=True).prev_sample
            # diffusers draws its own noise: recover it from its output (x_prev = mu + sigma z) to compare the MEAN and sigma
            got0 = mine.step(eps, int(tt), lat, torch.zeros_like(lat))
            got1 = mine.step(eps, int(tt), lat, torch.ones_like(lat))
            sigma = (got1 - got0).mean()
            torch.manual_seed(123)
            zz = torch.randn(lat.shape) if int(tt) > 0 else torch.zeros_like(lat)
            worst = max(worst, rel(got0 + sigma * zz, ref))
            del noise
    report.append(("ddpm step (5- and 50-step schedules)", worst))

    ok = True
    print(f"diffusers {diffusers.__version__}, size {args.size}")
    for name, v in report:
        flag = "ok" if v < TOL else "MISMATCH"
        ok &= v < TOL
        print(f"  {name:40s} rel-L2 {v:.3e}  {flag}")
    if args.write_golden and ok:
        path = os.path.join(ROOT, "tests", "golden", "sd_tiny.npz" if args.size == "tiny" else "sd15.npz")
        np.savez_compressed(path, **golden)
        print(f"wrote {path}")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
