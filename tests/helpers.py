"""Shared helpers of the parity tests."""
import dataclasses

import torch

from comat_amd import config, weights
from oracle import sd as O


def tol(dtype):
    return 2e-4 if dtype == torch.float32 else 3e-2


def check(got, ref, dtype, what="", factor=1.0):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{what}: non-finite values"
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item() + 1e-6
    assert err / scale < tol(dtype) * factor, f"{what}: max err {err:.3e} vs scale {scale:.3e} ({dtype})"


def rel_l2(got, ref):
    got, ref = got.detach().double().cpu().reshape(-1), ref.detach().double().cpu().reshape(-1)
    return float((got - ref).norm() / (ref.norm() + 1e-30))


def tok(t):
    """NCHW -> channels-last tokens"""
    return t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).contiguous()


def untok(t, B, H, W):
    return t.reshape(B, H, W, -1).permute(0, 3, 1, 2).contiguous()


def oracle_cfgs(ucfg=config.TINY_UNET, vcfg=config.TINY_VAE):
    return O.UNetConfig(**dataclasses.asdict(ucfg)), O.VAEConfig(**dataclasses.asdict(vcfg))


def tiny_weights(dtype, ucfg=config.TINY_UNET):
    """Tiny-config weights rounded to `dtype` (so oracle and kernels see identical values)."""
    q = lambda d: {k: v.to(dtype).float() for k, v in d.items()}
    usd = q(weights.make_unet_weights(ucfg, perturb_norms=True))
    vsd = q(weights.make_vae_weights(config.TINY_VAE, perturb_norms=True))
    lsd = q(weights.make_lora_weights(ucfg))
    # make LoRA up factors big enough that their gradients are well conditioned in the tiny model
    lsd = {k: (v * 5 if k.endswith("up.weight") else v).to(dtype).float() for k, v in lsd.items()}
    return usd, vsd, lsd
