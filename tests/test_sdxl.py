"""SDXL variant (BASELINE configs C4/C5): UNet with per-level heads / transformer depths, linear projections and the
text_time additional embedding; AttrConcenTrainableSDXLPipeline semantics (UNet input always detached, raw VAE decode
with return_latents, attention maps of the cond half on attrcon steps).  Tiny configuration, HIP kernels vs oracle."""
import dataclasses

import pytest
import torch

from comat_amd import config, weights
from comat_amd.pipeline import TrainableSDXLPipeline
from comat_amd.unet import LoRABank, UNet, VAEDecoder, regroup_maps
from helpers import check, rel_l2, tok
from oracle import sd as O

DTYPES = [torch.float32, torch.bfloat16]
UCFG = config.TINY_SDXL_UNET
VCFG = dataclasses.replace(config.TINY_VAE, scaling_factor=0.13025)


def rnd(*shape, seed, dtype=torch.float32):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)).to(dtype).float()


def world(dtype):
    q = lambda d: {k: v.to(dtype).float() for k, v in d.items()}
    usd = q(weights.make_unet_weights(UCFG, perturb_norms=True))
    vsd = q(weights.make_vae_weights(VCFG, perturb_norms=True))
    lsd = q({k: (v * 5 if k.endswith("up.weight") else v) for k, v in weights.make_lora_weights(UCFG).items()})
    return usd, vsd, lsd, O.UNetConfig(**dataclasses.asdict(UCFG)), O.VAEConfig(**dataclasses.asdict(VCFG))


def test_sdxl_shapes_match_survey():
    """70 transformer blocks -> 140 attentions, 185.8 M LoRA parameters at rank 128."""
    names = weights.attention_names(config.SDXL_UNET)
    assert len(names) == 140
    r = config.SDXL_UNET.lora_rank
    n = sum(r * (a + b) for (_, qd, kvd, inner) in names for (a, b) in ((qd, inner), (kvd, inner), (kvd, inner), (inner, qd)))
    assert abs(n / 1e6 - 185.8) < 0.1
    assert len(weights.attention_names(config.SD15_UNET)) == 32


@pytest.mark.parametrize("dtype", DTYPES)
def test_sdxl_unet_forward_backward(dev, dtype):
    usd, _, lsd, ocfg, _ = world(dtype)
    B, h, w, L = 2, 8, 8, 7
    x = rnd(B, 4, h, w, seed=1, dtype=dtype)
    ctx = rnd(B, L, UCFG.cross_attention_dim, seed=2, dtype=dtype)
    pooled = rnd(B, UCFG.pooled_dim, seed=3, dtype=dtype)
    time_ids = torch.tensor([[64.0, 64.0, 0.0, 0.0, 64.0, 64.0]] * B)
    g = rnd(B, 4, h, w, seed=4, dtype=dtype)
    lo = {k: v.clone().requires_grad_(True) for k, v in lsd.items()}
    store = O.AttentionStore(["mid_2", "up_4"])
    eo = O.unet_forward(usd, ocfg, x, 334, ctx, lo, store, (pooled, time_ids))
    mo = store.maps(reses=(8, 4, 2))
    ((eo * g).sum() + sum((m * m).sum() for m in mo["up_4"])).backward()
    bank = LoRABank(UCFG, lsd, dtype, dev)
    unet = UNet(UCFG, usd, dtype, dev, bank)
    e, maps = unet(tok(x).to(dev, dtype), B, h, w, 334, ctx.reshape(-1, ctx.shape[-1]).to(dev, dtype), L,
                   capture_places=("mid", "up"), added=(pooled.to(dev), time_ids.tolist()))
    md = regroup_maps(maps, reses=(8, 4, 2))
    assert {k: len(v) for k, v in md.items()} == {k: len(v) for k, v in mo.items()}
    for k in mo:
        for a, b in zip(md[k], mo[k]):
            check(a, b, dtype, f"map {k}", factor=1 if dtype == torch.float32 else 3)
    bank.zero_grad()
    ((e.float() * tok(g).to(dev)).sum() + sum((m.float() * m.float()).sum() for m in md["up_4"])).backward()
    check(e, tok(eo), dtype, "eps", factor=1 if dtype == torch.float32 else 3)
    total = rel_l2(bank.flat_grad, torch.cat([lo[n].grad.reshape(-1) for n in bank.names]))
    assert total < (1e-3 if dtype == torch.float32 else 0.1), f"LoRA grad rel-L2 {total:.3e}"


@pytest.mark.parametrize("dtype", DTYPES)
def test_sdxl_attrcon_sampler(dev, dtype):
    usd, vsd, lsd, ocfg, ovc = world(dtype)
    bs, h, w, L = 1, 8, 8, 7
    cd = UCFG.cross_attention_dim
    lat = rnd(bs, 4, h, w, seed=10)
    cu, cc = rnd(bs, L, cd, seed=11, dtype=dtype), rnd(bs, L, cd, seed=12, dtype=dtype)
    pu, pc = rnd(bs, UCFG.pooled_dim, seed=13, dtype=dtype), rnd(bs, UCFG.pooled_dim, seed=14, dtype=dtype)
    tids = (64, 64, 0, 0, 64, 64)
    noises = [rnd(bs, 4, h, w, seed=20 + i) for i in range(3)]
    layers = ["mid_2", "up_4"]
    lo = {k: v.clone().requires_grad_(True) for k, v in lsd.items()}
    img_o, lat_o, ad_o = O.sample_with_grad(usd, ocfg, vsd, ovc, lo, cu, cc, lat, noises, 3, [1, 2], 7.5,
                                            attrcon_steps=[2], train_layer_ls=layers, reses=(8, 4, 2),
                                            sdxl_cond=(pu, pc, torch.tensor([tids], dtype=torch.float32)))
    gi = rnd(*img_o.shape, seed=30)
    ((img_o * gi).sum() + sum((m * m).sum() for m in ad_o["1"]["up_4"])).backward()
    bank = LoRABank(UCFG, lsd, dtype, dev)
    pipe = TrainableSDXLPipeline(UNet(UCFG, usd, dtype, dev, bank), VAEDecoder(VCFG, vsd, dtype, dev))
    img, latf = pipe.forward(cc, cu, height=8 * h, width=8 * w, training_timesteps=[1, 2], num_inference_steps=3,
                             guidance_scale=7.5, latents=lat, noises=noises, return_latents=True,
                             attrcon_train_steps=[2], train_layer_ls=layers, attn_reses=(8, 4, 2),
                             pooled_prompt_embeds=pc, negative_pooled_prompt_embeds=pu, add_time_ids=tids)
    bank.zero_grad()
    ((img.float() * gi.to(dev)).sum() + sum((m.float() * m.float()).sum() for m in pipe.attn_dict["1"]["up_4"])).backward()
    f = 3 if dtype == torch.float32 else 6
    check(latf, lat_o, dtype, "final latents", factor=f)
    check(img, img_o, dtype, "raw image (no /2+0.5)", factor=f)
    total = rel_l2(bank.flat_grad, torch.cat([lo[n].grad.reshape(-1) for n in bank.names]))
    assert total < (1e-3 if dtype == torch.float32 else 0.15), f"LoRA grad rel-L2 {total:.3e}"


@pytest.mark.gpu
def test_sdxl_graphed_nograd_unet_matches_eager(hip):
    """hipGraph replay of the SDXL no-grad UNet forward with the prompt embedding as a graph INPUT: replays with
    different latents / text / pooled embeddings match eager launches bit for bit, with one captured graph."""
    from comat_amd.unet import GraphedUNetForward
    dtype = torch.bfloat16
    usd, _, lsd, _, _ = world(dtype)
    bank = LoRABank(UCFG, lsd, dtype, hip)
    unet = UNet(UCFG, usd, dtype, hip, bank)
    gu = GraphedUNetForward(unet)
    B, h, w, L = 2, 8, 8, 7
    time_ids = [[64.0, 64.0, 0.0, 0.0, 64.0, 64.0]] * B
    for it in range(3):
        x = tok(rnd(B, 4, h, w, seed=40 + it, dtype=dtype)).to(hip, dtype)
        ctx = rnd(B * L, UCFG.cross_attention_dim, seed=50 + it, dtype=dtype).to(hip, dtype)
        pooled = rnd(B, UCFG.pooled_dim, seed=60 + it, dtype=dtype).to(hip)
        with torch.no_grad():
            aug = unet.added_embedding(pooled, time_ids)
            ref, _ = unet(x, B, h, w, 334, ctx, L, added=aug)
            ref2, _ = unet(x, B, h, w, 334, ctx, L, added=(pooled, time_ids))
            gu.new_sampler_call()  # a new text context (the pipeline says so at the start of every forward)
            got = gu(x, B, h, w, 334, ctx, L, added=aug).clone()
        assert torch.equal(ref, ref2), "precomputed added embedding == tuple form"
        assert torch.equal(got, ref), f"iteration {it}: graph replay != eager"
        bank.flat.mul_(1.01)
        bank.mark_updated()
    assert len(gu.graphs) == 1
