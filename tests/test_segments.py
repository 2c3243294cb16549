"""Step segments (comat_amd/segments.py): the trained UNet calls, the head (VAE + BLIP + generator-side discriminator
loss) and the discriminator step replayed from hipGraphs behind the eager sampler loop / loss assembly / optimizer.
GPU: replays are bit-identical to eager launches over several steps with changing trained steps, attribute-concentration
steps, crops and inputs (SD1.5 and SDXL layouts).  CPU: the hooks themselves (slot numbering, map filtering, staging)
through the ABI simulator with the segments run eagerly (dry mode)."""
import dataclasses

import numpy as np
import pytest
import torch

from comat_amd import config, ops, weights
from comat_amd.segments import GraphedSegment, SegmentedStep
from test_step import make_world

PLAN = [([1, 2], (1, 0, 63, 63), [2]), ([1, 2], (0, 1, 63, 63), [1]), ([0, 1], (1, 1, 63, 63), [1]),
        ([1, 2], (0, 0, 63, 63), [2]), ([0, 2], (0, 1, 63, 63), [0]), ([1, 2], (1, 1, 63, 63), [1, 2])]


def vary(batch, gen, dtype):
    b = dict(batch)
    b["latents"] = torch.randn(batch["latents"].shape, generator=gen)
    b["noises"] = [torch.randn(n.shape, generator=gen) for n in batch["noises"]]
    b["prompt_embeds"] = torch.randn(batch["prompt_embeds"].shape, generator=gen).to(dtype).float()
    b["real_latents"] = torch.randn(batch["real_latents"].shape, generator=gen)
    return b


def run_plan(tr_e, stepper, batch, dtype, attrcon, same):
    gen = torch.Generator().manual_seed(11)
    for it, (ts, crop, acs) in enumerate(PLAN):
        b = vary(batch, gen, dtype)
        kw = dict(training_steps=ts, crop=crop)
        if attrcon:
            kw["attrcon_steps"] = acs
        le = tr_e.train_step(b, **kw)
        lg = stepper(b, **kw)
        if tr_e.device.type == "cuda":
            torch.cuda.synchronize()
        keys = ("step_loss", "Blip", "G_loss", "D_loss") + (("token_loss", "pixel_loss") if attrcon else ())
        for k in keys:
            assert same(le[k], lg[k]), f"step {it}: {k} {float(le[k])} (eager) vs {float(lg[k])} (segments)"
        tr_g = stepper.tr
        assert same(tr_e.bank.flat, tr_g.bank.flat), f"step {it}: generator LoRA parameters differ"
        assert same(tr_e.D.bank.flat, tr_g.D.bank.flat), f"step {it}: discriminator LoRA parameters differ"
        assert same(tr_e.D.head, tr_g.D.head)
        assert same(tr_e.opt.m[0], tr_g.opt.m[0]) and same(tr_e.opt.v[0], tr_g.opt.v[0])


@pytest.mark.parametrize("attrcon", [False, True])
def test_segment_hooks_dry(sim, attrcon):
    """CPU: SegmentedStep in dry mode (segments run eagerly through the same hooks) equals the plain eager step with the
    text key / value sharing off - slot numbering, wanted-map filtering, batch staging, head / D runners."""
    dtype = torch.float32
    cfg, batch, W, tr_e = make_world(dtype, sim, attrcon)
    cfg, _, _, tr_g = make_world(dtype, sim, attrcon)
    tr_e.pipe.share_text_kv = False
    st = SegmentedStep(tr_g, dry=True)
    run_plan(tr_e, st, batch, dtype, attrcon, lambda a, b: torch.allclose(a.float(), b.float(), rtol=0, atol=0))
    # without a GPU and without dry mode the wrapper is the eager step
    assert not SegmentedStep(tr_g).enabled


@pytest.mark.gpu
def test_graphed_segment_replays_match_eager(hip):
    """GraphedSegment on a small function with a LoRA projection group: outputs, input gradients and the in-place LoRA
    weight gradients of replays equal eager launches bit for bit, for changing inputs."""
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(3)
    M, K, N, r = 256, 64, 96, 8
    w = torch.randn(N, K, generator=g) * K ** -0.5
    lin = ops.FrozenLinear(w, torch.randn(N, generator=g), dtype, hip)
    spec = [[("d", "u", torch.randn(r, K, generator=g) * r ** -0.5, torch.randn(N, r, generator=g) * 0.2)]]
    store_e, store_g = ops.LoRAStore(spec, dtype, hip), ops.LoRAStore(spec, dtype, hip)
    gam, bet = torch.ones(N, device=hip), torch.zeros(N, device=hip)

    def make(store):
        def fn(x, y):
            h = ops.lora_linear(x, lin, store.groups[0])
            h = ops.layer_norm(h, gam, bet)
            return ops.add(h, y), ops.silu(h.detach())
        return fn

    seg = GraphedSegment(make(store_g), "toy")
    store_e.ensure_compute_copy()
    store_g.ensure_compute_copy()
    for it in range(4):
        x = (torch.randn(M, K, generator=g)).to(hip, dtype)
        y = (torch.randn(M, N, generator=g)).to(hip, dtype)
        go = (torch.randn(M, N, generator=g)).to(hip, dtype)
        xe, ye = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
        xg, yg = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
        oe, se = make(store_e)(xe, ye)
        oe.backward(go)
        if not seg.captured:
            og, sg = make(store_g)(xg, yg)
            seg.capture((xg, yg))
        else:
            og, sg = seg(xg, yg)
            assert not sg.requires_grad
        og.backward(go)
        ops.join_side_streams()
        torch.cuda.synchronize()
        assert torch.equal(oe, og) and torch.equal(se, sg)
        assert torch.equal(xe.grad, xg.grad) and torch.equal(ye.grad, yg.grad)
        assert torch.equal(store_e.flat_grad, store_g.flat_grad), f"iteration {it}: LoRA gradients differ"
    assert seg.replays == 3


@pytest.mark.gpu
@pytest.mark.parametrize("attrcon", [False, True])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_segmented_step_matches_eager(hip, dtype, attrcon):
    """six optimisation steps, eager vs segments: different trained steps (two slot signatures), attribute-concentration
    steps (both variants of a slot, sharing one memory pool), crops and inputs - parameters of G and D, optimizer
    moments and every logged loss stay bit-identical."""
    cfg, batch, W, tr_e = make_world(dtype, hip, attrcon)
    cfg, _, _, tr_g = make_world(dtype, hip, attrcon)
    tr_e.pipe.share_text_kv = False  # replayed segments project the text keys / values once per call
    st = SegmentedStep(tr_g)
    run_plan(tr_e, st, batch, dtype, attrcon, torch.equal)
    s = st.stats()
    assert s["replays"] >= 10, s
    assert st.head_seg is not None and st.head_seg.replays == len(PLAN) - 1
    assert st.d_seg is None and st.head_seg.side_out is not None  # the D step lives inside the head's backward graph


@pytest.mark.gpu
def test_step_graphs_at_a_per_gpu_batch_of_four(hip):
    """bs 4 per GPU (the reference's --train_batch_size 4, scripts/sd15.sh:4; CFG batch 8): the segment graphs and the
    whole-step graph of `secondary.c2_bs4` replay bit-identically to eager launches at the size that line uses."""
    from comat_amd.step import GraphedStep
    dtype = torch.bfloat16
    cfg, batch, W, tr_e = make_world(dtype, hip, False, bs=4)
    cfg, _, _, tr_g = make_world(dtype, hip, False, bs=4)
    assert batch["latents"].shape[0] == 4
    tr_e.pipe.share_text_kv = False
    st = SegmentedStep(tr_g)
    run_plan(tr_e, st, batch, dtype, False, torch.equal)
    assert st.failed is None and st.stats()["replays"] >= 10
    # the whole-step graph (static topology: every denoise step trained)
    cfg, _, _, tr_e2 = make_world(dtype, hip, False, bs=4)
    cfg, _, _, tr_w = make_world(dtype, hip, False, bs=4)
    gs = GraphedStep(tr_w)
    gen = torch.Generator().manual_seed(3)
    for it in range(4):
        b = vary(batch, gen, dtype)
        kw = dict(training_steps=[0, 1, 2], crop=(it % 2, 1, 63, 63))
        le, lg = tr_e2.train_step(b, **kw), gs(b, **kw)
        torch.cuda.synchronize()
        assert gs.failed is None, gs.failed
        for k in ("step_loss", "Blip", "G_loss", "D_loss"):
            assert torch.equal(le[k], lg[k]), f"step {it}: {k}"
        assert torch.equal(tr_e2.bank.flat, tr_w.bank.flat) and torch.equal(tr_e2.D.bank.flat, tr_w.D.bank.flat)
    assert len(gs.graphs) == 1


@pytest.mark.gpu
def test_segmented_step_with_its_own_discriminator_graph(hip):
    """use_d="own": the discriminator step as a separate graph on the discriminator's stream (same bits, less overlap)"""
    dtype = torch.bfloat16
    cfg, batch, W, tr_e = make_world(dtype, hip, False)
    cfg, _, _, tr_g = make_world(dtype, hip, False)
    tr_e.pipe.share_text_kv = False
    st = SegmentedStep(tr_g, use_d="own")
    run_plan(tr_e, st, batch, dtype, False, torch.equal)
    assert st.d_seg is not None and st.d_seg.replays == len(PLAN) - 1


@pytest.mark.gpu
def test_segmented_step_sdxl_matches_eager(hip):
    """SDXL layout (text_time conditioning as a segment input, UNet input always detached) + attribute concentration"""
    from comat_amd.blip import Blip
    from comat_amd.gan import D_sd
    from comat_amd.pipeline import TrainableSDXLPipeline
    from comat_amd.step import CoMatTrainer, StepConfig
    from comat_amd.unet import LoRABank, UNet, VAEDecoder
    dtype = torch.bfloat16
    ucfg = config.TINY_SDXL_UNET
    vcfg = dataclasses.replace(config.TINY_VAE, scaling_factor=0.13025)
    usd, vsd = weights.make_unet_weights(ucfg, perturb_norms=True), weights.make_vae_weights(vcfg, perturb_norms=True)
    lsd, bsd = weights.make_lora_weights(ucfg), weights.make_blip_weights(config.TINY_BLIP, perturb_norms=True)
    dsd, dl = weights.make_unet_weights(config.TINY_UNET, seed=77), weights.make_lora_weights(config.TINY_UNET, seed=78)
    g = torch.Generator().manual_seed(6)
    r = lambda *s: torch.randn(*s, generator=g)
    hw, hb = r(4) * 0.5, r(1) * 0.1
    cfg = StepConfig(resolution=64, total_step=3, K=2, gan_loss=True, attrcon=True, attrcon_train_steps=1,
                     train_layer_ls=("mid_2", "up_2", "up_4"), attn_reses=(8, 4, 2), lr=1e-2, lr_D=1e-2,
                     mask_token_loss_weight=0.5, mask_pixel_loss_weight=0.1)
    bs, L, T = 1, 7, 9
    ids = torch.randint(1, config.TINY_BLIP.vocab_size, (bs, T), generator=g)
    m = np.zeros((2, 64, 64), dtype=bool)
    m[0, 5:30, 8:40] = True
    m[1, 34:60, 20:64] = True
    batch = dict(prompt_embeds=r(bs, L, ucfg.cross_attention_dim), negative_prompt_embeds=r(bs, L, ucfg.cross_attention_dim),
                 pooled_prompt_embeds=r(bs, ucfg.pooled_dim), negative_pooled_prompt_embeds=r(bs, ucfg.pooled_dim),
                 add_time_ids=(64, 64, 0, 0, 64, 64), gan_null_embeds=r(bs, L, config.TINY_UNET.cross_attention_dim),
                 latents=r(bs, 4, 8, 8), noises=[r(bs, 4, 8, 8) for _ in range(3)], real_latents=r(bs, 4, 8, 8),
                 blip_input_ids=ids, blip_attention_mask=torch.ones_like(ids), masks=[m], attributes=[[[2, 3], [5]]])

    def world():
        bank = LoRABank(ucfg, lsd, dtype, hip)
        pipe = TrainableSDXLPipeline(UNet(ucfg, usd, dtype, hip, bank), VAEDecoder(vcfg, vsd, dtype, hip))
        dbank = LoRABank(config.TINY_UNET, dl, dtype, hip)
        disc = D_sd(UNet(config.TINY_UNET, dsd, dtype, hip, dbank), dbank, hw, hb)
        return CoMatTrainer(pipe, bank, Blip(config.TINY_BLIP, bsd, dtype, hip), disc, cfg, seed=0)

    tr_e, tr_g = world(), world()
    tr_e.pipe.share_text_kv = False
    run_plan(tr_e, SegmentedStep(tr_g), batch, dtype, True, torch.equal)
