"""fp8 forward (BASELINE.json configs[4]: "fp8 MFMA UNet forward with bf16 backward"), first slice:
  * the e4m3fn rounding rule of the oracle (oracle/fp8.py) against torch's float8_e4m3fn cast, value by value;
  * comat_fp8_scale / comat_fp8_quantize bit for bit against the oracle;
  * comat_gemm / comat_conv2d with COMAT_FP8_E4M3 operands (32x32x64 fp8 MFMA) against the oracle's dequantised products;
  * at the real SDXL-1024^2 shapes (128^2 latent level): integer-valued operands, where fp8 and bf16 arithmetic are both
    exact, must give bit-identical results on the fp8 and the bf16 kernel;
  * a small SDXL-topology UNet with fp8 forward against the oracle's fp8 emulation: eps, input gradient, LoRA gradients.

Stated fp8 tolerance: given THE SAME input bits the product and the oracle compute the same scale and the same e4m3
bytes, so one operator agrees to the fp32 summation order of its products: 2e-5 relative L2 with fp32 storage (1e-4 for
the raw GEMM/conv entry points), ~1e-2 with bf16 storage (the output rounding).  Across a whole network only a
statistical statement holds (see test_unet_fp8_forward_against_oracle_emulation for why)."""
import dataclasses

import numpy as np
import pytest
import torch

from comat_amd import config, ops, weights
from comat_amd.unet import LoRABank, UNet
from helpers import rel_l2, tok
from oracle import fp8 as OF
from oracle import sd as O

FP8_UNET = dataclasses.replace(config.TINY_SDXL_UNET, block_out_channels=(64, 128, 128), cross_attention_dim=64,
                               heads_per_level=(2, 4, 4), transformer_layers=(1, 1, 2), norm_groups=8, lora_rank=8)


def rnd(*shape, seed):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def test_oracle_rounding_rule_matches_the_e4m3fn_format():
    allb = torch.arange(256, dtype=torch.uint8)
    vals = allb.view(torch.float8_e4m3fn).float()
    finite = vals[torch.isfinite(vals)]
    assert finite.numel() == 254 and finite.abs().max() == 448.0 and finite[finite > 0].min() == 2.0 ** -9
    # every representable value is a fixed point; midpoints tie to even; everything in between goes to the nearer one
    pos = torch.sort(finite[finite >= 0]).values
    probes = [pos, -pos, (pos[:-1] + pos[1:]) / 2, -(pos[:-1] + pos[1:]) / 2]
    g = torch.Generator().manual_seed(0)
    probes.append((torch.rand(20000, generator=g) * 2 - 1) * 448)
    probes.append((torch.rand(20000, generator=g) * 2 - 1) * 0.05)
    probes.append(torch.tensor([460.0, -463.9, 1e-4, 0.00097656, 0.00097657, 0.0]))
    x = torch.cat(probes)
    want = x.clamp(-448, 448).to(torch.float8_e4m3fn).float()
    got = torch.from_numpy(OF.e4m3fn_round(x.numpy()))
    assert torch.equal(got, want)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n", [8, 1000, 4099, 320 * 4096 + 3])
def test_quantize_matches_oracle_bit_for_bit(dev, dtype, n):
    x = (rnd(n, seed=n) * 3.7).to(dtype)
    x[n // 2] = 0.0
    q_ref, s_ref = OF.quantize(x)
    q, s = ops.kernels().fp8_quantize(x.to(dev).contiguous())
    assert float(s.cpu()) == float(s_ref), (float(s.cpu()), float(s_ref))
    assert torch.equal(q.cpu(), q_ref)
    # values: the oracle's numpy rounding rule, applied to the scaled input, gives the same bytes' values
    want = OF.e4m3fn_round((x.float() * (1.0 / s_ref)).numpy()) * float(s_ref)
    assert np.array_equal(OF.dequantize(q.cpu(), s_ref).numpy(), want.astype(np.float32))


def test_quantize_of_zeros_is_zero(dev):
    q, s = ops.kernels().fp8_quantize(torch.zeros(777, device=dev))
    assert float(s.cpu()) == float(torch.tensor(2.0 ** -100) / 448.0) and int(q.cpu().abs().sum()) == 0


@pytest.mark.parametrize("shape", [(64, 64, 64), (200, 96, 128), (1, 40, 192), (333, 320, 640), (2048, 1280, 320)])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
def test_fp8_gemm_matches_oracle(dev, shape, out_dtype):
    M, N, K = shape
    x, w = rnd(M, K, seed=1) * 2.0, rnd(N, K, seed=2) * 0.05
    bias, res = rnd(N, seed=3), rnd(M, N, seed=4).to(out_dtype)
    (xq, sx), (wq, sw) = OF.quantize(x), OF.quantize(w)
    ref = OF.dequantize(xq, sx).double() @ OF.dequantize(wq, sw).double().t() + bias.double() + res.double()
    k = ops.kernels()
    x8, sxd = k.fp8_quantize(x.to(dev))
    w8, swd = k.fp8_quantize(w.to(dev))
    y = torch.empty(M, N, dtype=out_dtype, device=dev)
    k.gemm(x8, w8, y, M, N, K, K, K, N, bias=bias.to(dev), R=res.to(dev), ldr=N, beta=1.0, scales=(sxd, swd))
    lim = 1e-4 if out_dtype == torch.float32 else 6e-3  # bf16 output: one rounding of the result
    assert rel_l2(y, ref) < lim, rel_l2(y, ref)
    if dev.type == "cuda":
        from comat_amd import _hip
        assert _hip.last_gemm_kernel() == 3  # the fp8 MFMA kernel, not a fallback


@pytest.mark.parametrize("geo", [(2, 8, 8, 64, 128, 3, 1, 1, 1), (1, 16, 16, 128, 64, 3, 2, 1, 1), (2, 8, 8, 64, 64, 3, 1, 1, 2),
                                 (1, 12, 12, 192, 96, 1, 1, 0, 1)])
def test_fp8_conv_matches_oracle(dev, geo):
    B, H, W, Cin, Cout, ks, stride, pad, ups = geo
    x = rnd(B, Cin, H, W, seed=5)
    w = rnd(Cout, Cin, ks, ks, seed=6) * 0.05
    bias = rnd(Cout, seed=7)
    xq, wq = OF.fake_quant(x), OF.fake_quant(w)
    xu = torch.nn.functional.interpolate(xq, scale_factor=2, mode="nearest") if ups == 2 else xq
    ref = torch.nn.functional.conv2d(xu.double(), wq.double(), bias.double(), stride=stride, padding=pad)
    Ho, Wo = ref.shape[2:]
    k = ops.kernels()
    x8, sx = k.fp8_quantize(tok(x).to(dev))
    w8, sw = k.fp8_quantize(w.permute(0, 2, 3, 1).contiguous().to(dev))
    y = torch.empty(B * Ho * Wo, Cout, device=dev)
    k.conv2d(x8, w8, y, B, H, W, Cin, Ho, Wo, Cout, ks, ks, stride, pad, mode=0, ups=ups, bias=bias.to(dev), scales=(sx, sw))
    assert rel_l2(y, tok(ref)) < 1e-4, rel_l2(y, tok(ref))


def test_fp8_ineligible_problems_fail_loudly(dev):
    """no silent fallback: the fp8 path has no second kernel behind it (include/comat_hip.h)"""
    k = ops.kernels()
    x8, sx = k.fp8_quantize(torch.ones(32, 96, device=dev))
    w8, sw = k.fp8_quantize(torch.ones(16, 96, device=dev))
    y = torch.empty(32, 16, device=dev)
    with pytest.raises((RuntimeError, AssertionError)):
        k.gemm(x8, w8, y, 32, 16, 96, 96, 96, 16, scales=(sx, sw))  # K % 64 != 0


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["conv128_320", "conv64_640", "geglu", "attn_out"])
def test_full_shape_integer_operands_agree_bit_for_bit_with_the_bf16_kernel(hip, case):
    """SDXL at 1024^2 (128^2 latents, CFG batch 2): integer-valued activations in [-8, 8] and weights in [-4, 4] are
    exact in e4m3 AND in bf16, every product and every partial sum (< 2^24) is exact in fp32, so the fp8 MFMA path and the
    bf16 MFMA path must return the same bits whatever their tiling and summation order."""
    k = ops.kernels()
    g = torch.Generator().manual_seed(11)
    ri = lambda lo, hi, *s: torch.randint(lo, hi + 1, s, generator=g).float()
    one = torch.ones(1, device=hip)
    to8 = lambda t: t.to(torch.float8_e4m3fn).view(torch.uint8).to(hip)  # exact: small integers are e4m3 values
    if case.startswith("conv"):
        B, H, C = (2, 128, 320) if case == "conv128_320" else (2, 64, 640)
        x, w = ri(-8, 8, B * H * H, C), ri(-4, 4, C, 3, 3, C)
        x8, w8 = to8(x), to8(w)
        x, w = x.to(hip), w.to(hip)
        y8, y16 = torch.empty(B * H * H, C, device=hip), torch.empty(B * H * H, C, device=hip)
        k.conv2d(x8, w8, y8, B, H, H, C, H, H, C, 3, 3, 1, 1, scales=(one, one))
        k.conv2d(x.bfloat16(), w.bfloat16(), y16, B, H, H, C, H, H, C, 3, 3, 1, 1)
    else:
        M, N, K = (2 * 128 * 128, 2560, 320) if case == "geglu" else (2 * 64 * 64, 640, 640)
        x, w = ri(-8, 8, M, K), ri(-4, 4, N, K)
        x8, w8 = to8(x), to8(w)
        x, w = x.to(hip), w.to(hip)
        y8, y16 = torch.empty(M, N, device=hip), torch.empty(M, N, device=hip)
        k.gemm(x8, w8, y8, M, N, K, K, K, N, scales=(one, one))
        k.gemm(x.bfloat16(), w.bfloat16(), y16, M, N, K, K, K, N)
    torch.cuda.synchronize()
    assert float(y16.abs().max()) > 100 and torch.equal(y8, y16)
    # and one row against exact integer arithmetic on the host
    if not case.startswith("conv"):
        row = (x[7].cpu().double() @ w.cpu().double().t()).float()
        assert torch.equal(y8[7].cpu(), row)


def _fp8_world(dev, dtype=torch.float32):
    ucfg = FP8_UNET
    q = lambda d: {k_: v.to(dtype).float() for k_, v in d.items()}
    usd = q(weights.make_unet_weights(ucfg, perturb_norms=True))
    lsd = q({k_: (v * 5 if k_.endswith("up.weight") else v) for k_, v in weights.make_lora_weights(ucfg).items()})
    ocfg = O.UNetConfig(**dataclasses.asdict(ucfg))
    B, h, w, L = 2, 8, 8, 7
    x = rnd(B, 4, h, w, seed=1)
    ctx = rnd(B, L, ucfg.cross_attention_dim, seed=2)
    added = (rnd(B, ucfg.pooled_dim, seed=3), torch.tensor([[64.0, 64, 0, 0, 64, 64]] * B))
    gout = rnd(B, 4, h, w, seed=4)
    return ucfg, ocfg, usd, lsd, (B, h, w, L), x, ctx, added, gout


def _tagged(holder):
    holder.allow_fp8 = True
    return holder


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fp8_linear_op_matches_oracle_on_identical_inputs(dev, dtype):
    """ops.linear under fp8_forward against the oracle's `_lin` under its fp8 emulation, SAME input bits: same scale,
    same bytes, so the outputs differ by summation order only; the backward is the unquantised product's."""
    M, K, N = 96, 128, 192
    q = lambda t: t.to(dtype).float()
    x, W, b, res, g = q(rnd(M, K, seed=1)), q(rnd(N, K, seed=2) * 0.1), rnd(N, seed=3), q(rnd(M, N, seed=4)), q(rnd(M, N, seed=5))
    xo = x.clone().requires_grad_(True)
    with O.fp8_forward(True):
        yo = O._lin({"blk.weight": W, "blk.bias": b}, "blk", xo) + res
    (yo * g).sum().backward()
    y_exact = torch.nn.functional.linear(x, W, b) + res
    lin = _tagged(ops.FrozenLinear(W, b, dtype, dev))
    xd = x.to(dev, dtype).requires_grad_(True)
    with ops.fp8_forward(True):
        y = ops.linear(xd, lin, residual=res.to(dev, dtype))
    (y.float() * g.to(dev)).sum().backward()
    lim = 2e-5 if dtype == torch.float32 else 8e-3
    assert rel_l2(yo, y_exact) > 1e-2                       # the quantiser is on ...
    assert rel_l2(y, yo) < lim, rel_l2(y, yo)               # ... and the product follows the oracle, not the exact product
    assert rel_l2(xd.grad, xo.grad) < lim
    with ops.fp8_forward(False):                            # outside the context the same holder is exact
        assert rel_l2(ops.linear(xd.detach(), lin, residual=res.to(dev, dtype)), y_exact) < lim


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fp8_lora_group_op_matches_oracle_on_identical_inputs(dev, dtype):
    """q / k / v of one attention: frozen part on fp8 (x quantised once), LoRA part and all gradients unquantised"""
    M, K, N, r = 80, 64, 128, 8
    q = lambda t: t.to(dtype).float()
    x, g = q(rnd(M, K, seed=1)), [q(rnd(M, N, seed=20 + i)) for i in range(3)]
    names = ["a.to_q", "a.to_k", "a.to_v"]
    sd = {n + ".weight": q(rnd(N, K, seed=30 + i) * 0.1) for i, n in enumerate(names)}
    lora = {}
    for i, n in enumerate(names):
        lora[n + ".lora.down.weight"] = q(rnd(r, K, seed=40 + i) * 0.2)
        lora[n + ".lora.up.weight"] = q(rnd(N, r, seed=50 + i) * 0.2)
    lo = {k_: v.clone().requires_grad_(True) for k_, v in lora.items()}
    xo = x.clone().requires_grad_(True)
    with O.fp8_forward(True):
        yo = [O._lin(sd, n, xo, lo) for n in names]
    sum((a * b).sum() for a, b in zip(yo, g)).backward()
    store = ops.LoRAStore([[(n + ".lora.down.weight", n + ".lora.up.weight", lora[n + ".lora.down.weight"],
                             lora[n + ".lora.up.weight"]) for n in names]], dtype, dev)
    lins = [_tagged(l) for l in ops.frozen_linear_group([sd[n + ".weight"] for n in names], [None] * 3, dtype, dev)]
    xd = x.to(dev, dtype).requires_grad_(True)
    with ops.fp8_forward(True):
        ys = ops.lora_group_linear(xd, lins, store.groups[0])
    sum((a.float() * b.to(dev)).sum() for a, b in zip(ys, g)).backward()
    ops.join_side_streams()
    lim = 2e-5 if dtype == torch.float32 else 1e-2
    for a, b in zip(ys, yo):
        assert rel_l2(a, b) < lim, rel_l2(a, b)
    assert rel_l2(xd.grad, xo.grad) < lim
    for n, p in store.params.items():
        assert rel_l2(p.grad, lo[n].grad) < lim, (n, rel_l2(p.grad, lo[n].grad))


@pytest.mark.parametrize("geo", [(2, 8, 64, 128, 1, 1), (1, 8, 128, 64, 2, 1), (2, 4, 64, 64, 1, 2)])
def test_fp8_conv_op_matches_oracle_on_identical_inputs(dev, geo):
    B, H, Cin, Cout, stride, ups = geo
    x, W, b = rnd(B, Cin, H, H, seed=1), rnd(Cout, Cin, 3, 3, seed=2) * 0.05, rnd(Cout, seed=3)
    temb = rnd(B, Cout, seed=4)
    xo = x.clone().requires_grad_(True)
    xu = torch.nn.functional.interpolate(xo, scale_factor=2, mode="nearest") if ups == 2 else xo
    with O.fp8_forward(True):
        yo = O._conv({"blk.weight": W, "blk.bias": b}, "blk", xu, stride=stride) + temb[:, :, None, None]
    g = rnd(*yo.shape, seed=5)
    (yo * g).sum().backward()
    conv = _tagged(ops.FrozenConv(W, b, torch.float32, dev, stride=stride, pad=1))
    xd = tok(x).to(dev).requires_grad_(True)
    with ops.fp8_forward(True):
        y = ops.conv2d(xd, conv, B, H, H, ups=ups, bias2=temb.to(dev))
    (y * tok(g).to(dev)).sum().backward()
    assert rel_l2(y, tok(yo)) < 2e-5, rel_l2(y, tok(yo))
    assert rel_l2(xd.grad, tok(xo.grad)) < 2e-5


def test_unet_fp8_forward_against_oracle_emulation(dev):
    """Whole UNet, every eligible layer on fp8.  Quantisers in series amplify ANY difference between two fp32
    implementations: the first activation that lands on the other side of a rounding boundary (~1e-6 summation-order
    noise is enough) changes its successors by ~1e-3, which then re-rounds per cent of the next quantiser's inputs, and
    so on - after a few layers product and oracle carry independent quantisation noise.  Exact agreement is therefore a
    per-operator property (the tests above: identical input bits -> identical bytes); at network level the checks are
      (1) layers that see bit-identical inputs (the first quantised layers) produce the same scales to 1e-5,
      (2) the same NUMBER of tensors is quantised in product and oracle (same sizes, upsampler inputs apart),
      (3) the product's distance to the oracle emulation stays below the emulation's own distance to the exact network
          (two draws of the same quantisation noise), and the product's distance to the exact network is that of the
          oracle within a factor of two."""
    ucfg, ocfg, usd, lsd, (B, h, w, L), x, ctx, added, gout = _fp8_world(dev)
    o_scales, o_shapes = [], []
    orig = OF.fake_quant

    def spy(t):
        qv, sc = OF.quantize(t)
        if t.shape[0] == B:  # activations (weights are [out, in, ...]; no weight here has leading dim 2)
            o_scales.append(float(sc))
            o_shapes.append(t.numel())
        return OF.dequantize(qv, sc)
    OF.fake_quant = spy
    try:
        with torch.no_grad():
            eo = O.unet_forward(usd, ocfg, x, 417, ctx, lsd, None, added, fp8=True)
    finally:
        OF.fake_quant = orig
    with torch.no_grad():
        e_exact = O.unet_forward(usd, ocfg, x, 417, ctx, lsd, None, added)
    bank = LoRABank(ucfg, lsd, torch.float32, dev)
    unet = UNet(ucfg, usd, torch.float32, dev, bank, fp8_forward=True)
    k = ops.kernels()
    p_scales, p_shapes = [], []
    kq = k.fp8_quantize

    def pspy(t, **kw):
        r = kq(t, **kw)
        p_scales.append(r[1])
        p_shapes.append(t.numel())
        return r
    k.fp8_quantize = pspy
    try:
        with torch.no_grad():
            e, _ = unet(tok(x).to(dev), B, h, w, 417, ctx.reshape(B * L, -1).to(dev).contiguous(), L,
                        added=(added[0].to(dev), added[1].tolist()))
    finally:
        k.fp8_quantize = kq
    p_scales = [float(t.cpu()) for t in p_scales]
    # (2) the oracle quantises the shared input of a q/k/v (k/v) group once per projection, the product once per group
    dedup = [(n_, s_) for i, (n_, s_) in enumerate(zip(o_shapes, o_scales))
             if i == 0 or (n_, s_) != (o_shapes[i - 1], o_scales[i - 1])]
    # (the element counts differ only where the product quantises the input of an upsampler conv BEFORE the fused 2x upsample)
    assert len(dedup) == len(p_shapes) and sum(n_ for n_, _ in dedup) >= sum(p_shapes)
    # (1) down_blocks.0.resnets.0.conv1 / conv2: inputs produced by unquantised layers only
    assert abs(p_scales[0] - dedup[0][1]) < 1e-5 * dedup[0][1]
    # (3)
    cost_o, cost_p, mism = rel_l2(tok(eo), tok(e_exact)), rel_l2(e, tok(e_exact)), rel_l2(e, tok(eo))
    print(f"fp8 UNet: oracle-vs-exact {cost_o:.3e}, product-vs-exact {cost_p:.3e}, product-vs-oracle {mism:.3e}")
    assert cost_o > 1e-2 and 0.5 * cost_o < cost_p < 2.0 * cost_o and mism < 1.2 * cost_o


def test_fp8_forward_touches_only_the_tagged_layers(sim):
    """embeddings, conv_in / conv_out stay exact; a UNet built without the flag never quantises"""
    ucfg, ocfg, usd, lsd, (B, h, w, L), x, ctx, added, gout = _fp8_world(sim)
    calls = []
    k = ops.kernels()
    orig = k.fp8_quantize
    k.fp8_quantize = lambda t, **kw: (calls.append(tuple(t.shape)), orig(t, **kw))[1]
    unet = UNet(ucfg, usd, torch.float32, sim, LoRABank(ucfg, lsd, torch.float32, sim))
    with torch.no_grad():
        unet(tok(x), B, h, w, 417, ctx.reshape(B * L, -1).contiguous(), L, added=(added[0], added[1].tolist()))
    assert calls == []
    unet8 = UNet(ucfg, usd, torch.float32, sim, LoRABank(ucfg, lsd, torch.float32, sim), fp8_forward=True)
    n_weights = len(calls)
    assert n_weights > 20 and not getattr(unet8.conv_in, "allow_fp8") and not getattr(unet8.conv_out, "allow_fp8")
    assert not unet8.t1.allow_fp8 and not unet8.a1.allow_fp8 and getattr(unet8.conv_in, "_w8", None) is None
    with torch.no_grad():
        unet8(tok(x), B, h, w, 417, ctx.reshape(B * L, -1).contiguous(), L, added=(added[0], added[1].tolist()))
    acts = calls[n_weights:]
    assert len(acts) > 20 and all(s[-1] % 64 == 0 for s in acts)  # activations only, contraction length % 64 == 0
    k.fp8_quantize = orig


def _fp8_step_world(dev, dtype):
    """BASELINE config C5 in miniature: SDXL-layout generator with the fp8 forward, SD1.5-layout discriminator, attribute concentration on
    the two-resolution geometry C5 uses -> (trainer, bank, batch, cfg, oracle_world(fp8) -> dict)"""
    from comat_amd.blip import Blip
    from comat_amd.gan import D_sd
    from comat_amd.pipeline import TrainableSDXLPipeline
    from comat_amd.step import CoMatTrainer, StepConfig
    from comat_amd.unet import VAEDecoder
    from oracle import blip as OB
    from oracle import step as OS
    ucfg = FP8_UNET
    vcfg = dataclasses.replace(config.TINY_VAE, scaling_factor=0.13025)
    q = lambda d: {k_: v.to(dtype).float() for k_, v in d.items()}
    up5 = lambda d: {k_: (v * 5 if k_.endswith("up.weight") else v) for k_, v in d.items()}
    usd, vsd = q(weights.make_unet_weights(ucfg, perturb_norms=True)), q(weights.make_vae_weights(vcfg, perturb_norms=True))
    lsd, bsd = q(up5(weights.make_lora_weights(ucfg))), q(weights.make_blip_weights(config.TINY_BLIP, perturb_norms=True))
    dsd = q(weights.make_unet_weights(config.TINY_UNET, seed=77, perturb_norms=True))
    dl = q(up5(weights.make_lora_weights(config.TINY_UNET, seed=78)))
    g = torch.Generator().manual_seed(6)
    r = lambda *s: torch.randn(*s, generator=g)
    head_w, head_b = r(4) * 0.5, r(1) * 0.1
    cfg = StepConfig(resolution=64, total_step=3, K=2, gan_loss=True, attrcon=True, attrcon_train_steps=1,
                     train_layer_ls=("mid_2", "up_2", "up_4"), attn_reses=(4, 2), lr=1e-2, lr_D=1e-2,
                     mask_token_loss_weight=0.5, mask_pixel_loss_weight=0.1)
    bs, L, T = 1, 7, 9
    ids = torch.randint(1, config.TINY_BLIP.vocab_size, (bs, T), generator=g)
    m = np.zeros((2, 64, 64), dtype=bool)
    m[0, 5:30, 8:40] = True
    m[1, 34:60, 20:64] = True
    rq = lambda *s: r(*s).to(dtype).float()
    batch = dict(prompt_embeds=rq(bs, L, ucfg.cross_attention_dim), negative_prompt_embeds=rq(bs, L, ucfg.cross_attention_dim),
                 pooled_prompt_embeds=rq(bs, ucfg.pooled_dim), negative_pooled_prompt_embeds=rq(bs, ucfg.pooled_dim),
                 add_time_ids=(64, 64, 0, 0, 64, 64), gan_null_embeds=rq(bs, L, config.TINY_UNET.cross_attention_dim),
                 latents=r(bs, 4, 8, 8), noises=[r(bs, 4, 8, 8) for _ in range(3)], real_latents=r(bs, 4, 8, 8),
                 blip_input_ids=ids, blip_attention_mask=torch.ones_like(ids), masks=[m], attributes=[[[2, 3], [5]]])

    def oracle_world(fp8):
        return dict(unet=usd, vae=vsd, blip=bsd, d_unet=dsd, ucfg=O.UNetConfig(**dataclasses.asdict(ucfg)),
                    d_ucfg=O.UNetConfig(**dataclasses.asdict(config.TINY_UNET)), vcfg=O.VAEConfig(**dataclasses.asdict(vcfg)),
                    bcfg=OB.BlipConfig(**dataclasses.asdict(config.TINY_BLIP)), fp8_unet=fp8,
                    lora={k_: v.clone().requires_grad_(True) for k_, v in lsd.items()},
                    d_lora={k_: v.clone().requires_grad_(True) for k_, v in dl.items()},
                    head_w=head_w.clone().requires_grad_(True), head_b=head_b.clone().requires_grad_(True))
    bank = LoRABank(ucfg, lsd, dtype, dev)
    unet = UNet(ucfg, usd, dtype, dev, bank, fp8_forward=True)
    pipe = TrainableSDXLPipeline(unet, VAEDecoder(vcfg, vsd, dtype, dev))
    dbank = LoRABank(config.TINY_UNET, dl, dtype, dev)
    disc = D_sd(UNet(config.TINY_UNET, dsd, dtype, dev, dbank), dbank, head_w, head_b)
    trainer = CoMatTrainer(pipe, bank, Blip(config.TINY_BLIP, bsd, dtype, dev), disc, cfg, seed=0)
    return trainer, bank, batch, cfg, oracle_world


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fp8_step_against_oracle_emulation(dev, dtype):
    """BASELINE config C5 in miniature, at STEP level: SDXL-layout generator with the fp8 forward (bf16 / fp32 backward on
    the saved unquantised activations), attribute concentration on the two-resolution geometry C5 uses (mid at the lowest
    map resolution, up at both: `mid_32, up_32, up_64` at 1024^2 -> `mid_2, up_2, up_4` here), SD1.5-layout
    discriminator, one whole optimisation step against the oracle step with its fp8 emulation (oracle/sd.py fp8_forward:
    value of the quantised product, gradient of the unquantised one).
    Tolerance (stated): quantisers in series decorrelate two fp32 implementations (see
    test_unet_fp8_forward_against_oracle_emulation), so the yardstick is the emulation's own distance to the EXACT step:
    every loss term within 1.5x that distance (+5e-3 of its scale: on a single scalar the emulation's own distance can
    be accidentally tiny - measured on MI355X: reward -4.6824 vs -4.6666 where emulation and exact differ by 0.0019), LoRA gradients within 1.5x the emulation-vs-exact
    gradient distance.  bf16 storage compounds its own rounding with the quantisers' (measured on MI355X: token loss 0.323
    vs 0.255, gradients 0.77 vs 0.55 relative): + 0.1 of the scale on loss terms, + the bf16 step bound on gradients - the
    fp32-storage case is the tight one."""
    from oracle import step as OS
    trainer, bank, batch, cfg, oracle_world = _fp8_step_world(dev, dtype)
    ts, crop, acs = [1, 2], (1, 0, 63, 63), [2]
    ref8 = OS.train_step(oracle_world(True), batch, cfg, ts, crop, acs)
    ref = OS.train_step(oracle_world(False), batch, cfg, ts, crop, acs)
    n_q = []
    k = ops.kernels()
    kq = k.fp8_quantize
    k.fp8_quantize = lambda t, **kw: (n_q.append(t.numel()), kq(t, **kw))[1]
    try:
        logs = trainer.train_step(batch, training_steps=ts, crop=crop, attrcon_steps=acs)
    finally:
        k.fp8_quantize = kq
    assert len(n_q) > 40  # the trained and the no-grad generator calls really ran their block layers on fp8 operands
    names = bank.names
    cat = lambda d: torch.cat([d[n].reshape(-1) for n in names])
    g8, gx = cat(ref8["g_grads"]), cat(ref["g_grads"])
    own = rel_l2(g8, gx)  # the emulation's own distance to the exact step
    got = rel_l2(bank.flat_grad, g8)
    line = [f"fp8 step ({dtype}): grads emulation-vs-exact {own:.3e}, product-vs-emulation {got:.3e}"]
    slack = 0.0 if dtype == torch.float32 else 0.27  # bf16 storage: the SDXL-layout step bound of tests/test_step.py
    assert own > 1e-3 and got < 1.5 * own + slack, line
    for key, rk in (("Blip", "Blip"), ("G_loss", "G_loss"), ("D_loss", "D_loss"), ("step_loss", "loss"),
                    ("token_loss", "token_loss"), ("pixel_loss", "pixel_loss")):
        a, b8, bx = float(logs[key]), float(ref8[rk]), float(ref[rk])
        tol_ = 1.5 * abs(b8 - bx) + (5e-3 if dtype == torch.float32 else 1e-1) * max(1.0, abs(b8))
        line.append(f"{key}: product {a:.5f} emulation {b8:.5f} exact {bx:.5f}")
        assert abs(a - b8) <= tol_, line
    print("\n".join(line))


# ---- delayed scaling (ABI 8): one launch per tensor, none where the producer emits the bytes -----------------------------------
@pytest.fixture
def delayed():
    prev = ops.fp8_scaling()
    ops.set_fp8_scaling("delayed")
    yield
    ops.set_fp8_scaling(prev)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n", [8, 1000, 4099, 320 * 4096 + 3])
def test_quantize_scaled_matches_oracle_bit_for_bit_and_tracks_the_abs_max(dev, dtype, n):
    """comat_fp8_quantize_scaled: the bytes under a GIVEN scale (values beyond 448 * scale saturate) and the tensor's abs-max
    folded into the site's running maximum; comat_fp8_scales_update turns the maximum into the oracle's scale and re-arms it"""
    k = ops.kernels()
    x = (rnd(n, seed=n) * 3).to(dtype)
    for frac in (1.0, 0.37):  # the tensor's own scale, and one that saturates its tail
        s = OF.scale_of(x) * frac
        scale = s.reshape(1).to(dev)
        amax = torch.zeros(1, dtype=torch.int32, device=dev)
        q = k.fp8_quantize_scaled(x.to(dev).contiguous(), scale, amax)
        assert torch.equal(q.cpu(), OF.quantize_with_scale(x, s))
        assert float(amax.cpu().view(torch.float32)) == float(x.float().abs().max())
    q2 = k.fp8_quantize_scaled((x * 0.5).to(dev).contiguous(), scale, amax)  # a smaller tensor leaves the maximum alone
    assert float(amax.cpu().view(torch.float32)) == float(x.float().abs().max()) and q2.shape == x.shape
    scales = torch.tensor([7.0, 9.0], device=dev)
    both = torch.cat([amax, torch.zeros(1, dtype=torch.int32, device=dev)])
    k.fp8_scales_update(both, scales, 2)
    assert float(scales[0].cpu()) == float(OF.scale_of(x)) and float(scales[1].cpu()) == 9.0  # an unseen site keeps its scale
    assert int(both.cpu().abs().sum()) == 0


def test_jit_quantize_records_the_abs_max_for_calibration(dev):
    x = rnd(5000, seed=3)
    amax = torch.zeros(1, dtype=torch.int32, device=dev)
    q, s = ops.kernels().fp8_quantize(x.to(dev), amax=amax)
    assert float(amax.cpu().view(torch.float32)) == float(x.abs().max()) and float(s.cpu()) == float(OF.scale_of(x))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(37, 64), (512, 640), (100, 1280), (9, 2048)])
def test_layernorm_emits_the_bytes_of_its_own_output(dev, dtype, shape):
    """comat_layernorm_fwd_q: same y and statistics as comat_layernorm_fwd, q8 = comat_fp8_quantize_scaled(y) bit for bit, and the
    site's running maximum = max |y| - against the oracle's quantiser on the product's y"""
    k = ops.kernels()
    M, Cc = shape
    x = rnd(M, Cc, seed=M).to(dtype).to(dev)
    g, b = (1 + 0.1 * rnd(Cc, seed=1)).to(dev), (0.1 * rnd(Cc, seed=2)).to(dev)
    y0, st0 = torch.empty_like(x), torch.empty((M, 2), device=dev)
    k.layernorm_fwd(x, g, b, y0, st0, M, Cc, 1e-5)
    if not k.layernorm_fwd_q_ok(x):  # rows of more than 256 16-byte vectors (fp32: C > 1024) take the scalar kernel: two calls there
        assert dtype == torch.float32 and Cc > 1024
        return
    y, st, q8 = torch.empty_like(x), torch.empty((M, 2), device=dev), torch.empty((M, Cc), dtype=torch.uint8, device=dev)
    scale = (OF.scale_of(y0.cpu()) * 0.8).reshape(1).to(dev)
    amax = torch.zeros(1, dtype=torch.int32, device=dev)
    k.layernorm_fwd_q(x, g, b, y, st, M, Cc, 1e-5, q8, scale, amax)
    assert torch.equal(y.cpu(), y0.cpu()) and torch.equal(st.cpu(), st0.cpu())
    assert torch.equal(q8.cpu(), OF.quantize_with_scale(y0.cpu(), scale.cpu()[0]))
    assert float(amax.cpu().view(torch.float32)) == float(y0.float().abs().max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("geo", [(2, 1024, 64, 8, True), (1, 4096, 320, 32, True), (2, 400, 128, 32, False)])
def test_groupnorm_emits_the_bytes_of_its_own_output(dev, dtype, geo):
    k = ops.kernels()
    B, HW, Cc, G, silu = geo
    x = rnd(B * HW, Cc, seed=HW).to(dtype).to(dev)
    g, b = (1 + 0.1 * rnd(Cc, seed=1)).to(dev), (0.1 * rnd(Cc, seed=2)).to(dev)
    y0, st0 = torch.empty_like(x), torch.empty((B, G, 2), device=dev)
    k.groupnorm_fwd(x, g, b, y0, st0, B, HW, Cc, G, 1e-5, silu)
    assert k.groupnorm_fwd_q_ok(x, B, HW, Cc, G)
    y, st, q8 = torch.empty_like(x), torch.empty((B, G, 2), device=dev), torch.empty(x.shape, dtype=torch.uint8, device=dev)
    scale = (OF.scale_of(y0.cpu()) * 0.8).reshape(1).to(dev)
    amax = torch.zeros(1, dtype=torch.int32, device=dev)
    k.groupnorm_fwd_q(x, g, b, y, st, B, HW, Cc, G, 1e-5, silu, q8, scale, amax)
    assert torch.equal(y.cpu(), y0.cpu()) and torch.equal(st.cpu(), st0.cpu())
    assert torch.equal(q8.cpu(), OF.quantize_with_scale(y0.cpu(), scale.cpu()[0]))
    assert float(amax.cpu().view(torch.float32)) == float(y0.float().abs().max())


def test_delayed_scaling_site_protocol(dev, delayed):
    """ops.linear under the fp8 forward with delayed scaling: the calibration call quantises under its own abs-max; the next call
    uses THAT scale (not its own) until fp8_end_of_step installs the maximum over the calls since - against the oracle's
    quantiser with explicit scales"""
    lin = _tagged(ops.FrozenLinear(rnd(96, 128, seed=1) * 0.1, rnd(96, seed=2) * 0.1, torch.float32, dev))
    w = lin.w.cpu().float()
    wq = OF.dequantize(*OF.quantize(w))
    x1, x2, x3 = rnd(40, 128, seed=3), rnd(40, 128, seed=4) * 2.5, rnd(40, 128, seed=5) * 0.5

    def ref(x, s):
        return OF.dequantize(OF.quantize_with_scale(x, s), s) @ wq.t() + lin.bias.cpu()
    with torch.no_grad(), ops.fp8_forward(True):
        with ops.fp8_calibration():
            y1 = ops.linear(x1.to(dev), lin)
        ops.fp8_end_of_step()
        s1 = OF.scale_of(x1)
        assert rel_l2(y1, ref(x1, s1)) < 1e-4
        y2 = ops.linear(x2.to(dev), lin)       # 2.5x larger values under the scale of x1: the tail saturates, as the oracle's does
        assert rel_l2(y2, ref(x2, s1)) < 1e-4
        y3 = ops.linear(x3.to(dev), lin)
        assert rel_l2(y3, ref(x3, s1)) < 1e-4
        ops.fp8_end_of_step()                  # max(|x2|, |x3|) = |x2|
        y3b = ops.linear(x3.to(dev), lin)
        assert rel_l2(y3b, ref(x3, OF.scale_of(x2))) < 1e-4


def test_delayed_scaling_unet_matches_the_jit_form_after_calibration(dev, delayed):
    """Whole UNet (every eligible layer on fp8, norms emitting the bytes for the layers they feed): calibrated on an input, the
    delayed form evaluated on THE SAME input uses the same scales as the just-in-time form (abs-max over one call = that call's
    abs-max): same quantisation count, outputs equal up to the rounding-boundary noise the jit test describes"""
    ucfg, ocfg, usd, lsd, (B, h, w, L), x, ctx, added, gout = _fp8_world(dev)
    bank = LoRABank(ucfg, lsd, torch.float32, dev)
    unet = UNet(ucfg, usd, torch.float32, dev, bank, fp8_forward=True)
    args = (tok(x).to(dev), B, h, w, 417, ctx.reshape(B * L, -1).to(dev).contiguous(), L)
    kw = dict(added=(added[0].to(dev), added[1].tolist()))
    k = ops.kernels()
    counts = {"jit": 0, "scaled": 0, "ln": 0, "gn": 0, "fa": 0}

    def wrap(name, key):
        fn = getattr(k, name)

        def f(*a, **kk):
            if key != "fa" or kk.get("q8") is not None:  # the fused attention counts only when it emits the bytes
                counts[key] += 1
            return fn(*a, **kk)
        setattr(k, name, f)
        return fn
    saved = {n_: wrap(n_, key) for n_, key in (("fp8_quantize", "jit"), ("fp8_quantize_scaled", "scaled"),
                                                ("layernorm_fwd_q", "ln"), ("groupnorm_fwd_q", "gn"), ("flash_attn_fwd", "fa"))}
    try:
        with torch.no_grad():
            with ops.fp8_calibration():
                e_cal, _ = unet(*args, **kw)
            n_sites = counts["jit"]
            ops.fp8_end_of_step()
            counts.update(jit=0)
            e_del, _ = unet(*args, **kw)
    finally:
        for n_, fn in saved.items():
            setattr(k, n_, fn)
    assert counts["jit"] == 0 and counts["ln"] > 0
    assert counts["scaled"] + counts["ln"] + counts["gn"] + counts["fa"] == n_sites  # every jit pair became one launch or none
    ops.set_fp8_scaling("jit")
    with torch.no_grad():
        e_jit, _ = unet(*args, **kw)
    print(f"fp8 UNet delayed vs jit: {rel_l2(e_del, e_jit):.3e}; calibration pass vs jit: {rel_l2(e_cal, e_jit):.3e}; "
          f"{n_sites} sites: {counts['scaled']} quantize launches, {counts['ln']} LayerNorm + {counts['gn']} GroupNorm + {counts['fa']} attention "
          "producers")
    assert rel_l2(e_cal, e_jit) < 1e-6   # the calibration pass IS the jit form
    assert rel_l2(e_del, e_jit) < 2e-2   # same scales; only bytes on a rounding boundary may differ (see the jit UNet test)


@pytest.mark.parametrize("shape", [(64, 64, 64, 16), (200, 96, 128, 128), (333, 320, 640, 128), (2048, 1280, 1280, 128), (8192, 640, 640, 32)])
@pytest.mark.parametrize("splits", [0, 3])
def test_fp8_gemm_with_a_bf16_k_tail_matches_oracle(dev, shape, splits):
    """comat_gemm_params::A2k (ABI 8): scale_a * scale_b * (A8 B8^T) + A2 B2^T in ONE launch - the LoRA up projection riding in the
    frozen projection's fp8 product - against the oracle's dequantised product plus the bf16 product; also with the e4m3 product
    cut along k (the tail must be added once, by the first slice)"""
    M, N, K, K2 = shape
    x, w = rnd(M, K, seed=1) * 2.0, rnd(N, K, seed=2) * 0.05
    h, u = (rnd(M, 3 * K2, seed=5) * 0.5).bfloat16(), (rnd(N, K2, seed=6) * 0.2).bfloat16()
    bias, res = rnd(N, seed=3), rnd(M, N, seed=4).bfloat16()
    (xq, sx), (wq, sw) = OF.quantize(x), OF.quantize(w)
    hs = h[:, K2:2 * K2]  # a column slice of the rank-r buffer of a q / k / v group (lda2k = 3 K2)
    ref = (OF.dequantize(xq, sx).double() @ OF.dequantize(wq, sw).double().t() + hs.double() @ u.double().t() + bias.double()
           + res.double())
    k = ops.kernels()
    x8, sxd = k.fp8_quantize(x.to(dev))
    w8, swd = k.fp8_quantize(w.to(dev))
    hd, ud = h.to(dev), u.to(dev)
    y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    if dev.type == "cuda":
        from comat_amd import _hip
        _hip.set_option("g2_splits", splits)
    try:
        k.gemm(x8, w8, y, M, N, K, K, K, N, bias=bias.to(dev), R=res.to(dev), ldr=N, beta=1.0, scales=(sxd, swd),
               ktail=(hd[:, K2:2 * K2], ud, K2, 3 * K2, K2, 0, 0))
    finally:
        if dev.type == "cuda":
            _hip.set_option("g2_splits", 0)
    assert rel_l2(y, ref) < 6e-3, rel_l2(y, ref)
    if dev.type == "cuda":
        assert _hip.last_gemm_kernel() == 3


def test_fp8_batched_gemm_with_per_batch_weight_scales_and_k_tail(dev):
    """q / k / v of one attention in one launch: shared input bytes, a scale per frozen weight (s_scale_b), the three up projections
    as the batched k-tail"""
    G, M, N, K, r = 3, 300, 192, 256, 32
    x = rnd(M, K, seed=1)
    ws = [rnd(N, K, seed=10 + i) * (0.02 * (i + 1)) for i in range(G)]  # different magnitudes: different scales
    h, us = (rnd(M, G * r, seed=5) * 0.5).bfloat16(), (rnd(G, N, r, seed=6) * 0.2).bfloat16()
    xq, sx = OF.quantize(x)
    ref = torch.stack([OF.dequantize(xq, sx).double() @ OF.dequantize(*OF.quantize(ws[i])).double().t()
                       + h[:, i * r:(i + 1) * r].double() @ us[i].double().t() for i in range(G)])
    k = ops.kernels()
    x8, sxd = k.fp8_quantize(x.to(dev))
    w8 = torch.empty((G, N, K), dtype=torch.uint8, device=dev)
    sc = torch.empty(G, device=dev)
    for i in range(G):
        k.fp8_quantize(ws[i].to(dev), out=w8[i], scale=sc[i:i + 1])
    ys = torch.empty((G, M, N), dtype=torch.bfloat16, device=dev)
    k.gemm(x8, w8, ys, M, N, K, K, K, N, batch=(G, 1), sB=(N * K, 0), sC=(M * N, 0), scales=(sxd, sc, 1),
           ktail=(h.to(dev), us.to(dev), r, G * r, r, r, N * r))
    assert rel_l2(ys, ref) < 6e-3, rel_l2(ys, ref)


@pytest.mark.parametrize("need_grad", [False, True])
def test_geglu_epilogue_emits_the_bytes_its_consumer_multiplies(dev, delayed, need_grad):
    """comat_gemm_params::q8 (ABI 8): under the fp8 forward with delayed scaling `ff.net.0.proj` + GEGLU writes the e4m3 bytes of its
    product for `ff.net.2` instead of the bf16 product - the same bits as the two-step form (bf16 product, comat_fp8_quantize_scaled),
    one launch and one [M, D] round trip less; the site's running maximum sees the same abs-max"""
    T = torch.bfloat16
    M, Kd, D, N = 200, 128, 256, 128
    ff1 = _tagged(ops.FrozenGegluLinear(rnd(2 * D, Kd, seed=1) * 0.1, rnd(2 * D, seed=2) * 0.1, T, dev))
    ff2 = _tagged(ops.FrozenLinear(rnd(N, D, seed=3) * 0.1, rnd(N, seed=4) * 0.1, T, dev))
    x = rnd(M, Kd, seed=5).to(T).to(dev)
    k = ops.kernels()
    n_scaled = [0]
    orig = k.fp8_quantize_scaled

    def counted(*a, **kw):
        n_scaled[0] += 1
        return orig(*a, **kw)
    k.fp8_quantize_scaled = counted
    try:
        with ops.fp8_forward(True):
            with torch.no_grad(), ops.fp8_calibration():
                ops.geglu_feed_forward(x, ff1, ff2)
            ops.fp8_end_of_step()
            outs, maxima = [], []
            for fused in (True, False):
                ops.set_geglu_fused(fused)
                xin = x.clone().requires_grad_(need_grad)
                n_scaled[0] = 0
                with torch.set_grad_enabled(need_grad):
                    y = ops.geglu_feed_forward(xin, ff1, ff2)
                outs.append((y.detach().clone(), n_scaled[0]))
                maxima.append(int(ff2._fp8_site[1].cpu()))
                if need_grad:
                    y.backward(torch.ones_like(y))
                    outs[-1] += (xin.grad.clone(),)
                ops.fp8_end_of_step()
    finally:
        ops.set_geglu_fused(True)
        k.fp8_quantize_scaled = orig
    assert torch.equal(outs[0][0], outs[1][0]) and maxima[0] == maxima[1] and maxima[0] != 0
    assert outs[0][1] == 1 and outs[1][1] == 2  # fused: only x is quantised by a launch of its own
    if need_grad:
        assert torch.equal(outs[0][2], outs[1][2])


@pytest.mark.parametrize("geo", [(2, 4, 200, 200, 40), (1, 8, 130, 77, 64), (2, 2, 64, 64, 160)])
def test_fused_attention_emits_the_bytes_of_its_own_output(dev, geo):
    """comat_flash_attn_fwd_q: same O and lse as comat_flash_attn_fwd, q8 = comat_fp8_quantize_scaled(O) bit for bit, abs-max tracked"""
    k = ops.kernels()
    B, H, Nq, Nk, d = geo
    T = torch.bfloat16
    HD = H * d
    q, kk, v = (rnd(B * n_, HD, seed=s_).to(T).to(dev) for n_, s_ in ((Nq, 1), (Nk, 2), (Nk, 3)))
    o0, lse0 = torch.empty_like(q), torch.empty(B, H, Nq, device=dev)
    k.flash_attn_fwd(q, kk, v, o0, lse0, B, H, Nq, Nk, d, HD, HD, HD, HD, d ** -0.5)
    o, lse = torch.empty_like(q), torch.empty(B, H, Nq, device=dev)
    q8 = torch.empty((B * Nq, HD), dtype=torch.uint8, device=dev)
    scale = (OF.scale_of(o0.cpu()) * 0.7).reshape(1).to(dev)
    amax = torch.zeros(1, dtype=torch.int32, device=dev)
    k.flash_attn_fwd(q, kk, v, o, lse, B, H, Nq, Nk, d, HD, HD, HD, HD, d ** -0.5, q8=(q8, scale, amax))
    assert torch.equal(o.cpu(), o0.cpu()) and torch.equal(lse.cpu(), lse0.cpu())
    assert torch.equal(q8.cpu(), OF.quantize_with_scale(o0.cpu(), scale.cpu()[0]))
    assert float(amax.cpu().view(torch.float32)) == float(o0.float().abs().max())


def test_fp8_step_with_delayed_scaling(dev, delayed):
    """C5 in miniature under DELAYED scaling: calibration on the batch (one eager no-grad sampler pass: every site's abs-max over all
    denoise steps), then two optimisation steps - the first under the calibrated scales, the second under the abs-maxima the first one
    recorded.  No just-in-time quantisation after the calibration; the step stays as close to the exact (unquantised) oracle step as
    the just-in-time form is (both are the same network under fp8 noise: the yardstick is the jit step's own distance to the exact one)."""
    from oracle import step as OS
    dtype = torch.float32
    ts, crop, acs = [1, 2], (1, 0, 63, 63), [2]
    trainer, bank, batch, cfg, oracle_world = _fp8_step_world(dev, dtype)
    ref = OS.train_step(oracle_world(False), batch, cfg, ts, crop, acs)
    cat = lambda d_: torch.cat([d_[n].reshape(-1) for n in bank.names])
    gx = cat(ref["g_grads"])
    k = ops.kernels()
    n_jit = [0]
    kq = k.fp8_quantize
    k.fp8_quantize = lambda t, **kw: (n_jit.__setitem__(0, n_jit[0] + 1), kq(t, **kw))[1]
    try:
        assert trainer.fp8_calibrate(batch)
        n_cal = n_jit[0]
        logs1 = trainer.train_step(batch, training_steps=ts, crop=crop, attrcon_steps=acs)
        g1 = bank.flat_grad.detach().clone()
        logs2 = trainer.train_step(batch, training_steps=ts, crop=crop, attrcon_steps=acs)
    finally:
        k.fp8_quantize = kq
    assert n_cal > 40 and n_jit[0] == n_cal  # calibration quantises just in time; the steps never do
    d_del = rel_l2(g1, gx)
    # the jit form of the same step, for the yardstick (a second world: the first one's LoRA factors have been updated)
    ops.set_fp8_scaling("jit")
    trainer_j, bank_j, _, _, _ = _fp8_step_world(dev, dtype)
    trainer_j.train_step(batch, training_steps=ts, crop=crop, attrcon_steps=acs)
    d_jit = rel_l2(bank_j.flat_grad, gx)
    print(f"fp8 step, delayed scaling: LoRA gradient vs exact step {d_del:.3e} (just-in-time scales: {d_jit:.3e}); "
          f"step 1 loss {float(logs1['step_loss']):.4f}, step 2 loss {float(logs2['step_loss']):.4f}")
    assert torch.isfinite(bank.flat_grad).all() and all(torch.isfinite(torch.as_tensor(float(logs2[k_]))) for k_ in ("step_loss", "Blip", "G_loss", "D_loss"))
    assert d_del < 2.0 * d_jit + 0.05
