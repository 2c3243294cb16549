"""Operator parity: every comat_amd op (HIP kernels through the C ABI) against a plain PyTorch fp32 reference of the
same op, forward and backward.  Parametrised over the kernel backend: `sim` (CPU simulator of the ABI, validates the
test + host wiring anywhere) and `hip` (the real kernels on an MI355X, marked gpu).  Tolerances: fp32 storage uses
exact-f32 MFMA -> 2e-4 of the output scale; bf16 storage -> 2e-2 (8 mantissa bits) against the fp32 reference
evaluated on the bf16-rounded inputs."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

from comat_amd import ops
from comat_amd.resize import resize_tables

DTYPES = [torch.float32, torch.bfloat16]


def tol(dtype):
    return 2e-4 if dtype == torch.float32 else 2e-2


def rnd(*shape, dtype=torch.float32, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(*shape, generator=g) * scale
    return x.to(dtype).float()  # value representable in `dtype`, held in fp32


def rel_l2(got, ref):
    got, ref = got.detach().double().cpu().reshape(-1), ref.detach().double().cpu().reshape(-1)
    return float((got - ref).norm() / (ref.norm() + 1e-30))


def check(got, ref, dtype, what="", factor=1.0):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item() + 1e-6
    assert err / scale < tol(dtype) * factor, f"{what}: max err {err:.3e} vs scale {scale:.3e} ({dtype})"


def dv(x, dev, dtype=None, grad=False):
    t = x.detach().to(device=dev, dtype=dtype or x.dtype).contiguous()
    return t.clone().requires_grad_(True) if grad else t


# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("transA", [False, True])
@pytest.mark.parametrize("transB", [False, True])
@pytest.mark.parametrize("shape", [(150, 77, 93), (256, 320, 128), (33, 500, 40), (300, 200, 264), (16, 768, 768), (20, 72, 64)])
def test_gemm_layouts(dev, dtype, transA, transB, shape):
    M, N, K = shape
    A = rnd(M, K, dtype=dtype, seed=1)
    B = rnd(N, K, dtype=dtype, seed=2)
    bias = rnd(N, seed=3)
    bias2 = rnd(3, N, seed=4)
    R = rnd(M, N, dtype=dtype, seed=5)
    rpb = (M + 2) // 3
    ref = 0.5 * (A @ B.t()) + bias + bias2.repeat_interleave(rpb, 0)[:M]
    ref = F.silu(ref) + 2.0 * R
    Ad = dv(A.t() if transA else A, dev, dtype)
    Bd = dv(B.t() if transB else B, dev, dtype)
    C = torch.empty((M, N), dtype=dtype, device=dev)
    ops.kernels().gemm(Ad, Bd, C, M, N, K, M if transA else K, N if transB else K, N, transA=transA, transB=transB,
                       bias=dv(bias, dev), bias2=dv(bias2, dev), rows_per_bias2=rpb, R=dv(R, dev, dtype), ldr=N,
                       alpha=0.5, beta=2.0, act=ops.ACT_SILU)
    check(C, ref, dtype, "gemm")


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_asymmetric_identity(dev, dtype):
    """A = I against an asymmetric B catches a transposed C-write (cdna guide G9)."""
    n = 96
    A = torch.eye(n)
    B = (torch.arange(n * n).reshape(n, n) % 17).float() - 8 + torch.arange(n)[:, None].float() * 0.25
    B = B.to(dtype).float()
    C = torch.empty((n, n), dtype=torch.float32, device=dev)
    ops.kernels().gemm(dv(A, dev, dtype), dv(B, dev, dtype), C, n, n, n, n, n, n)
    check(C, B.t(), dtype, "A=I")


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_batched_heads_in_place(dev, dtype):
    """Two-level batch addressing heads inside [tokens, heads*dim] matrices, fp32 output."""
    B_, H, N, L, d = 2, 3, 70, 45, 40
    q = rnd(B_ * N, H * d, dtype=dtype, seed=1)
    k = rnd(B_ * L, H * d, dtype=dtype, seed=2)
    S = torch.empty((B_, H, N, L), dtype=torch.float32, device=dev)
    HD = H * d
    ops.kernels().gemm(dv(q, dev, dtype), dv(k, dev, dtype), S, N, L, d, HD, HD, L, batch=(B_, H),
                       sA=(N * HD, d), sB=(L * HD, d), sC=(H * N * L, N * L), alpha=0.3)
    ref = 0.3 * torch.einsum("bnhd,blhd->bhnl", q.reshape(B_, N, H, d), k.reshape(B_, L, H, d))
    check(S, ref, dtype, "batched QK^T")


CONV_CASES = [
    # B, H, W, Cin, Cout, k, stride, pad, ups
    (2, 12, 12, 16, 24, 3, 1, 1, 1),
    (1, 9, 7, 4, 40, 3, 1, 1, 1),      # Cin=4: unaligned gather path (conv_in)
    (2, 16, 16, 32, 8, 3, 2, 1, 1),    # Downsample2D; dgrad = transposed gather
    (1, 8, 8, 32, 16, 3, 1, 1, 2),     # Upsample2D: nearest 2x fused
    (2, 10, 10, 24, 136, 1, 1, 0, 1),  # 1x1
    (1, 24, 24, 64, 132, 3, 1, 1, 1),  # several k-tiles / n-tiles
    (2, 16, 16, 32, 64, 3, 2, 1, 1),   # Downsample2D whose dgrad runs on the pipelined kernel (zero-stuffed stride-1 gather)
    (1, 15, 17, 32, 32, 3, 2, 1, 1),   # ... odd sizes: the last source row / column lies beyond the stuffed image
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_fwd_bwd(dev, dtype, case):
    B_, H, W, Cin, Cout, k, stride, pad, ups = case
    x = rnd(B_, Cin, H, W, dtype=dtype, seed=1)
    w = rnd(Cout, Cin, k, k, dtype=dtype, seed=2, scale=1.0 / math.sqrt(Cin * k * k))
    b = rnd(Cout, seed=3)
    temb = rnd(B_, Cout, seed=4)
    conv = ops.FrozenConv(w, b, dtype, dev, stride=stride, pad=pad)
    Ho, Wo = ops.conv_out_hw(conv, H, W, ups)
    res = rnd(B_, Cout, Ho, Wo, dtype=dtype, seed=5)
    gy = rnd(B_, Cout, Ho, Wo, dtype=dtype, seed=6)

    xr = x.clone().requires_grad_(True)
    rr = res.clone().requires_grad_(True)
    xin = F.interpolate(xr, scale_factor=2, mode="nearest") if ups == 2 else xr
    yref = F.conv2d(xin, w, b, stride=stride, padding=pad) + temb[:, :, None, None] + rr
    yref.backward(gy)

    def tok(t):
        return t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])
    xd = dv(tok(x), dev, dtype, grad=True)
    rd = dv(tok(res), dev, dtype, grad=True)
    y = ops.conv2d(xd, conv, B_, H, W, ups=ups, residual=rd, bias2=dv(temb, dev))
    y.backward(dv(tok(gy), dev, dtype))
    check(y, tok(yref), dtype, "conv fwd")
    check(xd.grad, tok(xr.grad), dtype, "conv dgrad")
    check(rd.grad, tok(rr.grad), dtype, "conv residual grad")


@pytest.mark.gpu
def test_strided_conv_dgrad_runs_on_the_pipelined_kernel(hip):
    """comat_conv2d mode 1 (transposed gather), stride 2, bf16, Cin % 32 == 0: served by gemm2_kernel's zero-insertion
    gather, bit-identical in value to the register-staged kernel's result up to fp32 summation order"""
    from comat_amd import _hip
    dtype = torch.bfloat16
    B_, H, W, Cin, Cout = 2, 32, 32, 64, 96
    w = rnd(Cout, Cin, 3, 3, dtype=dtype, seed=2, scale=1.0 / math.sqrt(Cin * 9))
    conv = ops.FrozenConv(w, None, dtype, hip, stride=2, pad=1)
    g = dv(rnd(B_ * 16 * 16, Cout, dtype=dtype, seed=3), hip, dtype)
    k = ops.kernels()
    outs = []
    for use in (1, 0):
        _hip.set_option("gemm2", use)
        dx = torch.empty(B_ * H * W, Cin, dtype=dtype, device=hip)
        k.conv2d(g, conv.wd, dx, B_, 16, 16, Cout, H, W, Cin, 3, 3, 2, 1, mode=1)
        assert _hip.last_gemm_kernel() == (1 if use else 0)
        outs.append(dx.float())
    _hip.set_option("gemm2", 1)
    assert rel_l2(outs[0], outs[1]) < 1e-2  # two bf16 roundings of fp32 sums taken in different orders
    gy = g.float().cpu().reshape(B_, 16, 16, Cout).permute(0, 3, 1, 2)
    ref = F.conv_transpose2d(gy, w, stride=2, padding=1, output_padding=1)
    check(outs[0].cpu().reshape(B_, H, W, Cin).permute(0, 3, 1, 2), ref, dtype, "strided dgrad", factor=2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("silu", [False, True])
@pytest.mark.parametrize("shape", [(2, 50, 32, 8), (1, 130, 320, 32), (2, 64, 80, 8)])
def test_groupnorm(dev, dtype, silu, shape):
    B_, HW, C, G = shape
    x = rnd(B_, HW, C, dtype=dtype, seed=1) * 2 + 0.7
    gamma, beta = rnd(C, seed=2) * 0.5 + 1, rnd(C, seed=3) * 0.3
    gy = rnd(B_ * HW, C, dtype=dtype, seed=4)
    xr = x.clone().requires_grad_(True)
    yr = F.group_norm(xr.permute(0, 2, 1), G, gamma, beta, eps=1e-5).permute(0, 2, 1)
    if silu:
        yr = F.silu(yr)
    yr = yr.reshape(B_ * HW, C)
    yr.backward(gy)
    xd = dv(x.reshape(B_ * HW, C), dev, dtype, grad=True)
    y = ops.group_norm(xd, dv(gamma, dev), dv(beta, dev), B_, HW, G=G, eps=1e-5, silu=silu)
    y.backward(dv(gy, dev, dtype))
    check(y, yr, dtype, "gn fwd")
    check(xd.grad, xr.grad.reshape(B_ * HW, C), dtype, "gn bwd", factor=2)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 4096, 320, 32), (2, 64, 1280, 32), (1, 16384, 128, 32), (2, 50, 32, 8), (1, 130, 320, 32),
                                   (2, 1024, 640, 32), (2, 4096, 640, 32), (2, 4096, 960, 32), (2, 256, 2560, 32),
                                   (3, 77, 24, 8)])
def test_groupnorm_two_launch_forms(hip, dtype, shape, default_opts):
    """option norm_fused: 3 (default) = ONE launch, a workgroup per (sample, group) holding the group in registers (the
    shapes cover its three register variants, forward and backward capacity limits - beyond them the call takes the
    three-launch form - and a group with an odd channel count); 1 = the last-arriving statistics block finalises, 2 = the
    prologue of the apply kernel does.  Same statistics up to the summation order of the partial sums: outputs and
    gradients agree with the three-launch form to fp32 rounding, and with the torch reference as in test_groupnorm."""
    B_, HW, C, G = shape
    x = rnd(B_, HW, C, dtype=dtype, seed=1) * 2 + 0.7
    gamma, beta = rnd(C, seed=2) * 0.5 + 1, rnd(C, seed=3) * 0.3
    gy = rnd(B_ * HW, C, dtype=dtype, seed=4)
    xr = x.clone().requires_grad_(True)
    yr = F.silu(F.group_norm(xr.permute(0, 2, 1), G, gamma, beta, eps=1e-5).permute(0, 2, 1)).reshape(B_ * HW, C)
    gy2 = rnd(B_ * HW, C, dtype=dtype, seed=5)  # cotangent of the bypass branch: the `add` operand of the backward kernel
    (yr * gy).sum().backward()
    xr.grad += gy2.reshape(B_, HW, C)
    res = []
    vec_ok = C % (4 if dtype == torch.float32 else 8) == 0  # what the multi-launch forms need (else: one launch or an error)
    for mode in (0, 1, 2, 3, 3, 4, 5):  # 4 / 5: one launch where it pays, else the two-launch ticket / finalize-in-apply form (round 6)
        if mode not in (3, 4, 5) and not vec_ok:
            res.append(None)
            continue
        _set_opts(norm_fused=mode)
        xd = dv(x.reshape(B_ * HW, C), hip, dtype, grad=True)
        y, xa = ops.group_norm_fork(xd, dv(gamma, hip), dv(beta, hip), B_, HW, G=G, eps=1e-5, silu=True)
        ((y.float() * dv(gy, hip).float()).sum() + (xa.float() * dv(gy2, hip).float()).sum()).backward()
        res.append((y.detach().float(), xd.grad.float()))
        check(y, yr, dtype, f"gn fwd (norm_fused={mode})")
        check(xd.grad, xr.grad.reshape(B_ * HW, C), dtype, f"gn bwd (norm_fused={mode})", factor=2)
    lim = 2e-6 if dtype == torch.float32 else 1e-2  # bf16: an output may flip by one ulp
    for mode in (1, 2, 3, 5, 6):
        if res[0] is None:
            continue
        for a, b_, name in zip(res[0], res[mode], ("y", "dx")):
            assert rel_l2(b_, a) < lim, f"norm_fused={mode}: {name} differs from the three-launch form by {rel_l2(b_, a):.2e}"
    assert torch.equal(res[3][0], res[4][0]) and torch.equal(res[3][1], res[4][1])  # run-to-run bit-identical


@pytest.mark.gpu
@pytest.mark.parametrize("B", [4, 12])
def test_groupnorm_capture_with_a_real_batch(hip, B):
    """a capture on the package's capture stream with B >= 4 samples (per-GPU batch >= 2 under CFG
    doubling, the reference's train_batch_size 4 / 6) and the UNet's 32 groups.  Nothing ever runs eagerly on that stream,
    so its GroupNorm workspace is the one prepare_stream() made: it must hold B * G pairs without growing inside the
    capture (the former 2 MiB floor held 127).  Replay equals the eager launch bit for bit, forward and backward, for the
    three-launch form (HW = 1024) and the one-launch form (HW = 64)."""
    dtype = torch.bfloat16
    G = 32
    for HW, Cc in ((1024, 320), (64, 640)):
        x = dv(rnd(B * HW, Cc, dtype=dtype, seed=B), hip, dtype)
        gy = dv(rnd(B * HW, Cc, dtype=dtype, seed=B + 1), hip, dtype)
        gam, bet = dv(rnd(Cc, seed=3), hip, torch.float32), dv(rnd(Cc, seed=4), hip, torch.float32)

        def run(xin):
            y = ops.group_norm(xin, gam, bet, B, HW, G=G, eps=1e-5, silu=True)
            (dx,) = torch.autograd.grad(y, xin, gy)
            return y, dx
        ye, dxe = run(x.clone().requires_grad_(True))
        torch.cuda.synchronize()
        xs = x.clone().requires_grad_(True)
        g = torch.cuda.CUDAGraph()
        with ops.graph_capture(g, stream=ops.capture_stream(hip)):
            yg, dxg = run(xs)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(ye, yg) and torch.equal(dxe, dxg), f"B={B} HW={HW}: replayed GroupNorm differs from eager"


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(37, 96), (8, 1280), (5, 70)])
def test_layernorm(dev, dtype, shape):
    M, C = shape
    x = rnd(M, C, dtype=dtype, seed=1) * 1.5 - 0.4
    gamma, beta = rnd(C, seed=2) * 0.5 + 1, rnd(C, seed=3) * 0.3
    gy = rnd(M, C, dtype=dtype, seed=4)
    xr = x.clone().requires_grad_(True)
    yr = F.layer_norm(xr, (C,), gamma, beta, eps=1e-5)
    yr.backward(gy)
    xd = dv(x, dev, dtype, grad=True)
    y = ops.layer_norm(xd, dv(gamma, dev), dv(beta, dev), eps=1e-5)
    y.backward(dv(gy, dev, dtype))
    check(y, yr, dtype, "ln fwd")
    check(xd.grad, xr.grad, dtype, "ln bwd", factor=2)


def ref_attention(q, k, v, B_, Nq, Nk, H, d, causal, key_mask):
    qh = q.reshape(B_, Nq, H, d).permute(0, 2, 1, 3)
    kh = k.reshape(B_, Nk, H, d).permute(0, 2, 1, 3)
    vh = v.reshape(B_, Nk, H, d).permute(0, 2, 1, 3)
    s = (qh @ kh.transpose(-1, -2)) * d ** -0.5
    if causal:
        m = torch.ones(Nq, Nk, dtype=torch.bool).tril(diagonal=Nk - Nq)
        s = s.masked_fill(~m, float("-inf"))
    if key_mask is not None:
        s = s.masked_fill(~key_mask.bool()[:, None, None, :], float("-inf"))
    p = torch.softmax(s, -1)
    o = (p @ vh).permute(0, 2, 1, 3).reshape(B_ * Nq, H * d)
    return o, p


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [
    (2, 70, 70, 2, 40, False, False),   # self-attention, head dim 40 (SD1.5 level 0)
    (2, 64, 77, 8, 8, False, False),    # cross-attention to 77 text tokens, with a loss on the captured map
    (1, 10, 10, 3, 16, True, False),    # BLIP decoder causal self-attention
    (2, 6, 20, 2, 16, False, True),     # key padding mask
    (1, 200, 136, 1, 64, False, False),
])
def test_attention(dev, dtype, cfg):
    B_, Nq, Nk, H, d, causal, masked = cfg
    q, k, v = (rnd(B_ * n, H * d, dtype=dtype, seed=s) for n, s in ((Nq, 1), (Nk, 2), (Nk, 3)))
    go = rnd(B_ * Nq, H * d, dtype=dtype, seed=4)
    gp = rnd(B_, H, Nq, Nk, dtype=dtype, seed=5) * 0.1
    km = None
    if masked:
        km = torch.ones(B_, Nk, dtype=torch.int8)
        km[0, Nk - 5:] = 0
        km[1, Nk - 1:] = 0
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    o_ref, p_ref = ref_attention(qr, kr, vr, B_, Nq, Nk, H, d, causal, km)
    ((o_ref * go).sum() + (p_ref * gp).sum()).backward()
    qd, kd, vd = (dv(t, dev, dtype, grad=True) for t in (q, k, v))
    o, p = ops.attention(qd, kd, vd, B_, Nq, Nk, H, d, causal=causal, key_mask=None if km is None else km.to(dev))
    ((o.float() * dv(go, dev)).sum() + (p.float() * dv(gp, dev)).sum()).backward()
    check(o, o_ref, dtype, "attn out")
    check(p, p_ref, dtype, "attn probs")
    assert torch.allclose(p.float().sum(-1).cpu(), torch.ones(B_, H, Nq), atol=2e-2 if dtype == torch.bfloat16 else 1e-5)
    check(qd.grad, qr.grad, dtype, "dQ", factor=3)
    check(kd.grad, kr.grad, dtype, "dK", factor=3)
    check(vd.grad, vr.grad, dtype, "dV", factor=3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [
    (2, 4096, 4096, 8, 40),   # SD1.5 self-attention at 64x64 latents (bf16 run is the full-size property check)
    (2, 70, 70, 2, 40), (1, 300, 77, 8, 80), (2, 64, 200, 2, 160), (1, 577, 577, 3, 64), (2, 16, 37, 2, 16),
    (1, 130, 33, 1, 32), (1, 5, 577, 2, 96),
    (2, 1024, 77, 8, 40),     # cross-attention at 32x32 latents: dK/dV pass split over query ranges
])
def test_flash_attention(dev, dtype, cfg):
    B_, Nq, Nk, H, d = cfg
    if Nq == 4096 and (dtype == torch.float32 or dev.type == "cpu"):
        pytest.skip("full-size case runs in bf16 on the GPU only")
    q, k, v = (rnd(B_ * n, H * d, dtype=dtype, seed=s) for n, s in ((Nq, 1), (Nk, 2), (Nk, 3)))
    go = rnd(B_ * Nq, H * d, dtype=dtype, seed=4)
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    o_ref, _ = ref_attention(qr, kr, vr, B_, Nq, Nk, H, d, False, None)
    (o_ref * go).sum().backward()
    qd, kd, vd = (dv(t, dev, dtype, grad=True) for t in (q, k, v))
    o, p = ops.attention(qd, kd, vd, B_, Nq, Nk, H, d, need_probs=False)
    assert p is None
    (o.float() * dv(go, dev)).sum().backward()
    check(o, o_ref, dtype, "flash out")
    check(qd.grad, qr.grad, dtype, "flash dQ", factor=3)
    check(kd.grad, kr.grad, dtype, "flash dK", factor=3)
    check(vd.grad, vr.grad, dtype, "flash dV", factor=3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(130, 64, 8), (70, 40, 4), (257, 96, 16)])
def test_gemm_segments(dev, dtype, shape):
    """K-segmented GEMM (ragged M / N / K_s, strided operands, bias + residual epilogue) against a plain matmul."""
    M, N, r = shape
    Ks = [72, 3 * r, 33 if dtype == torch.float32 else 40, r]
    k = ops.kernels()
    acc = torch.zeros(M, N)
    segs = []
    for i, K in enumerate(Ks):
        lda, ldb = K + (8 if i % 2 else 0), K + (16 if i == 2 else 0)
        A = rnd(M, lda, dtype=dtype, seed=10 + i)
        B = rnd(N, ldb, dtype=dtype, seed=20 + i, scale=K ** -0.5)
        acc = acc + A[:, :K] @ B[:, :K].t()
        segs.append((dv(A, dev, dtype), dv(B, dev, dtype), K, lda, ldb))
    bias, R = rnd(N, seed=3), rnd(M, N, dtype=dtype, seed=4)
    for ns in (1, 2, 4):
        ref = 0.5 * sum(((s[0].float().cpu()[:, :s[2]] @ s[1].float().cpu()[:, :s[2]].t()) for s in segs[:ns]),
                        torch.zeros(M, N)) + bias + 2.0 * R
        out = torch.empty((M, N), dtype=dtype, device=dev)
        k.gemm_segments(segs[:ns], out, M, N, N, bias=dv(bias, dev), R=dv(R, dev, dtype), ldr=N, alpha=0.5, beta=2.0)
        check(out, ref, dtype, f"gemm_segments nseg={ns}")
    # long K: exercises the split-K path of the segmented kernel
    A1, B1 = rnd(64, 2048, dtype=dtype, seed=31, scale=0.3), rnd(48, 2048, dtype=dtype, seed=32, scale=0.3)
    A2, B2 = rnd(64, 520, dtype=dtype, seed=33, scale=0.3), rnd(48, 520, dtype=dtype, seed=34, scale=0.3)
    out = torch.empty((64, 48), dtype=torch.float32, device=dev)
    k.gemm_segments([(dv(A1, dev, dtype), dv(B1, dev, dtype), 2048, 2048, 2048),
                     (dv(A2, dev, dtype), dv(B2, dev, dtype), 520, 520, 520)], out, 64, 48, 48)
    check(out, A1 @ B1.t() + A2 @ B2.t(), dtype, "gemm_segments split-K")
    # batched: 3 problems in one launch — shared first A operand, column-sliced second A operand, stacked B operands
    Gb, Mb, Nb, K1, rb = 3, M, 72, 40, 8
    X, H = rnd(Mb, K1, dtype=dtype, seed=41), rnd(Mb, Gb * rb, dtype=dtype, seed=42)
    Wb, Ub = rnd(Gb, Nb, K1, dtype=dtype, seed=43, scale=0.2), rnd(Gb, Nb, rb, dtype=dtype, seed=44, scale=0.2)
    Rb = rnd(Gb, Mb, Nb, dtype=dtype, seed=45)
    out = torch.empty((Gb, Mb, Nb), dtype=dtype, device=dev)
    Hd, Wd, Ud = dv(H, dev, dtype), dv(Wb, dev, dtype), dv(Ub, dev, dtype)
    k.gemm_segments([(dv(X, dev, dtype), Wd, K1, K1, K1, 0, Nb * K1), (Hd, Ud, rb, Gb * rb, rb, rb, Nb * rb)], out, Mb, Nb,
                    Nb, R=dv(Rb, dev, dtype), ldr=Nb, beta=1.0, batch=Gb, sC=Mb * Nb, sR=Mb * Nb)
    ref = torch.stack([X @ Wb[i].t() + H[:, i * rb:(i + 1) * rb] @ Ub[i].t() + Rb[i] for i in range(Gb)])
    check(out, ref, dtype, "gemm_segments batched")


def test_gemm_tt_grouped(dev):
    """comat_gemm_tt_grouped: C_p += A_p^T B_p for many independent k-major problems in a few launches (the LoRA weight
    gradients, training_utils/pipeline.py:84-115) against fp32 matmuls of the same bf16 operands: ragged tiles, K that is
    no multiple of the k-tile (zero-page rows), K < one k-tile, strided operands, more problems than one launch holds."""
    dtype = torch.bfloat16
    k = ops.kernels()
    shapes = [(128, 320, 2048), (8, 8, 1), (136, 264, 77), (384, 64, 154), (64, 1280, 512), (320, 128, 4100), (16, 24, 31)]
    shapes += [(8 * (1 + i % 5), 8 * (1 + i % 3), 33 + 17 * i) for i in range(60)]
    probs, refs = [], []
    for i, (M, N, K) in enumerate(shapes):
        lda, ldb, ldc = M + (8 if i % 3 == 1 else 0), N + (16 if i % 4 == 2 else 0), N + (4 if i % 2 else 0)
        A = rnd(K, lda, dtype=dtype, seed=100 + i, scale=0.5)
        B = rnd(K, ldb, dtype=dtype, seed=300 + i, scale=0.5)
        C0 = rnd(M, ldc, seed=500 + i)
        Cd = dv(C0, dev)
        probs.append((dv(A, dev, dtype), dv(B, dev, dtype), Cd, M, N, K, lda, ldb, ldc))
        ref = C0.clone()
        ref[:, :N] += A[:, :M].t() @ B[:, :N]
        refs.append(ref)
        assert k.tt_group_ok(*probs[-1])
    k.gemm_tt_grouped(probs)
    for i, (pr, ref) in enumerate(zip(probs, refs)):
        check(pr[2], ref, torch.float32, f"tt_grouped problem {i} {shapes[i]}")  # padding columns untouched as well
    k.gemm_tt_grouped(probs[:3])  # accumulates
    for i in range(3):
        M, N, K = shapes[i]
        ref = refs[i].clone()
        ref[:, :N] += probs[i][0].float().cpu()[:, :M].t() @ probs[i][1].float().cpu()[:, :N]
        check(probs[i][2], ref, torch.float32, f"tt_grouped second call {i}")
    if dev.type == "cuda":
        from comat_amd import _hip
        assert _hip.last_gemm_kernel() == 4
        # bit-reproducible: the same call on the same inputs gives the same bits
        outs = []
        for _ in range(2):
            Cs = [torch.zeros_like(pr[2]) for pr in probs]
            k.gemm_tt_grouped([(pr[0], pr[1], c) + tuple(pr[3:]) for pr, c in zip(probs, Cs)])
            outs.append(torch.cat([c.reshape(-1) for c in Cs]))
        assert torch.equal(outs[0], outs[1])


def test_weight_gradient_queue(dev):
    """ops' deferred weight-gradient queue: problems are handed over in groups; a problem whose output is already in the
    pending group starts a new group (the two launches stay in stream order), everything is launched by the join."""
    dtype = torch.bfloat16
    M, N, K = 16, 8, 40
    A = [dv(rnd(K, M, dtype=dtype, seed=i), dev, dtype) for i in range(3)]
    B = [dv(rnd(K, N, dtype=dtype, seed=10 + i), dev, dtype) for i in range(3)]
    C1, C2 = torch.zeros(M, N, device=dev), torch.zeros(M, N, device=dev)
    ops.flush_weight_grads()
    calls = []
    real = ops.kernels().gemm_tt_grouped
    ops.kernels().gemm_tt_grouped = lambda probs: (calls.append(len(probs)), real(probs))[1]
    try:
        # enqueue outside of a backward pass: bypass the autograd callback and join by hand
        orig, ops._queue_join = ops._queue_join, lambda: None
        ops._tt_enqueue(dev, [(A[0], B[0], C1, M, N, K, M, N, N), (A[1], B[1], C2, M, N, K, M, N, N)], (A, B))
        ops._tt_enqueue(dev, [(A[2], B[2], C1, M, N, K, M, N, N)], (A, B))  # same output: flushes the first two
        assert calls == [2]
        ops.join_side_streams()
        assert calls == [2, 1]
    finally:
        ops._queue_join = orig
        ops.kernels().gemm_tt_grouped = real
    f = lambda t: t.float().cpu()
    check(C1, f(A[0]).t() @ f(B[0]) + f(A[2]).t() @ f(B[2]), torch.float32, "queue C1")
    check(C2, f(A[1]).t() @ f(B[1]), torch.float32, "queue C2")


@pytest.mark.parametrize("dtype", DTYPES)
def test_transpose_cast_tiles(dev, dtype):
    src = rnd(5000, seed=1)
    mats = [(7, 45, 33), (1600, 64, 40), (4200, 8, 96)]  # (offset, rows, cols)
    tiles, doff, refs = [], 0, []
    for so, rows, cols in mats:
        for r0 in range(0, rows, 32):
            for c0 in range(0, cols, 32):
                tiles.append((so, doff, rows, cols, r0, c0))
        refs.append((doff, src[so:so + rows * cols].view(rows, cols).t().contiguous()))
        doff += rows * cols
    dst = torch.zeros(doff, dtype=dtype, device=dev)
    ops.kernels().transpose_cast_tiles(dv(src, dev), dst, torch.tensor(tiles, dtype=torch.int64).to(dev))
    for o, ref in refs:
        assert torch.equal(dst[o:o + ref.numel()].cpu().view(ref.shape), ref.to(dtype))


@pytest.fixture
def train_merged_default():
    yield
    ops.set_train_merged(os.environ.get("COMAT_TRAIN_MERGED", "1") != "0")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("G", [1, 3])
@pytest.mark.parametrize("merged", [True, False])
def test_lora_group_linear(dev, dtype, G, merged, train_merged_default):
    """Projections sharing one input, each frozen Linear + LoRA branch (+ residual for a single projection):
    outputs, dx, dresidual and the fp32 LoRA gradients accumulated in place into the store's flat buffer - in the merged-weight
    form of round 5 (one plain GEMM forward, dx through the transposed merged weight, factor gradients on the side) and in the
    low-rank form of rounds 1-4, against the same torch reference."""
    ops.set_train_merged(merged)
    M, K, r = 130, 64, 8
    Ns = [96, 64, 80][:G]
    x = rnd(M, K, dtype=dtype, seed=1)
    xr = x.clone().requires_grad_(True)
    res = rnd(M, Ns[0], dtype=dtype, seed=6) if G == 1 else None
    rr = res.clone().requires_grad_(True) if G == 1 else None
    spec, lins, refs, loss = [], [], [], 0
    for i, N in enumerate(Ns):
        w = rnd(N, K, dtype=dtype, seed=20 + i, scale=K ** -0.5)
        b = rnd(N, seed=30 + i)
        down, up = rnd(r, K, seed=40 + i, scale=r ** -0.5), rnd(N, r, seed=50 + i, scale=0.2)
        gy = rnd(M, N, dtype=dtype, seed=60 + i)
        dr = down.to(dtype).float().clone().requires_grad_(True)  # the kernels see the compute-dtype copies
        ur = up.to(dtype).float().clone().requires_grad_(True)
        yr = xr @ w.t() + b + (xr @ dr.t()) @ ur.t() + (rr if G == 1 else 0)
        loss = loss + (yr * gy).sum()
        refs.append((yr, dr, ur, gy))
        spec.append((f"p{i}.down", f"p{i}.up", down, up))
        lins.append(ops.FrozenLinear(w, b, dtype, dev))
    loss.backward()
    store = ops.LoRAStore([spec], dtype, dev)
    assert store.names == [f"p{i}.down" for i in range(G)] + [f"p{i}.up" for i in range(G)]
    xd = dv(x, dev, dtype, grad=True)
    rd = dv(res, dev, dtype, grad=True) if G == 1 else None
    for rep in range(2):  # second pass: gradients ACCUMULATE in the flat buffer
        ys = ops.lora_group_linear(xd, lins, store.groups[0], residual=rd)
        sum((y.float() * dv(gy, dev, dtype).float()).sum() for y, (_, _, _, gy) in zip(ys, refs)).backward()
    for i, (y, (yr, dr, ur, _)) in enumerate(zip(ys, refs)):
        check(y, yr, dtype, f"lora group y{i}")
        gd, gu = store.params[f"p{i}.down"].grad, store.params[f"p{i}.up"].grad
        assert gd.dtype == torch.float32 and gu.dtype == torch.float32
        check(gd, 2 * dr.grad, dtype, f"lora d_down{i}", factor=2)
        check(gu, 2 * ur.grad, dtype, f"lora d_up{i}", factor=2)
    check(xd.grad, 2 * xr.grad, dtype, "lora group dx", factor=2)
    if G == 1:
        check(rd.grad, 2 * rr.grad, dtype, "lora dres")
    if G == 3:
        # batched paths: co-allocated frozen weights (one launch for the group's forward) and output gradients at a
        # constant spacing in one buffer (dQ/dK/dV of the fused attention backward: one launch for u, one for dU)
        Nq = 64
        ws = [rnd(Nq, K, dtype=dtype, seed=70 + i, scale=K ** -0.5) for i in range(3)]
        ups = [rnd(Nq, r, seed=80 + i, scale=0.2) for i in range(3)]
        downs = [rnd(r, K, seed=90 + i, scale=r ** -0.5) for i in range(3)]
        gl = ops.frozen_linear_group(ws, [None] * 3, dtype, dev)
        assert ops._uniform_stride([l.w for l in gl]) == Nq * K
        st2 = ops.LoRAStore([[(f"d{i}", f"u{i}", downs[i], ups[i]) for i in range(3)]], dtype, dev)
        x2r = x.clone().requires_grad_(True)
        d2 = [d.to(dtype).float().clone().requires_grad_(True) for d in downs]
        u2 = [u.to(dtype).float().clone().requires_grad_(True) for u in ups]
        gbuf = rnd(3, M, Nq, dtype=dtype, seed=99)
        y2r = [x2r @ ws[i].t() + (x2r @ d2[i].t()) @ u2[i].t() for i in range(3)]
        torch.autograd.backward(y2r, [gbuf[i] for i in range(3)])
        x2 = dv(x, dev, dtype, grad=True)
        y2 = ops.lora_group_linear(x2, gl, st2.groups[0])
        assert ops._uniform_stride(list(y2)) == M * Nq
        gd_buf = dv(gbuf, dev, dtype)
        torch.autograd.backward(y2, list(gd_buf.unbind(0)))
        for i in range(3):
            check(y2[i], y2r[i], dtype, f"batched group y{i}")
            check(st2.params[f"d{i}"].grad, d2[i].grad, dtype, f"batched group d_down{i}", factor=2)
            check(st2.params[f"u{i}"].grad, u2[i].grad, dtype, f"batched group d_up{i}", factor=2)
        check(x2.grad, x2r.grad, dtype, "batched group dx", factor=2)
    # frozen factors (the generator-side pass through the discriminator): dx only, gradient buffer untouched
    store.zero_grad()
    store.set_requires_grad(False)
    xd2 = dv(x, dev, dtype, grad=True)
    ys = ops.lora_group_linear(xd2, lins, store.groups[0])
    sum((y.float() * dv(gy, dev, dtype).float()).sum() for y, (_, _, _, gy) in zip(ys, refs)).backward()
    ref_dx = xr.grad - (rr.grad.sum() * 0 if G == 1 else 0)
    check(xd2.grad, ref_dx, dtype, "lora group dx (frozen factors)", factor=2)
    assert float(store.flat_grad.abs().max()) == 0.0
    # updated parameters are picked up after mark_updated()
    store.flat.mul_(0.5)
    store.mark_updated()
    y0 = ops.lora_group_linear(xd2.detach(), lins, store.groups[0])[0]
    (yr, dr, ur, _) = refs[0]
    half = x @ lins[0].w.float().cpu().t() + lins[0].bias.cpu() + (x @ (0.5 * dr.detach()).to(dtype).float().t()) @ \
        (0.5 * ur.detach()).to(dtype).float().t()
    check(y0, half, dtype, "lora group after update", factor=2)
    # plain frozen linear, with and without input grad
    y2 = ops.linear(xd.detach(), lins[0], act=ops.ACT_GELU)
    check(y2, F.gelu(x @ lins[0].w.float().cpu().t() + lins[0].bias.cpu()), dtype, "linear+gelu")


@pytest.mark.parametrize("dtype", DTYPES)
def test_elementwise(dev, dtype):
    x = rnd(37, 24, dtype=dtype, seed=1) * 2
    y = rnd(37, 24, dtype=dtype, seed=2)
    g = rnd(37, 24, dtype=dtype, seed=3)
    for name, fn, rf in (("silu", ops.silu, F.silu), ("gelu", ops.gelu, F.gelu)):
        xr = x.clone().requires_grad_(True)
        rf(xr).backward(g)
        xd = dv(x, dev, dtype, grad=True)
        out = fn(xd)
        out.backward(dv(g, dev, dtype))
        check(out, rf(x), dtype, name)
        check(xd.grad, xr.grad, dtype, name + " bwd")
    xd, yd = dv(x, dev, dtype, grad=True), dv(y, dev, dtype, grad=True)
    out = ops.add(xd, yd, 0.5, -2.0)
    out.backward(dv(g, dev, dtype))
    check(out, 0.5 * x - 2 * y, dtype, "axpby")
    check(xd.grad, 0.5 * g, dtype, "axpby dx")
    check(yd.grad, -2 * g, dtype, "axpby dy")
    check(ops.affine(dv(x, dev, dtype), 0.5, 0.5), 0.5 * x + 0.5, dtype, "affine")
    # geglu
    xr = x.clone().requires_grad_(True)
    ref = xr[:, :12] * F.gelu(xr[:, 12:])
    ref.backward(g[:, :12])
    xd = dv(x, dev, dtype, grad=True)
    out = ops.geglu(xd)
    out.backward(dv(g[:, :12], dev, dtype))
    check(out, ref, dtype, "geglu")
    check(xd.grad, xr.grad, dtype, "geglu bwd")
    # concat (cols / rows) + cast
    xd, yd = dv(x, dev, dtype, grad=True), dv(y[:, :8], dev, dtype, grad=True)
    cc = ops.concat_cols(xd, yd)
    gg = rnd(37, 32, dtype=dtype, seed=9)
    cc.backward(dv(gg, dev, dtype))
    check(cc, torch.cat([x, y[:, :8]], 1), dtype, "concat_cols")
    check(xd.grad, gg[:, :24], dtype, "concat_cols ga")
    check(yd.grad, gg[:, 24:], dtype, "concat_cols gb")
    cr = ops.concat_rows(dv(x, dev, dtype), dv(y, dev, dtype))
    check(cr, torch.cat([x, y], 0), dtype, "concat_rows")
    c32 = ops.cast(dv(x, dev, dtype), torch.float32)
    assert c32.dtype == torch.float32
    check(c32, x, dtype, "cast")
    check(ops.add_rowvec(dv(x, dev, dtype), dv(y[0], dev, dtype)), x + y[0], dtype, "add_rowvec")
    # NCHW <-> tokens
    img = rnd(2, 3, 5, 7, dtype=dtype, seed=11)
    t = ops.nchw_to_tokens(dv(img, dev, dtype))
    check(t, img.permute(0, 2, 3, 1).reshape(-1, 3), dtype, "to tokens")
    check(ops.tokens_to_nchw(t, 2, 5, 7), img, dtype, "to nchw")
    # sumpool (adjoint of nearest 2x)
    u = rnd(2, 6, 8, 5, dtype=dtype, seed=12)
    sp = torch.empty((2 * 3 * 4, 5), dtype=dtype, device=dev)
    ops.kernels().sumpool2x2(dv(u.reshape(-1, 5), dev, dtype), sp, 2, 3, 4, 5)
    check(sp, u.reshape(2, 3, 2, 4, 2, 5).sum(dim=(2, 4)).reshape(-1, 5), dtype, "sumpool")


@pytest.mark.parametrize("dtype", DTYPES)
def test_cfg_ddpm_step(dev, dtype):
    n = 2 * 4 * 8 * 8
    x, z = rnd(n, seed=1), rnd(n, seed=2)
    e = rnd(2 * n, dtype=dtype, seed=3)
    s, cx, ce, sg = 7.5, 0.93, -0.21, 0.05
    xr, er = x.clone().requires_grad_(True), e.clone().requires_grad_(True)
    ref = cx * xr + ce * (er[:n] + s * (er[n:] - er[:n])) + sg * z
    g = rnd(n, seed=4)
    ref.backward(g)
    xd, ed = dv(x, dev, grad=True), dv(e, dev, dtype, grad=True)
    out = ops.cfg_ddpm_step(xd, ed, dv(z, dev), s, cx, ce, sg)
    out.backward(dv(g, dev))
    check(out, ref, torch.float32, "ddpm fwd")
    check(xd.grad, xr.grad, torch.float32, "ddpm dx")
    check(ed.grad, er.grad, dtype, "ddpm deps")


@pytest.mark.parametrize("dtype", DTYPES)
def test_resample_and_patchify(dev, dtype):
    B_, C, Hf, crop, out = 2, 3, 40, (1, 2, 37, 37), (24, 24)
    img = rnd(B_, C, Hf, Hf, dtype=dtype, seed=1)
    mean, std = torch.tensor([0.48, 0.45, 0.40]), torch.tensor([0.27, 0.26, 0.28])
    ir = img.clone().requires_grad_(True)
    y0, x0, ch, cw = crop
    ref = F.interpolate(ir[:, :, y0:y0 + ch, x0:x0 + cw], size=out, mode="bicubic", antialias=True,
                        align_corners=False)
    ref = (ref - mean[None, :, None, None]) / std[None, :, None, None]
    g = rnd(B_, C, *out, dtype=dtype, seed=2)
    ref.backward(g)
    fwd, bwd = resize_tables(Hf, Hf, crop, out, "bicubic")
    tab = ops.ResampleTables(fwd, bwd, Hf, Hf, out[0], out[1], dev)

    def tok(t):
        return t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])
    idv = dv(tok(img), dev, dtype, grad=True)
    y = ops.resample(idv, tab, B_, C, scale=dv(1 / std, dev), shift=dv(-mean / std, dev))
    y.backward(dv(tok(g), dev, dtype))
    check(y, tok(ref), dtype, "resample fwd")
    check(idv.grad, tok(ir.grad), dtype, "resample bwd")
    # patchify == unfold with (ky, kx, c) inner order
    P = 8
    pr = rnd(B_, 24, 24, C, dtype=dtype, seed=3)
    pd = dv(pr.reshape(-1, C), dev, dtype, grad=True)
    pt = ops.patchify(pd, B_, 24, 24, C, P)
    refp = pr.reshape(B_, 3, P, 3, P, C).permute(0, 1, 3, 2, 4, 5).reshape(B_ * 9, P * P * C)
    check(pt, refp, dtype, "patchify")
    gp = rnd(*refp.shape, dtype=dtype, seed=4)
    pt.backward(dv(gp, dev, dtype))
    refg = gp.reshape(B_, 3, 3, P, P, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, C)
    check(pd.grad, refg, dtype, "patchify bwd")
    ids = torch.tensor([3, 0, 7, 7, 1])
    table = rnd(9, 16, dtype=dtype, seed=5)
    check(ops.embedding(ids.to(dev), dv(table, dev, dtype)), table[ids], dtype, "embedding")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("ls", [0.0, 0.1])
def test_cross_entropy(dev, dtype, ls):
    T, V = 12, 1531
    z = rnd(T, V, dtype=dtype, seed=1, scale=3.0)
    labels = torch.randint(0, V, (T,), generator=torch.Generator().manual_seed(2))
    labels[:4] = -100
    labels[9] = -100
    zr = z.clone().requires_grad_(True)
    ref = F.cross_entropy(zr, labels, ignore_index=-100, label_smoothing=ls)
    (2.5 * ref).backward()
    zd = dv(z, dev, dtype, grad=True)
    loss, logp = ops.cross_entropy(zd, labels.to(dev), -100, ls)
    (2.5 * loss).backward()
    check(loss, ref, torch.float32, "ce loss", factor=5)
    lp_ref = torch.log_softmax(z, -1).gather(1, labels.clamp(min=0)[:, None])[:, 0] * (labels >= 0)
    check(logp, lp_ref, torch.float32, "token log-probs", factor=5)
    check(zd.grad, zr.grad, dtype, "ce bwd")


@pytest.mark.parametrize("dtype", DTYPES)
def test_disc_head(dev, dtype):
    bs, pps = 2, 8 * 8
    x = rnd(bs * pps, 4, dtype=dtype, seed=1)
    w, b = rnd(4, seed=2), rnd(1, seed=3)
    target = torch.tensor([0.0, 1.0])
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    ref = F.binary_cross_entropy_with_logits(xr @ wr + br, target.repeat_interleave(pps))
    (1.7 * ref).backward()
    xd = dv(x, dev, dtype, grad=True)
    wd, bd = dv(w, dev, grad=True), dv(b, dev, grad=True)
    loss = ops.disc_head_loss(xd, wd, bd, dv(target, dev), pps)
    (1.7 * loss).backward()
    check(loss, ref, torch.float32, "bce", factor=5)
    check(xd.grad, xr.grad, dtype, "bce dx")
    check(wd.grad, wr.grad, torch.float32, "bce dw", factor=20)
    check(bd.grad, br.grad, torch.float32, "bce db", factor=20)


@pytest.mark.parametrize("dtype", DTYPES)
def test_attnmap_gather(dev, dtype):
    h, res, L = 4, 12, 77
    npix = res * res
    a = torch.softmax(rnd(h, npix, L, seed=1), -1).to(dtype).float()
    mask = (rnd(2, npix, seed=2) > 0).float()
    tok_idx = torch.tensor([2, 3, 6, 7, 3], dtype=torch.int32)
    tok_obj = torch.tensor([0, 0, 1, 1, 1], dtype=torch.int32)
    gn, gd, ga = rnd(h, 5, seed=3), rnd(h, 5, seed=4), rnd(5, npix, seed=5)
    ar = a.clone().requires_grad_(True)
    sel = ar[:, :, tok_idx.long()]
    num_r = torch.einsum("hpt,tp->ht", sel, mask[tok_obj.long()])
    den_r = sel.sum(1)
    avg_r = sel.mean(0).t()
    ((num_r * gn).sum() + (den_r * gd).sum() + (avg_r * ga).sum()).backward()
    ad = dv(a, dev, dtype, grad=True)
    num, den, avg = ops.attnmap_gather(ad, dv(mask, dev), tok_idx.to(dev), tok_obj.to(dev))
    ((num * dv(gn, dev)).sum() + (den * dv(gd, dev)).sum() + (avg * dv(ga, dev)).sum()).backward()
    check(num, num_r, torch.float32, "num", factor=5)
    check(den, den_r, torch.float32, "den", factor=5)
    check(avg, avg_r, torch.float32, "avg", factor=5)
    check(ad.grad, ar.grad, dtype, "damap")


def test_adamw_with_clip(dev):
    n = 5000
    p0, g0 = rnd(n, seed=1), rnd(n, seed=2) * 3
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pr], lr=5e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    p = dv(p0, dev)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step in range(1, 4):
        g = g0 * step
        pr.grad = g.clone()
        torch.nn.utils.clip_grad_norm_([pr], 0.1)
        opt.step()
        gd = dv(g, dev)
        nsq = torch.zeros(1, device=dev)
        ops.kernels().sumsq(gd, n, nsq)
        check(nsq, (g.double() ** 2).sum().float().reshape(1), torch.float32, "sumsq", factor=5)
        ops.kernels().adamw(p, gd, m, v, n, 5e-3, 0.9, 0.999, 1e-8, 1e-2, step, nsq, 0.1)
    check(p, pr, torch.float32, "adamw", factor=0.5)
    # a non-finite gradient (norm inf / NaN) skips the update: parameters and moments stay bit-identical
    before = (p.clone(), m.clone(), v.clone())
    for bad in (float("inf"), float("nan")):
        gb = g0.clone()
        gb[17] = bad
        gd = dv(gb, dev)
        nsq = torch.zeros(1, device=dev)
        ops.kernels().sumsq(gd, n, nsq)
        assert not torch.isfinite(nsq).item()
        ops.kernels().adamw(p, gd, m, v, n, 5e-3, 0.9, 0.999, 1e-8, 1e-2, 4, nsq, 0.1)
        assert all(torch.equal(a, b) for a, b in zip(before, (p, m, v)))


# the library's defaults for the options whose default moved in round 4 (runtime.hip)
DEFAULT_OPTS = dict(flash_xcd=1, g2_order=2, gemm3=1, norm_fused=5)


def _set_opts(**kw):
    from comat_amd import _hip
    for k_, v_ in kw.items():
        _hip.set_option(k_, v_)


@pytest.fixture
def default_opts():
    """restore the library's kernel-selection options after a test that forces variants"""
    yield
    _set_opts(gemm2=1, gemm2_tt=1, g2_cfg=0, g2_splits=0, force_splits=0, flash_trim=1, flash_tr=1, flash_kt=4, flash_merge=1, flash_xcd=DEFAULT_OPTS['flash_xcd'],
              g2_order=DEFAULT_OPTS['g2_order'], norm_fused=DEFAULT_OPTS['norm_fused'], gemm3=DEFAULT_OPTS['gemm3'], g3_cfg=0)


G2_GEMMS = [  # (M, N, K, batch): k-contiguous bf16 problems the pipelined kernel takes (K % 32 == 0)
    (300, 200, 320, 1), (8192, 320, 320, 1), (577, 1024, 4096, 1), (128, 1280, 1280, 1), (257, 136, 64, 1),
    (2048, 384, 640, 3), (64, 64, 32, 1), (1000, 4, 96, 1), (512, 1280, 10240, 1)]


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13])  # 8..11: 128-byte k-tiles; 12 (256 x 256), 13 (256 x 128, 4 waves): never split
@pytest.mark.parametrize("splits", [0, 1, 3])
def test_gemm2_matches_reference(hip, cfg, splits, default_opts):
    """gemm2.hip (LDS-DMA pipelined kernel) under every block tile (g2_cfg) and split count against an fp32 reference
    and against the general 64x64 kernel on the same bf16 operands (they differ only in fp32 summation order): ragged
    M / N, bias, per-row-group bias, residual in bf16 and fp32, activation, fp32 and bf16 outputs, a batched launch,
    in-place accumulation (R == C)."""
    dtype = torch.bfloat16
    k = ops.kernels()
    for (M, N, K_, nb) in G2_GEMMS:
        A = rnd(nb, M, K_, dtype=dtype, seed=1, scale=0.5)
        B = rnd(nb, N, K_, dtype=dtype, seed=2, scale=0.5)
        bias = rnd(N, seed=3)
        rows_b2 = 64 if M % 64 == 0 else M
        bias2 = rnd(M // rows_b2, N, seed=5)
        for out_dt, r_dt, act in ((torch.bfloat16, torch.bfloat16, ops.ACT_NONE), (torch.float32, torch.float32, ops.ACT_SILU),
                                  (torch.bfloat16, None, ops.ACT_GELU)):
            R = rnd(nb, M, N, dtype=r_dt, seed=4) if r_dt is not None else None
            Ad, Bd = dv(A, hip, dtype), dv(B, hip, dtype)
            Rd = dv(R, hip, r_dt) if R is not None else None
            use_b = nb == 1
            outs = []
            for g2 in (1, 0):
                _set_opts(gemm2=g2, g2_cfg=cfg, g2_splits=splits)
                out = torch.full((nb, M, N), float("nan"), dtype=out_dt, device=hip)
                k.gemm(Ad, Bd, out, M, N, K_, K_, K_, N, bias=dv(bias, hip) if use_b else None,
                       bias2=dv(bias2, hip) if use_b else None, rows_per_bias2=rows_b2 if use_b else 0, R=Rd, ldr=N,
                       beta=0.5 if R is not None else 0.0, alpha=0.25, act=act, batch=(nb, 1), sA=(M * K_, 0),
                       sB=(N * K_, 0), sC=(M * N, 0), sR=(M * N, 0))
                outs.append(out)
            pre = 0.25 * torch.einsum("bmk,bnk->bmn", A, B)
            if use_b:
                pre = pre + bias + bias2.repeat_interleave(rows_b2, dim=0)
            ref = {ops.ACT_NONE: lambda t: t, ops.ACT_SILU: F.silu, ops.ACT_GELU: F.gelu}[act](pre)
            if R is not None:
                ref = ref + 0.5 * R
            what = f"gemm2 cfg={cfg} splits={splits} M={M} N={N} K={K_} b={nb} out={out_dt} act={act}"
            check(outs[0], ref, dtype, what)
            check(outs[0], outs[1], dtype, what + " vs general kernel", factor=0.5 if out_dt == torch.bfloat16 else 0.02)
    # in-place accumulation into an fp32 buffer (the LoRA weight-gradient pattern), twice: R aliases C
    _set_opts(gemm2=1, g2_cfg=cfg, g2_splits=splits)
    A, B = rnd(320, 4096, dtype=dtype, seed=7, scale=0.3), rnd(128, 4096, dtype=dtype, seed=8, scale=0.3)
    acc = torch.zeros((320, 128), dtype=torch.float32, device=hip)
    for _ in range(2):
        k.gemm(dv(A, hip, dtype), dv(B, hip, dtype), acc, 320, 128, 4096, 4096, 4096, 128, R=acc, ldr=128, beta=1.0)
    check(acc, 2 * (A @ B.t()), dtype, f"gemm2 accumulate cfg={cfg} splits={splits}")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("g2", [1, 0])
def test_gemm_tail_columns(hip, dtype, g2, default_opts):
    """comat_gemm_params::epi2 = 4: the last n2 columns of the product come from a second B matrix and go to a second output
    (h = s x D^T riding in the projection's launch, u = s g U in its data-gradient's): single and batched problems, bias +
    residual on the main part only, ragged M, a strided C2 (columns i r .. of a [M, G r] buffer), against fp32 references
    and against the two-launch form (pipelined kernel off / fp32 mode: the library's fallback)."""
    k = ops.kernels()
    _set_opts(gemm2=g2, g2_cfg=0, g2_splits=0)
    for (M, N1, K_, n2, G) in ((8192, 320, 320, 128, 1), (2048, 640, 640, 128, 3), (300, 200, 1280, 128, 1), (154, 640, 768, 128, 2),
                               (512, 1280, 1280, 128, 1), (130, 96, 64, 8, 2)):
        A = rnd(M, K_, dtype=dtype, seed=1, scale=0.5)
        B = rnd(G, N1, K_, dtype=dtype, seed=2, scale=0.5)
        B2 = rnd(G * n2, K_, dtype=dtype, seed=3, scale=0.5)
        bias, R = (rnd(N1, seed=4), rnd(M, N1, dtype=dtype, seed=5)) if G == 1 else (None, None)
        Ad, Bd, B2d = dv(A, hip, dtype), dv(B, hip, dtype), dv(B2, hip, dtype)
        out = torch.full((G, M, N1), float("nan"), dtype=dtype, device=hip)
        h = torch.full((M, G * n2), float("nan"), dtype=dtype, device=hip)
        k.gemm(Ad, Bd, out, M, N1 + n2, K_, K_, K_, N1, batch=(G, 1), sA=(0, 0), sB=(N1 * K_, 0), sC=(M * N1, 0),
               bias=dv(bias, hip) if bias is not None else None, R=dv(R, hip, dtype) if R is not None else None, ldr=N1,
               beta=0.5 if R is not None else 0.0, tail=(B2d, h, n2, G * n2, n2 * K_, n2, 0.75))
        ref = torch.einsum("mk,gnk->gmn", A, B)
        if G == 1:
            ref = ref + bias + 0.5 * R
        what = f"tail columns gemm2={g2} {dtype} M={M} N1={N1} K={K_} n2={n2} G={G}"
        check(out, ref, dtype, what)
        check(h, 0.75 * (A @ B2.t()), dtype, what + " (tail)")


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13])  # 8..11: 128-byte k-tiles; 12 (256 x 256), 13 (256 x 128, 4 waves): never split
@pytest.mark.parametrize("splits", [0, 2])
def test_gemm2_segments_and_conv(hip, cfg, splits, default_opts):
    """K-segmented products (frozen + low-rank, batched q/k/v launch) and implicit-GEMM convs (stride 1 / 2, fused 2x
    upsample, padding taps, ragged M and Cout, time-embedding bias, residual) of the pipelined kernel, forward and the
    data-gradient through ops.conv2d, against fp32 references."""
    dtype = torch.bfloat16
    k = ops.kernels()
    _set_opts(gemm2=1, g2_cfg=cfg, g2_splits=splits)
    tokf = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])
    for (M, N, K1, K2, G) in ((8192, 320, 320, 128, 1), (577, 200, 1280, 128, 1), (2048, 640, 640, 128, 3), (130, 96, 64, 32, 2)):
        A1 = rnd(M, K1, dtype=dtype, seed=31, scale=0.3)
        W = rnd(G, N, K1, dtype=dtype, seed=32, scale=0.3)
        H_ = rnd(M, G * K2, dtype=dtype, seed=33, scale=0.3)
        U = rnd(G, N, K2, dtype=dtype, seed=34, scale=0.3)
        out = torch.full((G, M, N), float("nan"), dtype=dtype, device=hip)
        Wd, Ud, Hd = dv(W, hip, dtype), dv(U, hip, dtype), dv(H_, hip, dtype)
        k.gemm_segments([(dv(A1, hip, dtype), Wd, K1, K1, K1, 0, N * K1), (Hd, Ud, K2, G * K2, K2, K2, N * K2)], out, M, N, N,
                        batch=G, sC=M * N)
        ref = torch.stack([A1 @ W[i].t() + H_[:, i * K2:(i + 1) * K2] @ U[i].t() for i in range(G)])
        check(out, ref, dtype, f"gemm2 segments cfg={cfg} splits={splits} M={M} N={N} G={G}")
    for (Bn, H, W_, Cin, Cout, stride, ups, res) in ((2, 16, 16, 64, 96, 1, 1, True), (1, 64, 64, 320, 320, 1, 1, False),
                                                    (2, 13, 9, 32, 72, 1, 1, False), (1, 16, 16, 64, 136, 2, 1, False),
                                                    (1, 8, 8, 96, 64, 1, 2, True), (2, 8, 8, 1280, 128, 1, 1, False)):
        x = rnd(Bn, Cin, H, W_, dtype=dtype, seed=5)
        w = rnd(Cout, Cin, 3, 3, dtype=dtype, seed=6, scale=(9 * Cin) ** -0.5)
        b = rnd(Cout, seed=7)
        b2 = rnd(Bn, Cout, seed=9)
        conv = ops.FrozenConv(w, b, dtype, hip, stride=stride, pad=1)
        xd = dv(tokf(x), hip, dtype, grad=True)
        xr = x.clone().requires_grad_(True)
        xin = F.interpolate(xr, scale_factor=2, mode="nearest") if ups == 2 else xr
        yr = F.conv2d(xin, w, b, stride=stride, padding=1) + b2[:, :, None, None]
        rr = rnd(*yr.shape, dtype=dtype, seed=10) if res else None
        y = ops.conv2d(xd, conv, Bn, H, W_, ups=ups, bias2=dv(b2, hip),
                       residual=dv(tokf(rr), hip, dtype) if res else None)
        if res:
            yr = yr + rr
        g = rnd(*yr.shape, dtype=dtype, seed=8)
        yr.backward(g)
        y.backward(dv(tokf(g), hip, dtype))
        what = f"gemm2 conv cfg={cfg} splits={splits} {Bn}x{H}x{W_} {Cin}->{Cout} s={stride} ups={ups}"
        check(y, tokf(yr), dtype, what)
        check(xd.grad, tokf(xr.grad), dtype, what + " dgrad", factor=2)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("splits", [2, 5, 16])
def test_inlaunch_split_k_general_kernel(hip, dtype, splits, default_opts):
    """Split-K of the general 64x64 kernel: the last-arriving block of every tile sums the slices in slice order inside
    the launch (no reduce kernel).  Every operand layout, ragged shapes, repeated launches on the same workspace (the
    ticket counters are re-armed by the kernel), bit-reproducible results, conv and K-segmented variants."""
    k = ops.kernels()
    tokf = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])
    _set_opts(gemm2=0, force_splits=splits)
    for (M, N, K_, tA, tB) in ((320, 128, 8192, True, True), (130, 70, 4100, False, False), (512, 1280, 1280, False, False),
                               (64, 64, 1000, False, True), (200, 136, 2055, True, False), (16, 768, 768, False, False)):
        A = rnd(*((K_, M) if tA else (M, K_)), dtype=dtype, seed=1, scale=0.5)
        B = rnd(*((K_, N) if tB else (N, K_)), dtype=dtype, seed=2, scale=0.5)
        bias, R = rnd(N, seed=3), rnd(M, N, dtype=dtype, seed=4)
        ref = 0.25 * ((A.t() if tA else A) @ (B if tB else B.t())) + bias + R
        outs = []
        for rep in range(3):
            out = torch.full((M, N), float("nan"), dtype=torch.float32, device=hip)
            k.gemm(dv(A, hip, dtype), dv(B, hip, dtype), out, M, N, K_, M if tA else K_, N if tB else K_, N, transA=tA,
                   transB=tB, bias=dv(bias, hip), R=dv(R, hip, dtype), ldr=N, beta=1.0, alpha=0.25)
            check(out, ref, dtype, f"split-K gemm splits={splits} M={M} N={N} K={K_} tA={tA} tB={tB} rep={rep}")
            outs.append(out)
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), "split-K result is not bit-reproducible"
    for (Bn, H, W_, Cin, Cout, stride) in ((2, 8, 8, 320, 200, 1), (1, 16, 16, 128, 136, 2), (1, 8, 8, 20, 64, 1)):
        x = rnd(Bn, Cin, H, W_, dtype=dtype, seed=5)
        w = rnd(Cout, Cin, 3, 3, dtype=dtype, seed=6, scale=(9 * Cin) ** -0.5)
        b = rnd(Cout, seed=7)
        conv = ops.FrozenConv(w, b, dtype, hip, stride=stride, pad=1)
        xd = dv(tokf(x), hip, dtype, grad=True)
        y = ops.conv2d(xd, conv, Bn, H, W_)
        xr = x.clone().requires_grad_(True)
        yr = F.conv2d(xr, w, b, stride=stride, padding=1)
        g = rnd(*yr.shape, dtype=dtype, seed=8)
        yr.backward(g)
        y.backward(dv(tokf(g), hip, dtype))
        check(y, tokf(yr), dtype, f"split-K conv splits={splits} {Bn}x{H}x{W_} {Cin}->{Cout} s={stride}")
        check(xd.grad, tokf(xr.grad), dtype, f"split-K conv dgrad splits={splits}", factor=2)
    A1, B1 = rnd(512, 1280, dtype=dtype, seed=31, scale=0.3), rnd(200, 1280, dtype=dtype, seed=32, scale=0.3)
    A2, B2 = rnd(512, 136, dtype=dtype, seed=33, scale=0.3), rnd(200, 136, dtype=dtype, seed=34, scale=0.3)
    out = torch.empty((512, 200), dtype=torch.float32, device=hip)
    k.gemm_segments([(dv(A1, hip, dtype), dv(B1, hip, dtype), 1280, 1280, 1280),
                     (dv(A2, hip, dtype), dv(B2, hip, dtype), 136, 136, 136)], out, 512, 200, 200)
    check(out, A1 @ B1.t() + A2 @ B2.t(), dtype, f"split-K gemm_segments splits={splits}")


def test_adamw_step_count_on_device(dev):
    """FlatAdamW keeps the number of APPLIED updates in device memory: a skipped (non-finite) step does not advance the
    bias correction, and the skip is visible in the counters."""
    from comat_amd.step import FlatAdamW
    n = 3000
    p0, g0 = rnd(n, seed=1), rnd(n, seed=2)
    pr = p0.clone().requires_grad_(True)
    ref = torch.optim.AdamW([pr], lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    p, g = dv(p0, dev), torch.zeros(n, device=dev)
    opt = FlatAdamW([(p, g)], 1e-2, (0.9, 0.999), 1e-8, 1e-2, 0.0)
    for step in range(1, 5):
        if step == 2:  # a non-finite gradient in between: skipped, counted, and NOT a bias-correction step
            g.copy_(dv(g0, dev))
            g[5] = float("nan")
            opt.step()
            assert opt.counters.tolist() == [1, 1]
        g.copy_(dv(g0 * step, dev))
        pr.grad = (g0 * step).clone()
        ref.step()
        opt.step()
    assert opt.counters.tolist() == [4, 1] and opt.t == 4
    check(p, pr, torch.float32, "adamw with a skipped step", factor=0.5)


@pytest.mark.parametrize("dtype", DTYPES)
def test_attnmap_gather_many_tokens(dev, dtype):
    """more than 32 attribute / noun tokens in one prompt (the reference's get_grounding_loss_by_layer takes any number
    up to the 77 text positions, attn_utils/tc_loss_utils.py:104-167)"""
    heads, res, L, n_tok, n_obj = 2, 8, 77, 45, 3
    npix = res * res
    amap = torch.softmax(rnd(heads, npix, L, seed=1), dim=-1).to(dtype)
    mask = (rnd(n_obj, npix, seed=2) > 0).float()
    g = torch.Generator().manual_seed(3)
    tok_idx = torch.randint(1, L, (n_tok,), generator=g, dtype=torch.int32)
    tok_obj = torch.randint(0, n_obj, (n_tok,), generator=g, dtype=torch.int32)
    ad = dv(amap, dev, dtype, grad=True)
    num, den, avg = ops.attnmap_gather(ad, dv(mask, dev), tok_idx.to(dev), tok_obj.to(dev))
    af = amap.float()
    sel = af[:, :, tok_idx.long()]                                  # [heads, npix, n_tok]
    num_r = (sel * mask[tok_obj.long()].t()[None]).sum(1)
    den_r = sel.sum(1)
    avg_r = sel.mean(0).t()
    check(num, num_r, torch.float32, "num", factor=5)
    check(den, den_r, torch.float32, "den", factor=5)
    check(avg, avg_r, torch.float32, "avg", factor=5)
    (num.sum() + den.sum() + avg.sum()).backward()
    assert torch.isfinite(ad.grad.float()).all()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [(2, 1024, 1024, 8, 40), (1, 300, 77, 8, 80), (2, 70, 70, 2, 40), (1, 130, 33, 1, 48),
                                 (1, 577, 577, 3, 64), (2, 64, 200, 2, 160), (2, 16, 37, 2, 16)])
def test_flash_trim_is_bit_identical(hip, dtype, cfg, default_opts):
    """flash_trim skips only MFMA steps whose operands are zero padding, flash_tr only changes how the k-major operand
    tiles are staged in LDS (transposed image, wide reads): outputs and gradients of every combination (the default is
    both on) must be bit-identical to the plain kernels."""
    B, Nq, Nk, H, d = cfg
    q, k_, v = (rnd(B * n, H * d, dtype=dtype, seed=i) for i, n in ((1, Nq), (2, Nk), (3, Nk)))
    g = rnd(B * Nq, H * d, dtype=dtype, seed=4)
    res = []
    _set_opts(flash_kt=1)  # the one-tile kernels: the two-tile ones (flash_kt >= 2) add their products in another order
    for trim, tr in ((0, 0), (1, 0), (0, 2), (1, 2), (1, 1)):  # tr: 0 never, 2 always, 1 by head dim (the default)
        _set_opts(flash_trim=trim, flash_tr=tr)
        qd, kd, vd = (dv(t, hip, dtype, grad=True) for t in (q, k_, v))
        o, _ = ops.attention(qd, kd, vd, B, Nq, Nk, H, d, need_probs=False)
        o.backward(dv(g, hip, dtype))
        res.append((o.detach(), qd.grad, kd.grad, vd.grad))
    for i, variant in enumerate(res[1:], 1):
        for a, b, name in zip(res[0], variant, ("O", "dQ", "dK", "dV")):
            assert torch.equal(a, b), f"variant {i}: {name} differs from the default kernels"


@pytest.mark.gpu
def test_workspaces_are_created_before_a_capture_not_inside_it(hip):
    """A split-K workspace carries ticket counters that must be zero before their first use.  Created inside a capture it
    would be zeroed by a node of that one graph: a SECOND graph captured on the same stream and replayed first would run on
    whatever the memory holds.  So creation during a capture raises, `prepare_stream()` creates the workspaces
    eagerly, and then two graphs captured on one stream give the right result in either replay order."""
    from comat_amd import _hip
    k = _hip.HipKernels()
    dtype = torch.bfloat16
    M, N, K = 64, 64, 4096  # few tiles, long contraction: split along k (tickets + slabs in the workspace)
    A, B = dv(rnd(M, K, dtype=dtype, seed=1, scale=0.1), hip, dtype), dv(rnd(N, K, dtype=dtype, seed=2, scale=0.1), hip, dtype)
    ref = A.float() @ B.float().t()
    st = torch.cuda.Stream()
    g0 = torch.cuda.CUDAGraph()
    C0 = torch.empty(M, N, device=hip)
    with pytest.raises(RuntimeError, match="must exist before the capture"):
        with torch.cuda.graph(g0, stream=st):
            k.gemm(A, B, C0, M, N, K, K, K, N)
    st2 = torch.cuda.Stream()  # (the failed capture may have left `st` in capture mode)
    with torch.cuda.stream(st2):
        assert k.prepare_stream(hip) and not k.prepare_stream(hip)  # created once, idempotent
    torch.cuda.synchronize()
    graphs, outs = [], []
    for _ in range(2):
        g, C = torch.cuda.CUDAGraph(), torch.full((M, N), float("nan"), device=hip)
        with torch.cuda.graph(g, stream=st2):
            k.gemm(A, B, C, M, N, K, K, K, N)
        graphs.append(g)
        outs.append(C)
    for i in (1, 0, 1):  # the graph captured SECOND replays first
        outs[i].fill_(float("nan"))
        graphs[i].replay()
        torch.cuda.synchronize()
        check(outs[i], ref.cpu(), torch.float32, f"graph {i}", factor=50)  # bf16 products, fp32 sums: ~1e-2 of the scale


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(2, 1024, 1024, 8, 40), (1, 300, 200, 8, 80), (2, 70, 130, 2, 40), (1, 577, 577, 3, 64),
                                 (1, 130, 65, 1, 48), (2, 1024, 77, 8, 80), (1, 96, 4096, 2, 40), (1, 4096, 100, 2, 40)])
def test_flash_two_tiles_per_iteration(hip, cfg, default_opts):
    """option flash_kt = 5 (2: forward only, 3: + dQ, 4: + dK/dV up to head dim 64, 5: up to 96): the bf16 fused attention kernels with two 32-row tiles per iteration (one barrier per 64 keys /
    queries; forward: one rescale decision per 64 keys) against the one-tile kernels - same arithmetic per element, other
    summation order / max granularity: equal to bf16 rounding (log-sum-exp to fp32 rounding) - and against a materialised
    fp32 reference.  Ragged tile counts (a masked second tile), the query-split dK/dV path (77 keys) included."""
    dtype = torch.bfloat16
    B, Nq, Nk, H, d = cfg
    q, k_, v = (rnd(B * n, H * d, dtype=dtype, seed=i) for i, n in ((1, Nq), (2, Nk), (3, Nk)))
    g = rnd(B * Nq, H * d, dtype=dtype, seed=4)
    K = ops.kernels()
    outs = []
    for kt in (1, 5):
        _set_opts(flash_kt=kt, flash_merge=0)
        qd, kd, vd, gd = (dv(t, hip, dtype) for t in (q, k_, v, g))
        o = torch.empty_like(qd)
        lse, dbuf = torch.empty(B, H, Nq, device=hip), torch.empty(B, H, Nq, device=hip)
        HD = H * d
        K.flash_attn_fwd(qd, kd, vd, o, lse, B, H, Nq, Nk, d, HD, HD, HD, HD, d ** -0.5)
        dq, dk, dvv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
        K.flash_attn_bwd(qd, kd, vd, o, gd, lse, dbuf, dq, dk, dvv, B, H, Nq, Nk, d, HD, HD, HD, HD, d ** -0.5)
        outs.append((o.float(), lse.clone(), dq.float(), dk.float(), dvv.float()))
    for i, name in ((0, "O"), (2, "dQ"), (3, "dK"), (4, "dV")):
        assert rel_l2(outs[1][i], outs[0][i]) < 6e-3, f"{name}: two-tile kernels differ from the one-tile kernels by {rel_l2(outs[1][i], outs[0][i]):.2e}"
    assert (outs[1][1] - outs[0][1]).abs().max() < 1e-4 * (1 + outs[0][1].abs().max())
    qr = q.reshape(B, Nq, H, d).permute(0, 2, 1, 3).clone().requires_grad_(True)
    kr = k_.reshape(B, Nk, H, d).permute(0, 2, 1, 3).clone().requires_grad_(True)
    vr = v.reshape(B, Nk, H, d).permute(0, 2, 1, 3).clone().requires_grad_(True)
    ref = (torch.softmax(qr @ kr.transpose(-1, -2) * d ** -0.5, -1) @ vr).permute(0, 2, 1, 3).reshape(B * Nq, H * d)
    ref.backward(g)
    back = lambda t, n: t.grad.permute(0, 2, 1, 3).reshape(B * n, H * d)
    check(outs[1][0], ref, dtype, "flash forward, two tiles per iteration")
    check(outs[1][2], back(qr, Nq), dtype, "flash dQ, two tiles per iteration", factor=3)
    check(outs[1][3], back(kr, Nk), dtype, "flash dK, two tiles per iteration", factor=3)
    check(outs[1][4], back(vr, Nk), dtype, "flash dV, two tiles per iteration", factor=3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(64, 96, 64), (300, 320, 1280), (16, 64, 32)])
def test_geglu_linear(dev, dtype, shape):
    """ops.geglu_linear (the feed-forward's first projection with its GEGLU in the GEMM epilogue, weight rows interleaved in
    sixteens) against `value, gate = (x W^T + b).chunk(2, -1); value * gelu(gate)` on the ORIGINAL weight: output, input
    gradient; and the no-grad call (which stores no pre-activations) gives the same output."""
    M, K, D = shape
    w, b = rnd(2 * D, K, dtype=dtype, seed=1, scale=K ** -0.5), rnd(2 * D, seed=2)
    lin = ops.FrozenGegluLinear(w, b, dtype, dev)
    x = rnd(M, K, dtype=dtype, seed=3)
    go = rnd(M, D, dtype=dtype, seed=4)
    xd = dv(x, dev, dtype, grad=True)
    y = ops.geglu_linear(xd, lin)
    y.backward(dv(go, dev, dtype))
    xr = x.clone().requires_grad_(True)
    pre = (xr @ w.t() + b).to(dtype).float() if dtype != torch.float32 else xr @ w.t() + b
    pre = xr @ w.t() + b
    ref = pre[:, :D] * F.gelu(pre[:, D:])
    ref.backward(go)
    check(y, ref, dtype, "geglu_linear", factor=2)
    check(xd.grad, xr.grad, dtype, "geglu_linear dx", factor=3)
    with torch.no_grad():
        y2 = ops.geglu_linear(dv(x, dev, dtype), lin)
    assert torch.equal(y2, y.detach()), "no-grad call differs"
    # the two-launch form (projection, then the interleaved-layout GEGLU kernel) computes the same bits: both round the
    # pre-activations to the storage type before the product
    ops.set_geglu_fused(False)
    try:
        xs = dv(x, dev, dtype, grad=True)
        y3 = ops.geglu_linear(xs, lin)
        y3.backward(dv(go, dev, dtype))
    finally:
        ops.set_geglu_fused(True)
    assert torch.equal(y3, y) and torch.equal(xs.grad, xd.grad), "fused and two-launch GEGLU differ"


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(130, 320, 640), (4096, 640, 320), (77, 24, 8), (50, 7, 9)])
def test_concat_cols_pair_copy(dev, dtype, shape):
    """ops.concat_cols and its backward split: both halves in ONE launch (comat_copy2d_pair) where the rows are whole 16-byte
    vectors, two comat_copy2d launches otherwise (odd column counts) - exact copies either way."""
    M, Ca, Cb = shape
    a, b, g = rnd(M, Ca, dtype=dtype, seed=1), rnd(M, Cb, dtype=dtype, seed=2), rnd(M, Ca + Cb, dtype=dtype, seed=3)
    ad, bd = dv(a, dev, dtype, grad=True), dv(b, dev, dtype, grad=True)
    out = ops.concat_cols(ad, bd)
    out.backward(dv(g, dev, dtype))
    assert torch.equal(out.detach().float().cpu(), torch.cat([a, b], 1))
    assert torch.equal(ad.grad.float().cpu(), g[:, :Ca]) and torch.equal(bd.grad.float().cpu(), g[:, Ca:])
    a2 = dv(a, dev, dtype, grad=True)  # only one side needs a gradient
    ops.concat_cols(a2, dv(b, dev, dtype)).backward(dv(g, dev, dtype))
    assert torch.equal(a2.grad.float().cpu(), g[:, :Ca])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(64, 96, 64, 80), (300, 320, 1280, 320), (16, 64, 32, 64), (2048, 640, 2560, 640)])
def test_geglu_feed_forward(dev, dtype, shape):
    """ops.geglu_feed_forward = GEGLU(x W1^T + b1) W2^T + b2 + residual as one autograd node: output, input and residual
    gradients against torch on the ORIGINAL weights; its backward (the GEGLU's gradient in the epilogue of the second
    projection's data-gradient GEMM, comat_gemm_params::epi2 = 3) gives the bits of the two separate operators."""
    M, K, D, N = shape
    if dev.type != "cuda" and M * D > 300 * 1280:
        pytest.skip("large shape: GPU only")
    w1, b1 = rnd(2 * D, K, dtype=dtype, seed=1, scale=K ** -0.5), rnd(2 * D, seed=2)
    w2, b2 = rnd(N, D, dtype=dtype, seed=5, scale=D ** -0.5), rnd(N, seed=6)
    ff1, ff2 = ops.FrozenGegluLinear(w1, b1, dtype, dev), ops.FrozenLinear(w2, b2, dtype, dev)
    x, res, go = rnd(M, K, dtype=dtype, seed=3), rnd(M, N, dtype=dtype, seed=7), rnd(M, N, dtype=dtype, seed=4)
    xr, rr = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
    pre = xr @ w1.t() + b1
    ref = (pre[:, :D] * F.gelu(pre[:, D:])) @ w2.t() + b2 + rr
    ref.backward(go)
    xd, rd = dv(x, dev, dtype, grad=True), dv(res, dev, dtype, grad=True)
    y = ops.geglu_feed_forward(xd, ff1, ff2, residual=rd)
    y.backward(dv(go, dev, dtype))
    check(y, ref, dtype, "feed-forward", factor=3)
    check(xd.grad, xr.grad, dtype, "feed-forward dx", factor=4)
    check(rd.grad, rr.grad, dtype, "feed-forward dresidual")
    xs, rs = dv(x, dev, dtype, grad=True), dv(res, dev, dtype, grad=True)
    y2 = ops.linear(ops.geglu_linear(xs, ff1), ff2, residual=rs)
    y2.backward(dv(go, dev, dtype))
    assert torch.equal(y2, y), "one node vs two operators: outputs differ"
    if dtype == torch.bfloat16:
        assert torch.equal(xs.grad, xd.grad) and torch.equal(rs.grad, rd.grad), "GEGLU' in the epilogue differs from the two-launch form"
    else:
        check(xd.grad, xs.grad, dtype, "fp32: one node vs two operators")
    with torch.no_grad():
        assert torch.equal(ops.geglu_feed_forward(dv(x, dev, dtype), ff1, ff2, residual=dv(res, dev, dtype)), y.detach())


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(300, 320, 1280), (2048, 640, 2560)])
def test_geglu_linear_without_the_pipelined_kernel(hip, shape, default_opts):
    """with option gemm2 = 0 (every contraction on the general kernel) the GEGLU epilogue of comat_gemm is not
    available - the library then runs the product and the interleaved-layout GEGLU kernel as two launches (the no-grad form,
    which has no buffer for the pre-activations, through its workspace) instead of refusing: same bits as the fused call."""
    dtype = torch.bfloat16
    M, K, D = shape
    w, b = rnd(2 * D, K, dtype=dtype, seed=1, scale=K ** -0.5), rnd(2 * D, seed=2)
    lin = ops.FrozenGegluLinear(w, b, dtype, hip)
    x, go = rnd(M, K, dtype=dtype, seed=3), rnd(M, D, dtype=dtype, seed=4)
    res = []
    w2 = ops.FrozenLinear(rnd(K, D, dtype=dtype, seed=8, scale=D ** -0.5), None, dtype, hip)
    for g2 in (1, 0):
        _set_opts(gemm2=g2, gemm3=0)
        xd = dv(x, hip, dtype, grad=True)
        y = ops.geglu_linear(xd, lin)
        y.backward(dv(go, hip, dtype))
        with torch.no_grad():
            y2 = ops.geglu_linear(dv(x, hip, dtype), lin)
        xf = dv(x, hip, dtype, grad=True)  # the backward epilogue (epi2 = 3) and its two-launch form
        ops.geglu_feed_forward(xf, lin, w2).backward(dv(rnd(M, K, dtype=dtype, seed=9), hip, dtype))
        res.append((y.detach(), xd.grad, y2, xf.grad))
    # the general kernel accumulates in another order: equal up to bf16 rounding of the pre-activations
    for a, b_, name in zip(res[0], res[1], ("output", "input gradient", "no-grad output", "feed-forward input gradient")):
        assert rel_l2(b_, a) < 1e-2, f"{name}: {rel_l2(b_, a):.2e}"
    assert torch.equal(res[1][0], res[1][2]), "gemm2 = 0: no-grad call differs from the grad-mode call"


G3_CFGS = [1, 2, 3, 4, 5, 6, 7, 8, 9]


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", G3_CFGS)
def test_gemm3_lean_kernel(hip, cfg, default_opts):
    """gemm3.hip (option gemm3 = 2: every eligible problem), every tile shape: plain / batched / K-segmented GEMMs with the
    fused epilogue (bias, residual, activation, fp32 output), ragged M and N, a contraction shorter than the wave count -
    against the fp32 reference of the bf16-rounded operands, and against the pipelined kernel (same operands: the two differ
    only in summation order)."""
    dtype = torch.bfloat16
    k = ops.kernels()
    from comat_amd import _hip
    for (M, N, K_, nb, extra) in ((512, 128, 1280, 1, ""), (2048, 128, 640, 1, ""), (300, 200, 320, 1, "bias res"), (77, 1280, 768, 1, ""),
                                  (512, 1280, 1280, 1, "res"), (130, 96, 64, 2, ""), (16, 768, 768, 1, "bias gelu"),
                                  (577, 1024, 1024, 1, "f32"), (2048, 384, 640, 3, "")):
        A = dv(rnd(nb, M, K_, dtype=dtype, seed=1, scale=0.5), hip, dtype)
        B = dv(rnd(nb, N, K_, dtype=dtype, seed=2, scale=0.5), hip, dtype)
        bias = dv(rnd(N, seed=3), hip, torch.float32) if "bias" in extra else None
        R = dv(rnd(M, N, dtype=dtype, seed=4), hip, dtype) if "res" in extra and nb == 1 else None
        odt = torch.float32 if "f32" in extra else dtype
        act = ops.ACT_GELU if "gelu" in extra else ops.ACT_NONE
        outs = []
        for g3 in (0, 2):
            _set_opts(gemm3=g3, g3_cfg=cfg)
            out = torch.full((nb, M, N), float("nan"), dtype=odt, device=hip)
            k.gemm(A, B, out, M, N, K_, K_, K_, N, batch=(nb, 1), sA=(M * K_, 0), sB=(N * K_, 0), sC=(M * N, 0), bias=bias, R=R,
                   ldr=N, beta=1.0 if R is not None else 0.0, act=act)
            assert _hip.last_gemm_kernel() == (5 if g3 else 1), (M, N, K_, _hip.last_gemm_kernel())
            outs.append(out)
        ref = torch.einsum("bmk,bnk->bmn", A.float(), B.float())
        if bias is not None:
            ref = ref + bias
        if act == ops.ACT_GELU:
            ref = F.gelu(ref)
        if R is not None:
            ref = ref + R.float()
        check(outs[1], ref, dtype, f"lean kernel cfg {cfg}: gemm {M}x{N}x{K_} b={nb} {extra}")
        check(outs[1], outs[0].float(), dtype, f"lean kernel cfg {cfg} vs pipelined: gemm {M}x{N}x{K_} b={nb} {extra}")
    # K-segmented, batched (the q / k / v projections of one activation with their low-rank parts)
    M, N, K_, r, G = 512, 1280, 1280, 128, 3
    x = dv(rnd(M, K_, dtype=dtype, seed=5, scale=0.3), hip, dtype)
    W = dv(rnd(G, N, K_, dtype=dtype, seed=6, scale=0.3), hip, dtype)
    H_ = dv(rnd(M, G * r, dtype=dtype, seed=7, scale=0.3), hip, dtype)
    U = dv(rnd(G, N, r, dtype=dtype, seed=8, scale=0.3), hip, dtype)
    outs = []
    for g3 in (0, 2):
        _set_opts(gemm3=g3, g3_cfg=cfg)
        out = torch.full((G, M, N), float("nan"), dtype=dtype, device=hip)
        k.gemm_segments([(x, W[0], K_, K_, K_, 0, N * K_), (H_, U[0], r, G * r, r, r, N * r)], out, M, N, N, batch=G, sC=M * N)
        assert _hip.last_gemm_kernel() == (5 if g3 else 1)
        outs.append(out)
    ref = torch.einsum("mk,gnk->gmn", x.float(), W.float()) + torch.einsum("mgr,gnr->gmn", H_.float().reshape(M, G, r), U.float())
    check(outs[1], ref, dtype, f"lean kernel cfg {cfg}: batched K-segmented GEMM")
    check(outs[1], outs[0].float(), dtype, f"lean kernel cfg {cfg} vs pipelined: batched K-segmented GEMM")
    # bit-reproducible run to run
    _set_opts(gemm3=2, g3_cfg=cfg)
    out2 = torch.empty_like(outs[1])
    k.gemm_segments([(x, W[0], K_, K_, K_, 0, N * K_), (H_, U[0], r, G * r, r, r, N * r)], out2, M, N, N, batch=G, sC=M * N)
    assert torch.equal(out2, outs[1])


@pytest.mark.gpu
@pytest.mark.parametrize("splits", [0, 3])
def test_gemm2_tile_order_is_bit_identical(hip, splits, default_opts):
    """option g2_order: the output tiles of the pipelined kernel walked row-block-major (0), column-block-major (1) or as the
    L2-miss model of the launcher prefers per problem (2) inside each XCD's chunk - which block computes which tile changes,
    the arithmetic of a tile (and the slice order of its split-K combine) does not: same bits, for plain / batched /
    K-segmented GEMMs, convs, ragged edges and forced split counts."""
    dtype = torch.bfloat16
    k = ops.kernels()
    outs = {}
    for order in (0, 1, 2):
        _set_opts(gemm2=1, g2_order=order, g2_splits=splits)
        res = []
        for (M, N, K_, nb) in ((512, 1280, 1280, 1), (300, 200, 320, 1), (8192, 320, 320, 1), (577, 1024, 4096, 1),
                               (2048, 384, 640, 3), (512, 10240, 1280, 1), (130, 96, 64, 2)):
            A = dv(rnd(nb, M, K_, dtype=dtype, seed=1, scale=0.5), hip, dtype)
            B = dv(rnd(nb, N, K_, dtype=dtype, seed=2, scale=0.5), hip, dtype)
            out = torch.full((nb, M, N), float("nan"), dtype=dtype, device=hip)
            k.gemm(A, B, out, M, N, K_, K_, K_, N, batch=(nb, 1), sA=(M * K_, 0), sB=(N * K_, 0), sC=(M * N, 0))
            res.append(out)
        x = dv(rnd(2 * 16 * 16, 1280, dtype=dtype, seed=3, scale=0.3), hip, dtype)
        conv = ops.FrozenConv(rnd(1280, 1280, 3, 3, seed=4, scale=0.02), None, dtype, hip)
        res.append(ops.conv2d(x, conv, 2, 16, 16))
        A1, H_ = dv(rnd(512, 1280, dtype=dtype, seed=5, scale=0.3), hip, dtype), dv(rnd(512, 128, dtype=dtype, seed=6, scale=0.3), hip, dtype)
        W, U = dv(rnd(1280, 1280, dtype=dtype, seed=7, scale=0.3), hip, dtype), dv(rnd(1280, 128, dtype=dtype, seed=8, scale=0.3), hip, dtype)
        out = torch.full((512, 1280), float("nan"), dtype=dtype, device=hip)
        k.gemm_segments([(A1, W, 1280, 1280, 1280), (H_, U, 128, 128, 128)], out, 512, 1280, 1280)
        res.append(out)
        outs[order] = res
    for order in (1, 2):
        for i, (a, b) in enumerate(zip(outs[0], outs[order])):
            assert torch.isfinite(a.float()).all() and torch.equal(a, b), f"g2_order={order}: result {i} differs from row-block-major order"


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(2, 4096, 4096, 8, 40), (2, 1024, 1024, 8, 80), (1, 577, 577, 16, 64), (2, 1024, 77, 8, 80),
                                 (1, 300, 200, 3, 40), (2, 256, 256, 8, 160)])
def test_flash_xcd_renumbering_is_bit_identical(hip, cfg, default_opts):
    """option flash_xcd: the fused attention's workgroups renumbered so that one (batch, head) runs on one XCD - forward,
    dQ, dK/dV (separate launches; the one-launch backward ignores the option) give the same bits."""
    dtype = torch.bfloat16
    B, Nq, Nk, H, d = cfg
    q, k_, v = (rnd(B * n, H * d, dtype=dtype, seed=i) for i, n in ((1, Nq), (2, Nk), (3, Nk)))
    g = rnd(B * Nq, H * d, dtype=dtype, seed=4)
    K = ops.kernels()
    HD = H * d
    got = []
    for xcd in (0, 1):
        _set_opts(flash_xcd=xcd, flash_merge=0)
        qd, kd, vd, gd = (dv(t, hip, dtype) for t in (q, k_, v, g))
        o = torch.full_like(qd, float("nan"))
        lse, dbuf = torch.empty(B, H, Nq, device=hip), torch.empty(B, H, Nq, device=hip)
        K.flash_attn_fwd(qd, kd, vd, o, lse, B, H, Nq, Nk, d, HD, HD, HD, HD, d ** -0.5)
        dq, dk, dvv = torch.full_like(qd, float("nan")), torch.full_like(kd, float("nan")), torch.full_like(vd, float("nan"))
        K.flash_attn_bwd(qd, kd, vd, o, gd, lse, dbuf, dq, dk, dvv, B, H, Nq, Nk, d, HD, HD, HD, HD, d ** -0.5)
        got.append((o, lse, dq, dk, dvv))
    for name, a, b in zip(("O", "lse", "dQ", "dK", "dV"), *got):
        assert torch.isfinite(a.float()).all() and torch.equal(a, b), f"{name}: flash_xcd = 1 differs"


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [(2, 256, 256, 8, 160), (2, 1024, 1024, 8, 80), (1, 577, 577, 16, 64), (2, 256, 77, 8, 160),
                                 (2, 1024, 77, 8, 80), (1, 300, 200, 3, 40), (1, 16, 577, 12, 64), (2, 70, 130, 2, 32),
                                 (1, 2048, 2048, 2, 40)])
def test_flash_backward_in_one_launch_is_bit_identical(hip, cfg, dtype, default_opts):
    """option flash_merge: D = rowsum(dO . O) from its own small kernel, then the dQ blocks and the dK/dV blocks of the
    attention in ONE launch (side by side on the chip where neither grid fills it) - the same device bodies as the separate
    kernels, so dQ / dK / dV / D are the same bits; one- and two-tile dQ (flash_kt 1 / 3), the query-split dK/dV path
    (77 keys), ragged tiles, fp32 parity mode."""
    B, Nq, Nk, H, d = cfg
    q, k_, v = (rnd(B * n, H * d, dtype=dtype, seed=i) for i, n in ((1, Nq), (2, Nk), (3, Nk)))
    g = rnd(B * Nq, H * d, dtype=dtype, seed=4)
    K = ops.kernels()
    HD = H * d
    qd, kd, vd, gd = (dv(t, hip, dtype) for t in (q, k_, v, g))
    o = torch.empty_like(qd)
    lse = torch.empty(B, H, Nq, device=hip)
    K.flash_attn_fwd(qd, kd, vd, o, lse, B, H, Nq, Nk, d, HD, HD, HD, HD, d ** -0.5)
    for kt in (1, 3):  # (4 / 5 pick the two-tile dK/dV body for the separate launch only: another summation order)
        got = []
        for merge in (0, 2, 1):
            _set_opts(flash_kt=kt, flash_merge=merge)
            dbuf = torch.full((B, H, Nq), float("nan"), device=hip)
            dq, dk, dvv = torch.full_like(qd, float("nan")), torch.full_like(kd, float("nan")), torch.full_like(vd, float("nan"))
            K.flash_attn_bwd(qd, kd, vd, o, gd, lse, dbuf, dq, dk, dvv, B, H, Nq, Nk, d, HD, HD, HD, HD, d ** -0.5)
            got.append((dq, dk, dvv, dbuf))
        for other in got[1:]:
            for name, a, b in zip(("dQ", "dK", "dV", "D"), got[0], other):
                assert torch.isfinite(a.float()).all(), f"{name}: not written"
                assert torch.equal(a, b), f"{name}: one-launch backward differs from the separate kernels (flash_kt={kt})"


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("G", [1, 3])
def test_merged_nograd_weights(dev, dtype, G, monkeypatch):
    """COMAT_NOGRAD_MERGED=1: no-grad calls through W + s U D (merged once per optimizer step) give the unmerged result
    (fp32: to rounding; bf16: within bf16 tolerance), follow parameter updates, and leave grad-mode calls untouched."""
    M, K, r, N = 130, 64, 8, 96
    x = rnd(M, K, dtype=dtype, seed=1)
    ws = [rnd(N, K, dtype=dtype, seed=20 + i, scale=K ** -0.5) for i in range(G)]
    spec = [(f"p{i}.down", f"p{i}.up", rnd(r, K, seed=40 + i, scale=r ** -0.5), rnd(N, r, seed=50 + i, scale=0.2))
            for i in range(G)]
    res = rnd(M, N, dtype=dtype, seed=6) if G == 1 else None
    lins = (ops.frozen_linear_group(ws, [None] * G, dtype, dev) if G > 1
            else [ops.FrozenLinear(ws[0], rnd(N, seed=30), dtype, dev)])
    store = ops.LoRAStore([spec], dtype, dev)
    xd = dv(x, dev, dtype)
    rd = dv(res, dev, dtype) if res is not None else None
    with torch.no_grad():
        plain = ops.lora_group_linear(xd, lins, store.groups[0], residual=rd)
        monkeypatch.setenv("COMAT_NOGRAD_MERGED", "1")
        merged = ops.lora_group_linear(xd, lins, store.groups[0], residual=rd)
    f = 1.0 if dtype == torch.float32 else 1.5
    for a, b in zip(merged, plain):
        check(a, b, dtype, "merged vs unmerged", factor=f)
    # parameter update -> merged weights refreshed in place (same buffers)
    ptrs = [w.data_ptr() for ws_ in store.merged_weights(store.groups[0], tuple(lins)) for w in ws_]
    store.flat.mul_(0.5)
    store.mark_updated()
    with torch.no_grad():
        merged2 = ops.lora_group_linear(xd, lins, store.groups[0], residual=rd)
        monkeypatch.setenv("COMAT_NOGRAD_MERGED", "0")
        plain2 = ops.lora_group_linear(xd, lins, store.groups[0], residual=rd)
    wm, wmt = store.merged_weights(store.groups[0], tuple(lins))
    assert ptrs == [w.data_ptr() for w in wm + wmt]
    for a, b in zip(wm, wmt):  # the data-gradient's copy is the transpose of the forward's (fp32: to rounding)
        if dtype == torch.bfloat16 and dev.type == "cuda":
            check(b, a.t(), dtype, "transposed merged weight")
        else:
            check(b, a.t(), dtype, "transposed merged weight")
    for a, b, c in zip(merged2, plain2, plain):
        check(a, b, dtype, "merged vs unmerged after update", factor=f)
        assert (a.float() - c.float()).abs().max() > 0
    # grad-mode calls record a backward node whatever the no-grad switch says
    monkeypatch.setenv("COMAT_NOGRAD_MERGED", "1")
    xg = dv(x, dev, dtype, grad=True)
    y = ops.lora_group_linear(xg, lins, store.groups[0], residual=rd)
    assert y[0].grad_fn is not None


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 320, 320, 128), (2, 640, 768, 128), (1, 1280, 1280, 128), (1, 200, 136, 16), (3, 64, 64, 32)])
def test_lora_merge_grouped_kernel(hip, shape):
    """comat_lora_merge: Wm = bf16(W + s U D) and WmT = Wm^T for every projection of a store in ONE launch - against the fp32
    formula (one rounding), the transposed copy bit-identical to the forward copy, refreshed in place after a parameter
    update, ragged tile edges included; and the trained call built on it against the low-rank form."""
    G, N, K, r = shape
    dtype = torch.bfloat16
    ws = [rnd(N, K, dtype=dtype, seed=20 + i, scale=K ** -0.5) for i in range(G)]
    spec = [(f"p{i}.down", f"p{i}.up", rnd(r, K, seed=40 + i, scale=K ** -0.5), rnd(N, r, seed=50 + i, scale=0.05)) for i in range(G)]
    lins = ops.frozen_linear_group(ws, [None] * G, dtype, hip) if G > 1 else [ops.FrozenLinear(ws[0], None, dtype, hip)]
    store = ops.LoRAStore([spec], dtype, hip, scale=0.75)
    grp = store.groups[0]
    assert ops.kernels().lora_merge_ok(lins[0].w, store.group_views[0][1][0], store.group_views[0][2][:, :r], r, r, G * r)
    for rep in range(2):
        wm, wmt = store.merged_weights(grp, tuple(lins))
        assert store._merge_table is not None and not store._merge_rest, "the grouped kernel did not take the store"
        for i in range(G):
            d, u = store.group_views[0][0][i * r:(i + 1) * r].float(), store.group_views[0][1][i].float()
            ref = (lins[i].w.float() + 0.75 * (u @ d)).to(dtype)
            # fp32 accumulation order of the MFMA differs from torch's: allow the last bf16 bit on a few elements
            diff = (wm[i].float() - ref.float()).abs()
            assert diff.max() <= 2.0 ** -7 * ref.float().abs().max(), f"merged weight {i}: {diff.max():.3e}"
            assert (diff > 0).float().mean() < 2e-2, f"merged weight {i}: {(diff > 0).float().mean():.3e} of the elements differ"
            assert torch.equal(wmt[i], wm[i].t()), f"transposed merged weight {i} is not the forward copy's transpose"
        store.flat.mul_(0.5)
        store.mark_updated()
    # the same entry through comat_gemm (the fp32-mode path) agrees to bf16 rounding
    ent = next(iter(store._merged.values()))
    a = [w.clone() for w in ent["wm"]]
    store._merge_into(ent)
    for x, y in zip(a, ent["wm"]):
        assert rel_l2(x, y) < 2e-3


@pytest.mark.gpu
def test_flash_backward_partials_do_not_touch_the_gemm_tickets(hip, default_opts):
    """Regression (round 2): the fused-attention backward with few keys (cross-attention, 77 text tokens) parks fp32
    dK / dV partials in the stream's workspace; they must live behind the split-K ticket counters at its head, or every
    later split-K GEMM on that stream combines garbage.  Run such a backward, then split-K GEMMs of both kernels."""
    dtype = torch.bfloat16
    B, Nq, Nk, H, d = 2, 4096, 77, 8, 40
    q, k_, v = (rnd(B * n, H * d, dtype=dtype, seed=i) for i, n in ((1, Nq), (2, Nk), (3, Nk)))
    qd, kd, vd = (dv(t, hip, dtype, grad=True) for t in (q, k_, v))
    o, _ = ops.attention(qd, kd, vd, B, Nq, Nk, H, d, need_probs=False)
    o.backward(dv(rnd(B * Nq, H * d, dtype=dtype, seed=4), hip, dtype))
    k = ops.kernels()
    A, Bm = rnd(512, 4096, dtype=dtype, seed=5, scale=0.3), rnd(256, 4096, dtype=dtype, seed=6, scale=0.3)
    for g2, opt in ((1, "g2_splits"), (0, "force_splits")):
        _set_opts(gemm2=g2, **{opt: 4})
        out = torch.full((512, 256), float("nan"), dtype=torch.float32, device=hip)
        k.gemm(dv(A, hip, dtype), dv(Bm, hip, dtype), out, 512, 256, 4096, 4096, 4096, 256)
        check(out, A @ Bm.t(), dtype, f"split-K GEMM after a split flash backward (gemm2={g2})")


@pytest.mark.gpu
def test_flash_refuses_a_batch_entry_beyond_the_descriptor_range(hip, default_opts):
    """Round 4: full attention tiles come through buffer loads whose offsets inside one (batch, head) slab are 32 bits wide
    (TileMover::load_full).  A batch entry of 2 GiB or more must be refused by the entry point, not wrapped around; the
    same shapes with an ordinary leading dimension go through, mixing full tiles (fast path) and a ragged last tile."""
    k = ops.kernels()
    dtype = torch.bfloat16
    N, H, d = 200, 1, 64  # 3 full 64-key pairs + a ragged one
    q = dv(rnd(N, H * d, dtype=dtype, seed=1), hip, dtype)
    o, lse = torch.empty_like(q), torch.empty(1, H, N, device=hip)
    with pytest.raises(RuntimeError, match="2 GiB"):
        k.flash_attn_fwd(q, q, q, o, lse, 1, H, N, N, d, 1 << 24, H * d, H * d, H * d, d ** -0.5)  # 200 rows x 2^24 x 2 B
    k.flash_attn_fwd(q, q, q, o, lse, 1, H, N, N, d, H * d, H * d, H * d, H * d, d ** -0.5)
    qf = q.float()
    ref = torch.softmax(qf @ qf.t() * d ** -0.5, dim=-1) @ qf
    check(o, ref, dtype, "flash forward, 200 keys (full pairs + ragged tail)")


@pytest.mark.gpu
@pytest.mark.parametrize("splits", [0, 1, 5])
def test_gemm2_k_major_operands(hip, splits, default_opts):
    """C (+)= A^T B with both operands stored k-major (the LoRA weight gradients dU = g^T h, dD = u^T x: contraction over
    the token axis) on the pipelined kernel: DMA of the [k][column] tiles + hardware transpose reads.  Ragged column counts
    (multiples of 8), in-place fp32 accumulation, a batched launch, against an fp32 reference and the general kernel."""
    dtype = torch.bfloat16
    k = ops.kernels()
    for (M, N, K_, nb) in ((320, 128, 8192, 1), (384, 320, 8192, 1), (1280, 128, 512, 3), (128, 640, 2048, 1), (200, 136, 1024, 1),
                           (8, 8, 256, 1)):
        A = rnd(nb, K_, M, dtype=dtype, seed=1, scale=0.3)
        B = rnd(nb, K_, N, dtype=dtype, seed=2, scale=0.3)
        C0 = rnd(nb, M, N, seed=3)
        ref = torch.einsum("bkm,bkn->bmn", A, B) + C0
        outs = []
        for tt in (1, 0):
            _set_opts(gemm2=1, gemm2_tt=tt, g2_splits=splits, force_splits=splits)
            out = dv(C0, hip)
            k.gemm(dv(A, hip, dtype), dv(B, hip, dtype), out, M, N, K_, M, N, N, transA=True, transB=True, R=out, ldr=N,
                   beta=1.0, batch=(nb, 1), sA=(K_ * M, 0), sB=(K_ * N, 0), sC=(M * N, 0), sR=(M * N, 0))
            outs.append(out)
        check(outs[0], ref, dtype, f"k-major gemm2 M={M} N={N} K={K_} b={nb} splits={splits}")
        check(outs[0], outs[1], dtype, f"k-major gemm2 vs general kernel M={M} N={N}", factor=0.02)
