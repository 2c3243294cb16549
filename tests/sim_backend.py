"""CPU simulator of the libcomat_hip C ABI — TEST INFRASTRUCTURE ONLY.

It implements, with plain torch CPU ops, the documented semantics of every entry point in include/comat_hip.h
(same argument lists as comat_amd._hip.HipKernels).  tests/ plug it in through
`comat_amd.ops.set_kernel_backend(SimKernels())` to exercise the host logic (operator wiring, model assembly,
gradient gating, the step graph, the 2-rank gloo path) on machines without a GPU.  The product never imports it.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

ACT_NONE, ACT_SILU, ACT_GELU = 0, 1, 2
UN_COPY, UN_SILU, UN_GELU, UN_AFFINE = 0, 1, 2, 3


def _v(t, sizes, strides):
    return torch.as_strided(t, sizes, strides, t.storage_offset())


def _f(t, scale=None):
    """operand -> fp32 values; uint8 tensors hold fp8 e4m3 bytes (value = scale * fp8)"""
    if t.dtype == torch.uint8:
        return t.view(torch.float8_e4m3fn).float() * scale.float()
    return t.float()


def _act(x, act):
    if act == ACT_SILU:
        return F.silu(x)
    if act == ACT_GELU:
        return F.gelu(x)
    return x


class SimKernels:
    name = "sim"

    # ---- contraction ---------------------------------------------------------------------------------------
    def gemm(self, A, B, Cout, M, N, K, lda, ldb, ldc, transA=False, transB=False, batch=(1, 1), sA=(0, 0),
             sB=(0, 0), sC=(0, 0), bias=None, bias2=None, rows_per_bias2=0, R=None, ldr=0, sR=(0, 0), alpha=1.0,
             beta=0.0, act=ACT_NONE, scales=None, geglu=None, tail=None, ktail=None, q8=None):
        b1, b2 = batch
        if tail is not None:  # comat_gemm_params::epi2 = 4: the last n2 columns are alpha2 * A B2^T into C2, the rest as usual
            B2, C2t, n2, ldc2, sB2t, sC2t, alpha2 = tail
            assert geglu is None and not transA and not transB and b2 == 1 and 0 < n2 < N
            self.gemm(A, B, Cout, M, N - n2, K, lda, ldb, ldc, batch=batch, sA=sA, sB=sB, sC=sC, bias=bias, bias2=bias2,
                      rows_per_bias2=rows_per_bias2, R=R, ldr=ldr, sR=sR, alpha=alpha, beta=beta, act=act, scales=scales)
            self.gemm(A, B2, C2t, M, n2, K, lda, ldb, ldc2, batch=batch, sA=sA, sB=(sB2t, 0), sC=(sC2t, 0), alpha=alpha2,
                      scales=scales)
            return
        assert (scales is not None) == (A.dtype == torch.uint8)
        if scales is not None:
            assert not transA and not transB and K % 64 == 0 and lda % 16 == 0 and ldb % 16 == 0 and b2 == 1
        sa, sb = scales[:2] if scales is not None else (None, None)
        if scales is not None and len(scales) > 2 and scales[2] and b1 > 1:  # per-batch scale of B (comat_gemm_params::s_scale_b)
            sb = torch.as_strided(sb, (b1, 1, 1, 1), (scales[2], 0, 0, 0), sb.storage_offset())
        Av = _v(A, (b1, b2, M, K), (sA[0], sA[1], 1, lda) if transA else (sA[0], sA[1], lda, 1))
        Bv = _v(B, (b1, b2, N, K), (sB[0], sB[1], 1, ldb) if transB else (sB[0], sB[1], ldb, 1))
        prod = _f(Av, sa) @ _f(Bv, sb).transpose(-1, -2)
        if ktail is not None:  # comat_gemm_params::A2k: bf16 k-tail on top of the scaled fp8 product
            A2, B2, K2, lda2, ldb2, sA2, sB2 = ktail
            assert scales is not None and K2 % 16 == 0 and geglu is None and b2 == 1
            prod = prod + _v(A2, (b1, b2, M, K2), (sA2, 0, lda2, 1)).float() @ _v(B2, (b1, b2, N, K2), (sB2, 0, ldb2, 1)).float().transpose(-1, -2)
        acc = alpha * prod
        if bias is not None:
            acc = acc + bias.float()
        if bias2 is not None:
            acc = acc + bias2.float().repeat_interleave(rows_per_bias2, 0)[:M]
        acc = _act(acc, act)
        if R is not None:
            acc = acc + beta * _v(R, (b1, b2, M, N), (sR[0], sR[1], ldr, 1)).float()
        if geglu is not None and geglu[1] == "bwd":  # epi2 = 3: dF (rounded) -> gradient of the interleaved pre-activations
            pre = geglu[0]
            assert b1 == b2 == 1 and N % 16 == 0 and R is None and act == ACT_NONE and bias is None and bias2 is None
            dF = acc.reshape(M, N).to(pre.dtype)
            self.geglu_il_bwd(dF, pre, Cout, M, N)
            return
        if geglu is not None:  # comat_gemm_params::epi2: value / gate columns interleaved in sixteens, both rounded first
            y, keep = geglu
            assert b1 == b2 == 1 and N % 32 == 0 and R is None and act == ACT_NONE and bias2 is None
            assert y is not None or q8 is not None
            pre = acc.reshape(M, N).to(torch.bfloat16 if y is None else y.dtype)
            if keep:
                _v(Cout, (M, N), (ldc, 1)).copy_(pre)
            t = pre.float().reshape(M, N // 32, 2, 16)
            out = (t[:, :, 0] * F.gelu(t[:, :, 1])).reshape(M, N // 2).to(pre.dtype)
            if y is not None:
                y.copy_(out)
            if q8 is not None:  # comat_gemm_params::q8: the e4m3 bytes of the rounded product + its abs-max
                self._quantize_scaled(out, q8[1], q8[2], out=q8[0])
            return
        _v(Cout, (b1, b2, M, N), (sC[0], sC[1], ldc, 1)).copy_(acc.to(Cout.dtype))

    @staticmethod
    def geglu_gemm_ok(x, w, M, N, K):
        return N % 32 == 0 and M >= 1

    def geglu_il_fwd(self, x, y, M, D):
        t = x.float().reshape(M, D // 16, 2, 16)
        y.copy_((t[:, :, 0] * F.gelu(t[:, :, 1])).reshape(M, D).to(y.dtype))

    def geglu_il_bwd(self, dy, x, dx, M, D):
        t = x.float().detach().reshape(M, D // 16, 2, 16).requires_grad_(True)
        with torch.enable_grad():
            out = (t[:, :, 0] * F.gelu(t[:, :, 1])).reshape(M, D)
            (g,) = torch.autograd.grad(out, t, dy.float())
        dx.copy_(g.reshape(M, 2 * D).to(dx.dtype))

    def gemm_segments(self, segs, Cout, M, N, ldc, bias=None, R=None, ldr=0, alpha=1.0, beta=0.0, batch=1, sC=0, sR=0):
        acc = torch.zeros((batch, M, N), dtype=torch.float32)
        for sg in segs:
            A, B, K, lda, ldb = sg[:5]
            sA, sB = (sg[5], sg[6]) if len(sg) > 5 else (0, 0)
            acc = acc + _v(A, (batch, M, K), (sA, lda, 1)).float() @ _v(B, (batch, N, K), (sB, ldb, 1)).float().transpose(1, 2)
        acc = alpha * acc
        if bias is not None:
            acc = acc + bias.float().reshape(batch, 1, N)
        if R is not None:
            acc = acc + beta * _v(R, (batch, M, N), (sR, ldr, 1)).float()
        _v(Cout, (batch, M, N), (sC, ldc, 1)).copy_(acc.to(Cout.dtype))

    @staticmethod
    def tt_group_ok(A, B, Cacc, M, N, K, lda, ldb, ldc):
        return (A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16 and Cacc.dtype == torch.float32
                and M >= 8 and N >= 8 and K >= 1 and M % 8 == 0 and N % 8 == 0 and lda % 8 == 0 and ldb % 8 == 0
                and ldc % 4 == 0)

    def gemm_tt_grouped(self, problems):
        ptrs = [p[2].data_ptr() for p in problems]
        assert len(set(ptrs)) == len(ptrs), "comat_gemm_tt_grouped: the outputs of one call must not overlap"
        for A, B, Cacc, M, N, K, lda, ldb, ldc in problems:
            assert self.tt_group_ok(A, B, Cacc, M, N, K, lda, ldb, ldc)
            Cv = _v(Cacc, (M, N), (ldc, 1))
            Cv.add_(_v(A, (M, K), (1, lda)).float() @ _v(B, (K, N), (ldb, 1)).float())

    def transpose_cast_tiles(self, src, dst, tiles):
        sf, df = src.reshape(-1), dst.reshape(-1)
        for so, do, rows, cols, r0, c0 in tiles.tolist():
            r1, c1 = min(r0 + 32, rows), min(c0 + 32, cols)
            blk = sf[so:so + rows * cols].view(rows, cols)[r0:r1, c0:c1]
            df[do:do + rows * cols].view(cols, rows)[c0:c1, r0:r1] = blk.t().to(dst.dtype)

    def conv2d(self, X, W, Y, B, Hin, Win, Cin, Hout, Wout, Cout, KH, KW, stride, pad, mode=0, ups=1, bias=None,
               bias2=None, R=None, alpha=1.0, beta=0.0, act=ACT_NONE, scales=None):
        assert (scales is not None) == (X.dtype == torch.uint8)
        if scales is not None:
            assert mode == 0 and Cin % 64 == 0
        sa, sb = scales if scales is not None else (None, None)
        x = _f(_v(X, (B, Hin, Win, Cin), (Hin * Win * Cin, Win * Cin, Cin, 1)), sa).permute(0, 3, 1, 2)
        w = _f(_v(W, (Cout, KH, KW, Cin), (KH * KW * Cin, KW * Cin, Cin, 1)), sb).permute(0, 3, 1, 2)
        if mode == 0:
            if ups == 2:
                x = F.interpolate(x, scale_factor=2, mode="nearest")
            y = F.conv2d(x, w, stride=stride, padding=pad)
        else:
            # transposed gather: src = (dst + k - pad)/stride when divisible
            xz = torch.zeros((B, Cin, (Hin - 1) * stride + 1, (Win - 1) * stride + 1))
            xz[:, :, ::stride, ::stride] = x
            pb = Hout + KH - 1 - pad - xz.shape[2]
            pr = Wout + KW - 1 - pad - xz.shape[3]
            xz = F.pad(xz, (pad, max(pr, 0), pad, max(pb, 0)))
            y = F.conv2d(xz, w)[:, :, :Hout, :Wout]
        assert y.shape == (B, Cout, Hout, Wout), (y.shape, (B, Cout, Hout, Wout))
        acc = alpha * y.permute(0, 2, 3, 1).reshape(B * Hout * Wout, Cout)
        if bias is not None:
            acc = acc + bias.float()
        if bias2 is not None:
            acc = acc + bias2.float().repeat_interleave(Hout * Wout, 0)
        acc = _act(acc, act)
        if R is not None:
            acc = acc + beta * _v(R, (B * Hout * Wout, Cout), (Cout, 1)).float()
        _v(Y, (B * Hout * Wout, Cout), (Cout, 1)).copy_(acc.to(Y.dtype))

    # ---- fp8 operands ---------------------------------------------------------------------------------------
    @staticmethod
    def _track(amax, m):
        """amax [1] int32 holds float bits; non-negative floats order like their bits"""
        cur = amax.view(torch.float32)
        cur.copy_(torch.maximum(cur, m.reshape(1).float()))

    def fp8_quantize_scaled(self, x, scale, amax, out=None):
        return self._quantize_scaled(x, scale, amax, out)

    def _quantize_scaled(self, x, scale, amax, out=None):
        xf = x.float()
        q = (xf * (1.0 / scale.float())).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)
        self._track(amax, xf.abs().max())
        if out is None:
            out = torch.empty(x.shape, dtype=torch.uint8)
        out.copy_(q)
        return out

    def fp8_scales_update(self, amax, scale, n):
        a = amax[:n].view(torch.float32)
        seen = amax[:n] != 0
        scale[:n].copy_(torch.where(seen, torch.clamp(a, min=2.0 ** -100) / 448.0, scale[:n]))
        amax[:n].zero_()

    def layernorm_fwd_q_ok(self, x):
        return x.shape[1] % 8 == 0

    def layernorm_fwd_q(self, x, gamma, beta, y, stats, M, Cc, eps, q8, scale, amax):
        self.layernorm_fwd(x, gamma, beta, y, stats, M, Cc, eps)
        self._quantize_scaled(y, scale, amax, out=q8)

    def groupnorm_fwd_q_ok(self, x, B, HW, Cc, G):
        return Cc % 8 == 0 and HW > 256

    def groupnorm_fwd_q(self, x, gamma, beta, y, stats, B, HW, Cc, G, eps, silu, q8, scale, amax):
        self.groupnorm_fwd(x, gamma, beta, y, stats, B, HW, Cc, G, eps, silu)
        self._quantize_scaled(y, scale, amax, out=q8)

    def fp8_quantize(self, x, out=None, scale=None, amax=None):
        """include/comat_hip.h: scale = max(amax, 2^-100) / 448, bytes = e4m3fn(x * (1 / scale)) (RNE, saturating)"""
        xf = x.float()
        if amax is not None:
            self._track(amax, xf.abs().max())
        sc = torch.clamp(xf.abs().max(), min=2.0 ** -100) / 448.0
        q = (xf * (1.0 / sc)).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)
        if scale is None:
            scale = torch.empty(1, dtype=torch.float32)
        scale.copy_(sc.reshape(1))
        if out is None:
            out = torch.empty(x.shape, dtype=torch.uint8)
        out.copy_(q)
        return out, scale

    # ---- normalisation -------------------------------------------------------------------------------------
    @staticmethod
    def _gn(xf, gamma, beta, B, HW, Cc, G, eps, silu):
        xg = xf.reshape(B, HW, G, Cc // G)
        mean = xg.mean(dim=(1, 3), keepdim=True)
        var = xg.var(dim=(1, 3), unbiased=False, keepdim=True)
        rstd = (var + eps).rsqrt()
        y = ((xg - mean) * rstd).reshape(B * HW, Cc) * gamma + beta
        if silu:
            y = F.silu(y)
        return y, mean.reshape(B, G), rstd.reshape(B, G)

    def groupnorm_fwd(self, x, gamma, beta, y, stats, B, HW, Cc, G, eps, silu):
        yy, mean, rstd = self._gn(x.float(), gamma, beta, B, HW, Cc, G, eps, silu)
        y.copy_(yy.to(y.dtype))
        stats.copy_(torch.stack([mean, rstd], dim=-1))

    def groupnorm_bwd(self, dy, x, gamma, beta, stats, dx, B, HW, Cc, G, silu, add=None):
        mean = stats[..., 0].reshape(B, 1, G, 1)
        rstd = stats[..., 1].reshape(B, 1, G, 1)
        xf = x.float()
        xh = ((xf.reshape(B, HW, G, Cc // G) - mean) * rstd).reshape(B * HW, Cc)
        g = dy.float()
        if silu:
            yhat = xh * gamma + beta
            s = torch.sigmoid(yhat)
            g = g * (s * (1 + yhat * (1 - s)))
        g = g * gamma
        gg = g.reshape(B, HW, G, Cc // G)
        xhg = xh.reshape(B, HW, G, Cc // G)
        s1 = gg.mean(dim=(1, 3), keepdim=True)
        s2 = (gg * xhg).mean(dim=(1, 3), keepdim=True)
        out = (rstd * (gg - s1 - xhg * s2)).reshape(B * HW, Cc)
        if add is not None:
            out = out + add.float()
        dx.copy_(out.to(dx.dtype))

    def layernorm_fwd(self, x, gamma, beta, y, stats, M, Cc, eps):
        xf = x.float()
        mean = xf.mean(-1, keepdim=True)
        rstd = (xf.var(-1, unbiased=False, keepdim=True) + eps).rsqrt()
        y.copy_((((xf - mean) * rstd) * gamma + beta).to(y.dtype))
        stats.copy_(torch.cat([mean, rstd], dim=-1))

    def layernorm_bwd(self, dy, x, gamma, stats, dx, M, Cc, add=None):
        mean, rstd = stats[:, :1], stats[:, 1:]
        xh = (x.float() - mean) * rstd
        g = dy.float() * gamma
        s1 = g.mean(-1, keepdim=True)
        s2 = (g * xh).mean(-1, keepdim=True)
        d = rstd * (g - s1 - xh * s2)
        if add is not None:
            d = d + add.float()
        dx.copy_(d.to(dx.dtype))

    # ---- softmax -------------------------------------------------------------------------------------------
    def softmax_fwd(self, S, P, rows, cols, q_len=0, causal=False, causal_offset=0, key_mask=None, rows_per_mask=0):
        s = S.reshape(rows, cols).float().clone()
        if causal:
            q = torch.arange(rows) % q_len
            j = torch.arange(cols)
            s = s.masked_fill(j[None, :] > (q[:, None] + causal_offset), float("-inf"))
        if key_mask is not None:
            km = key_mask.reshape(-1, cols).bool().repeat_interleave(rows_per_mask, 0)[:rows]
            s = s.masked_fill(~km, float("-inf"))
        p = torch.softmax(s, dim=-1)
        p = torch.nan_to_num(p, nan=0.0)
        P.reshape(rows, cols).copy_(p.to(P.dtype))

    def softmax_bwd(self, P, dP, dS, rows, cols, scale):
        p = P.reshape(rows, cols).float()
        g = dP.reshape(rows, cols).float()
        dS.reshape(rows, cols).copy_((scale * p * (g - (g * p).sum(-1, keepdim=True))).to(dS.dtype))

    @staticmethod
    def _heads(t, B, N, H, d, ld):
        return _v(t, (B, H, N, d), (N * ld, d, ld, 1)).float()

    def flash_attn_fwd(self, q, k, v, o, lse, B, H, Nq, Nk, d, ldq, ldk, ldv, ldo, scale, q8=None):
        qh, kh, vh = self._heads(q, B, Nq, H, d, ldq), self._heads(k, B, Nk, H, d, ldk), self._heads(v, B, Nk, H, d, ldv)
        s = qh @ kh.transpose(-1, -2) * scale
        lse.copy_(torch.logsumexp(s, -1))
        _v(o, (B, H, Nq, d), (Nq * ldo, d, ldo, 1)).copy_((torch.softmax(s, -1) @ vh).to(o.dtype))
        if q8 is not None:  # comat_flash_attn_fwd_q: the e4m3 bytes of the rounded output + its abs-max
            assert ldo == H * d and o.is_contiguous()
            self._quantize_scaled(o, q8[1], q8[2], out=q8[0])

    def flash_attn_bwd(self, q, k, v, o, do, lse, dbuf, dq, dk, dv, B, H, Nq, Nk, d, ldq, ldk, ldv, ldo, scale):
        qh, kh, vh = self._heads(q, B, Nq, H, d, ldq), self._heads(k, B, Nk, H, d, ldk), self._heads(v, B, Nk, H, d, ldv)
        oh, gh = self._heads(o, B, Nq, H, d, ldo), self._heads(do, B, Nq, H, d, ldo)
        p = torch.exp(qh @ kh.transpose(-1, -2) * scale - lse[..., None])
        D = (oh * gh).sum(-1)
        dbuf.copy_(D)
        dp = gh @ vh.transpose(-1, -2)
        ds = p * (dp - D[..., None]) * scale
        _v(dv, (B, H, Nk, d), (Nk * ldv, d, ldv, 1)).copy_((p.transpose(-1, -2) @ gh).to(dv.dtype))
        _v(dq, (B, H, Nq, d), (Nq * ldq, d, ldq, 1)).copy_((ds @ kh).to(dq.dtype))
        _v(dk, (B, H, Nk, d), (Nk * ldk, d, ldk, 1)).copy_((ds.transpose(-1, -2) @ qh).to(dk.dtype))

    # ---- elementwise ---------------------------------------------------------------------------------------
    def unary(self, op, x, y, n, p0=0.0, p1=0.0):
        v = x.reshape(-1)[:n].float()
        if op == UN_SILU:
            v = F.silu(v)
        elif op == UN_GELU:
            v = F.gelu(v)
        elif op == UN_AFFINE:
            v = p0 * v + p1
        y.reshape(-1)[:n].copy_(v.to(y.dtype))

    def unary_bwd(self, op, dy, x, dx, n):
        v = x.reshape(-1)[:n].float().detach().requires_grad_(True)
        with torch.enable_grad():
            out = F.silu(v) if op == UN_SILU else F.gelu(v)
            (g,) = torch.autograd.grad(out, v, dy.reshape(-1)[:n].float())
        dx.reshape(-1)[:n].copy_(g.to(dx.dtype))

    def axpby(self, a, x, b, y, out, n):
        v = a * x.reshape(-1)[:n].float()
        if y is not None:
            v = v + b * y.reshape(-1)[:n].float()
        out.reshape(-1)[:n].copy_(v.to(out.dtype))

    def geglu_fwd(self, x, y, M, D):
        xf = x.float()
        y.copy_((xf[:, :D] * F.gelu(xf[:, D:])).to(y.dtype))

    def geglu_bwd(self, dy, x, dx, M, D):
        xf = x.float().detach().requires_grad_(True)
        with torch.enable_grad():
            out = xf[:, :D] * F.gelu(xf[:, D:])
            (g,) = torch.autograd.grad(out, xf, dy.float())
        dx.copy_(g.to(dx.dtype))

    def copy2d(self, src, ld_src, dst, ld_dst, rows, cols):
        _v(dst, (rows, cols), (ld_dst, 1)).copy_(_v(src, (rows, cols), (ld_src, 1)).to(dst.dtype))

    @staticmethod
    def copy2d_pair_ok(items):
        return True

    def copy2d_pair(self, items, rows):
        for src, ld_src, dst, ld_dst, cols in items:
            self.copy2d(src, ld_src, dst, ld_dst, rows, cols)

    def add_rowvec(self, x, v, out, rows, cols):
        out.copy_((x.float() + v.float().reshape(1, cols)).to(out.dtype))

    def sumpool2x2(self, x, y, B, H, W, Cc):
        xv = x.reshape(B, H, 2, W, 2, Cc).float()
        y.copy_(xv.sum(dim=(2, 4)).reshape(B * H * W, Cc).to(y.dtype))

    def permute_nchw_nhwc(self, x, y, B, Cc, H, W, to_nhwc):
        if to_nhwc:
            y.reshape(B, H, W, Cc).copy_(x.reshape(B, Cc, H, W).permute(0, 2, 3, 1).to(y.dtype))
        else:
            y.reshape(B, Cc, H, W).copy_(x.reshape(B, H, W, Cc).permute(0, 3, 1, 2).to(y.dtype))

    def cfg_ddpm_fwd(self, x, eps2, z, x_prev, n, s, cx, ce, sigma):
        e = eps2.reshape(2, n).float()
        eps = e[0] + s * (e[1] - e[0])
        v = cx * x.reshape(-1) + ce * eps
        if z is not None:
            v = v + sigma * z.reshape(-1)
        x_prev.reshape(-1).copy_(v)

    def cfg_ddpm_bwd(self, g, dx, deps2, n, s, cx, ce):
        gf = g.reshape(-1).float()
        if dx is not None:
            dx.reshape(-1).copy_(cx * gf)
        d = deps2.reshape(2, n)
        d[0].copy_((ce * (1 - s) * gf).to(deps2.dtype))
        d[1].copy_((ce * s * gf).to(deps2.dtype))

    # ---- image path ----------------------------------------------------------------------------------------
    def resample2d(self, src, out, B, Hin, Win, Hout, Wout, Cc, ystart, ywt, xstart, xwt, KT, scale, shift):
        def dense(start, wt, n_out, n_in):
            Mx = torch.zeros((n_out, n_in))
            for o in range(n_out):
                for t in range(KT):
                    i = int(start[o]) + t
                    if 0 <= i < n_in and float(wt[o, t]) != 0.0:
                        Mx[o, i] += float(wt[o, t])
            return Mx
        Wy = dense(ystart, ywt.reshape(Hout, KT), Hout, Hin)
        Wx = dense(xstart, xwt.reshape(Wout, KT), Wout, Win)
        x = src.reshape(B, Hin, Win, Cc).float()
        y = torch.einsum("oh,bhwc->bowc", Wy, x)
        y = torch.einsum("pw,bowc->bopc", Wx, y)
        if scale is not None:
            y = y * scale
        if shift is not None:
            y = y + shift
        out.reshape(B, Hout, Wout, Cc).copy_(y.to(out.dtype))

    def patchify(self, img, patches, B, H, W, Cc, P, inverse):
        nH, nW = H // P, W // P
        if not inverse:
            v = img.reshape(B, nH, P, nW, P, Cc).permute(0, 1, 3, 2, 4, 5).reshape(B * nH * nW, P * P * Cc)
            patches.copy_(v)
        else:
            v = patches.reshape(B, nH, nW, P, P, Cc).permute(0, 1, 3, 2, 4, 5).reshape(B * H * W, Cc)
            img.copy_(v)

    def embedding(self, ids, table, out, n, dim, vocab):
        out.copy_(table[ids.clamp(0, vocab - 1)])

    # ---- losses --------------------------------------------------------------------------------------------
    def cross_entropy_fwd(self, logits, labels, logp, row_lse, loss_sum_cnt, T, V, ld, ignore_index, ls):
        z = logits.float()
        lse = torch.logsumexp(z, dim=-1)
        row_lse.copy_(lse)
        valid = (labels != ignore_index) & (labels >= 0) & (labels < V)
        y = labels.clamp(0, V - 1)
        lp = z.gather(1, y[:, None])[:, 0] - lse
        logp.copy_(torch.where(valid, lp, torch.zeros_like(lp)))
        loss = (1 - ls) * (-lp) + ls * (lse - z.mean(-1))
        loss_sum_cnt[0] = loss[valid].sum()
        loss_sum_cnt[1] = valid.sum().float()

    def cross_entropy_bwd(self, logits, labels, row_lse, dlogits, T, V, ld, ignore_index, ls, g_up, loss_sum_cnt):
        gscale = float(g_up[0]) / max(float(loss_sum_cnt[1]), 1.0)
        z = logits.float()
        valid = (labels != ignore_index) & (labels >= 0) & (labels < V)
        g = torch.exp(z - row_lse[:, None]) - ls / V
        y = labels.clamp(0, V - 1)
        g[torch.arange(T), y] -= (1 - ls)
        g = g * gscale * valid[:, None].float()
        dlogits.copy_(g.to(dlogits.dtype))

    def disc_head_fwd(self, x, w, b, target, loss, P, pix_per_sample):
        z = x.float() @ w.float() + b.float()
        t = target.float().repeat_interleave(pix_per_sample)[:P]
        loss[0] = F.binary_cross_entropy_with_logits(z, t)

    def disc_head_bwd(self, x, w, b, target, g_up, dx, dwb, P, pix_per_sample):
        gscale = float(g_up[0])
        xf = x.float()
        z = xf @ w.float() + b.float()
        t = target.float().repeat_interleave(pix_per_sample)[:P]
        dz = gscale * (torch.sigmoid(z) - t) / P
        if dx is not None:
            dx.copy_((dz[:, None] * w.float()[None, :]).to(dx.dtype))
        if dwb is not None:
            dwb[:4] += dz @ xf
            dwb[4] += dz.sum()

    def attnmap_gather_fwd(self, amap, mask, tok_idx, tok_obj, num, den, avg, heads, npix, L, n_tok):
        a = amap.float()[:, :, tok_idx.long()]  # [h, npix, n_tok]
        m = mask[tok_obj.long()]  # [n_tok, npix]
        num += torch.einsum("hpt,tp->ht", a, m)
        den += a.sum(1)
        avg += a.mean(0).t()

    def attnmap_gather_bwd(self, g_num, g_den, g_avg, mask, tok_idx, tok_obj, damap, heads, npix, L, n_tok):
        m = mask[tok_obj.long()]  # [n_tok, npix]
        g = g_num[:, None, :] * m.t()[None] + g_den[:, None, :]
        if g_avg is not None:
            g = g + g_avg.t()[None] / heads
        d = torch.zeros(damap.shape, dtype=torch.float32)  # the kernel writes every element of dA
        for t in range(n_tok):
            d[:, :, int(tok_idx[t])] += g[:, :, t]
        damap.copy_(d.to(damap.dtype))

    # ---- optimizer -----------------------------------------------------------------------------------------
    def sumsq(self, x, n, out):
        out[0] += (x.reshape(-1)[:n].double() ** 2).sum().float()

    def adamw_tick(self, counters, gnorm_sq):
        counters[0 if math.isfinite(float(gnorm_sq[0])) else 1] += 1

    def adamw(self, p, g, m, v, n, lr, beta1, beta2, eps, wd, step, gnorm_sq, max_norm, step_dev=None, grad_scale=1.0):
        if step_dev is not None:
            step = int(step_dev[0]) + 1
        clip = grad_scale
        if gnorm_sq is not None and not math.isfinite(float(gnorm_sq[0])):
            return  # non-finite gradient norm: the update is skipped
        if gnorm_sq is not None and max_norm > 0:
            clip = grad_scale * min(1.0, max_norm / (math.sqrt(float(gnorm_sq[0])) * grad_scale + 1e-6))
        gg = g * clip
        m.mul_(beta1).add_(gg, alpha=1 - beta1)
        v.mul_(beta2).addcmul_(gg, gg, value=1 - beta2)
        bc1 = 1 - beta1 ** step
        bc2s = math.sqrt(1 - beta2 ** step)
        p.mul_(1 - lr * wd)
        p.addcdiv_(m, v.sqrt() / bc2s + eps, value=-lr / bc1)
