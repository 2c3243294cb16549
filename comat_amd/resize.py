"""Tap tables of separable antialiased resampling (host side, numpy).

Restates the index/weight computation of ATen's antialiased upsampling (`F.interpolate(..., antialias=True,
align_corners=False)`, which is what torchvision's `Resize(antialias=True)` calls on tensors — the reference's
`Resize(size=(384,384), interpolation=BICUBIC, antialias=True)`, concept_mat_utils/caption_blip.py:33-36, and the
bilinear mask resize of attn_utils/tc_loss_utils.py:88):
    scale   = in/out;  support = (interp_size/2) * max(scale, 1);  center = scale*(i + 0.5)
    xmin    = max(int(center - support + 0.5), 0);  xsize = min(int(center + support + 0.5), in) - xmin
    w_j     = filter((j + xmin - center + 0.5) / max(scale, 1)),  normalised to sum 1
with the cubic filter at a = -0.5 (bicubic) or the triangle filter (bilinear).
A crop of the source (training_script.py:606-611) is folded in as an index offset, so crop + resize is ONE operator
on the un-cropped image; `transpose_tables` gives the same operator's adjoint in the same table format.
"""
from __future__ import annotations

import math

import numpy as np


def _cubic(x, a=-0.5):
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0
    if x < 2.0:
        return (((x - 5.0) * x + 8.0) * x - 4.0) * a
    return 0.0


def _triangle(x):
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


def aa_taps(in_size: int, out_size: int, mode: str = "bicubic", offset: int = 0):
    """Returns (start[out] int32, weights[out, KT] float32) of the 1-D operator; `offset` shifts source indices
    (the resize acts on src[offset : offset + in_size])."""
    filt, interp = (_cubic, 4) if mode == "bicubic" else (_triangle, 2)
    scale = in_size / out_size
    support = (interp * 0.5) * scale if scale >= 1.0 else interp * 0.5
    invscale = 1.0 / scale if scale >= 1.0 else 1.0
    KT = int(math.ceil(support)) * 2 + 1
    start = np.zeros((out_size,), dtype=np.int32)
    wts = np.zeros((out_size, KT), dtype=np.float64)
    for i in range(out_size):
        center = scale * (i + 0.5)
        xmin = max(int(center - support + 0.5), 0)
        xsize = min(int(center + support + 0.5), in_size) - xmin
        w = np.array([filt((j + xmin - center + 0.5) * invscale) for j in range(xsize)], dtype=np.float64)
        tot = w.sum()
        if tot != 0.0:
            w = w / tot
        start[i] = xmin + offset
        wts[i, :xsize] = w
    return start, wts.astype(np.float32), KT


def dense_from_taps(start, wts, n_in):
    n_out, KT = wts.shape
    M = np.zeros((n_out, n_in), dtype=np.float64)
    for o in range(n_out):
        for t in range(KT):
            i = int(start[o]) + t
            if 0 <= i < n_in:
                M[o, i] += float(wts[o, t])
    return M


def transpose_taps(start, wts, n_in):
    """Adjoint operator in the same (start, weights) format: rows indexed by source position."""
    M = dense_from_taps(start, wts, n_in)  # [n_out, n_in]
    Mt = M.T  # [n_in, n_out]
    n_out = M.shape[0]
    nz = [np.nonzero(Mt[i])[0] for i in range(n_in)]
    KT = max([int(z[-1] - z[0] + 1) if len(z) else 1 for z in nz])
    s = np.zeros((n_in,), dtype=np.int32)
    w = np.zeros((n_in, KT), dtype=np.float32)
    for i in range(n_in):
        if len(nz[i]) == 0:
            continue
        s[i] = nz[i][0]
        seg = Mt[i, nz[i][0]: nz[i][-1] + 1]
        w[i, : len(seg)] = seg
    return s, w, KT


def resize_tables(Hin_full, Win_full, crop, out_hw, mode="bicubic"):
    """crop = (y0, x0, h, w) of the full image that is resized to out_hw.  Returns (fwd, bwd) dicts for
    comat_amd.ops.ResampleTables; both operate on / produce the FULL image grid."""
    y0, x0, ch, cw = crop
    Ho, Wo = out_hw
    ys, yw, kty = aa_taps(ch, Ho, mode, y0)
    xs, xw, ktx = aa_taps(cw, Wo, mode, x0)
    KT = max(kty, ktx)

    def padw(w, K):
        out = np.zeros((w.shape[0], K), dtype=np.float32)
        out[:, : w.shape[1]] = w
        return out
    fwd = dict(ystart=ys, ywt=padw(yw, KT), xstart=xs, xwt=padw(xw, KT), KT=KT)
    tys, tyw, tky = transpose_taps(ys, yw, Hin_full)
    txs, txw, tkx = transpose_taps(xs, xw, Win_full)
    KTb = max(tky, tkx)
    bwd = dict(ystart=tys, ywt=padw(tyw, KTb), xstart=txs, xwt=padw(txw, KTb), KT=KTb)
    return fwd, bwd
