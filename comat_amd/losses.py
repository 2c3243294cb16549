"""Attribute-concentration losses on the captured cross-attention maps (HIP gather kernel + tiny reductions).

Mirrors `get_grounding_loss_by_layer` (attn_utils/tc_loss_utils.py:66-173) and the per-sample / per-timestep /
per-layer assembly of `GsamSegModel.get_mask_loss` (attr_concen_utils/gsam_interface.py:140-228).  The object masks
(FastSAM + GroundingDINO, out of scope) and the attribute token lists (spaCy, out of scope) are INPUTS.
One pass of the strided gather kernel over each captured map produces, for all attribute tokens at once, the masked
and unmasked spatial sums per head (token loss) and the head-mean map per token (pixel loss); the remaining
reductions run on [heads, n_tok] and [n_tok, res*res] tensors.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import ops
from .resize import aa_taps


def _apply_taps(x: np.ndarray, start: np.ndarray, wts: np.ndarray) -> np.ndarray:
    """x [n, S, W] -> [n, res, W]: out[:, o] = sum_t wts[o, t] * x[:, start[o] + t]  (taps beyond the source carry
    weight 0).  <= 17 taps per output row: a gather + small contraction, no dense [res, S] GEMM on the host."""
    kt = wts.shape[1]
    idx = np.minimum(start[:, None].astype(np.int64) + np.arange(kt)[None], x.shape[1] - 1)  # [res, kt]
    return np.einsum("ok,nokw->now", wts.astype(np.float64), x[:, idx, :])


def resize_masks(masks: np.ndarray, res: int) -> np.ndarray:
    """masks [n_obj, H, W] bool -> [n_obj, res*res] float {0,1}: torchvision Resize(antialias=True) on a bool mask
    followed by `> 0` (tc_loss_utils.py:88-98)."""
    n, H, W = masks.shape
    ys, yw, _ = aa_taps(H, res, "bilinear")
    xs, xw, _ = aa_taps(W, res, "bilinear")
    rows = _apply_taps(masks.astype(np.float64), ys, yw)                              # [n, res, W]
    out = _apply_taps(np.ascontiguousarray(rows.transpose(0, 2, 1)), xs, xw)          # [n, res(x), res(y)]
    return (out.transpose(0, 2, 1) > 0.0).astype(np.float32).reshape(n, res * res)


def grounding_loss_by_layer(masks_res: torch.Tensor, word_token_idx_ls, res, attn_maps):
    """masks_res: [n_obj, res*res] fp32 {0,1} on the device; attn_maps: list of [heads, res, res, L] maps of ONE
    sample (views of the stored probabilities).  Returns (token_loss, pixel_loss) scalars."""
    n_obj = len(word_token_idx_ls)
    dev = masks_res.device
    if n_obj == 0:
        z = torch.zeros((), device=dev)
        return z, z
    tok_idx = torch.tensor([t for w in word_token_idx_ls for t in w], dtype=torch.int32, device=dev)
    tok_obj = torch.tensor([i for i, w in enumerate(word_token_idx_ls) for _ in w], dtype=torch.int32, device=dev)
    inv_len = torch.tensor([1.0 / len(w) for w in word_token_idx_ls for _ in w], dtype=torch.float32, device=dev)
    onehot = F.one_hot(tok_obj.long(), n_obj).float()  # [n_tok, n_obj]
    token_loss = torch.zeros((), device=dev)
    avg_sum = None
    for a in attn_maps:
        h = a.shape[0]
        num, den, avg = ops.attnmap_gather(a.reshape(h, res * res, a.shape[-1]), masks_res, tok_idx, tok_obj)
        act = (num / den).mean(0)  # [n_tok]
        token_loss = token_loss + (((1.0 - act) ** 2) * inv_len).sum()
        avg_sum = avg if avg_sum is None else avg_sum + avg
    token_loss = token_loss / n_obj
    # [n_obj, npix]: the tokens of one object are summed - a masked reduction over <= a handful of tokens, in a fixed
    # order (no vendor GEMM on the path, no atomics)
    word = (onehot.t()[:, :, None] * (avg_sum / len(attn_maps))[None]).sum(1)
    pixel_loss = F.binary_cross_entropy(word, masks_res, reduction="none").mean(1).sum() / n_obj
    return token_loss, pixel_loss


def mask_loss(attn_dict, masks_per_sample, attributes_per_sample, train_layer_ls, bs, device):
    """`get_mask_loss` with detector outputs as inputs.  attn_dict: {timestep: {place_res: [(bs*heads,res,res,L)]}};
    masks_per_sample[i]: [n_obj, H, W] bool numpy array or None; attributes_per_sample[i]: list[n_obj] of token
    index lists.  Returns (token_loss, pixel_loss) averaged over the batch (gsam_interface.py:225-226)."""
    token_loss = torch.zeros((), device=device)
    pixel_loss = torch.zeros((), device=device)
    for idx in range(bs):
        masks, attrs = masks_per_sample[idx], attributes_per_sample[idx]
        if masks is None or len(attrs) == 0:
            continue
        cache = {}
        for ts in attn_dict:
            for layer in train_layer_ls:
                res = int(layer.split("_")[1])
                if res not in cache:
                    cache[res] = torch.from_numpy(resize_masks(masks, res)).to(device)
                maps = [m.reshape(bs, m.shape[0] // bs, *m.shape[1:])[idx] for m in attn_dict[ts][layer]]
                tl, pl = grounding_loss_by_layer(cache[res], attrs, res, maps)
                token_loss = token_loss + tl
                pixel_loss = pixel_loss + pl
    return token_loss / bs, pixel_loss / bs
