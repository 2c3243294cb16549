"""Attribute-concentration loss assembly (gsam_interface.py:140-228 / tc_loss_utils.py:66-173): ragged and empty cases
of the product (comat_amd/losses.py, HIP gather kernel) against the oracle (oracle/losses.py, pinned to the reference's
own get_grounding_loss_by_layer by tests/golden/grounding_loss.npz)."""
import numpy as np
import pytest
import torch

from comat_amd import losses
from helpers import check
from oracle import losses as OL

DTYPES = [torch.float32, torch.bfloat16]


def _maps(bs, heads, reses, L, n_per, seed, dtype):
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, res in reses:
        out[name] = [torch.softmax(torch.randn(bs * heads, res, res, L, generator=g) * 2, -1).to(dtype).float()
                     for _ in range(n_per)]
    return out


@pytest.mark.parametrize("dtype", DTYPES)
def test_mask_loss_ragged_batch(dev, dtype):
    """bs=3: a sample with two objects (one single-token attribute), a sample without masks (detector found nothing:
    skipped), a sample with one object; two captured timesteps; maps of three layer keys with 1..3 maps each.
    Checks values and the gradient that flows back into every captured map."""
    bs, heads, L, H = 3, 2, 11, 48
    layers = ("mid_4", "up_8", "up_16")
    reses = (("mid_4", 4), ("up_8", 8), ("up_16", 16))
    attn_o = {"41": _maps(bs, heads, reses, L, 2, 1, dtype), "1": _maps(bs, heads, reses, L, 3, 2, dtype)}
    for ts in attn_o:
        for k in attn_o[ts]:
            attn_o[ts][k] = [m.requires_grad_(True) for m in attn_o[ts][k]]
    m0 = np.zeros((2, H, H), dtype=bool)
    m0[0, 3:20, 5:30] = True
    m0[1, 25:47, 10:40] = True
    m2 = np.zeros((1, H, H), dtype=bool)
    m2[0, 0:9, 40:48] = True
    masks = [m0, None, m2]
    attrs = [[[2, 3, 4], [7]], [[1]], [[5, 6]]]
    om = [None if m is None else [torch.from_numpy(x)[None, None] for x in m] for m in masks]
    tl_o, pl_o = OL.mask_loss(attn_o, om, attrs, layers, bs)
    (tl_o + 0.3 * pl_o).backward()
    attn_d = {ts: {k: [m.detach().to(dev, dtype).requires_grad_(True) for m in v] for k, v in d.items()}
              for ts, d in attn_o.items()}
    tl, pl = losses.mask_loss(attn_d, masks, attrs, layers, bs, dev)
    (tl + 0.3 * pl).backward()
    check(tl, tl_o, torch.float32, "token loss", factor=10 if dtype == torch.bfloat16 else 1)
    check(pl, pl_o, torch.float32, "pixel loss", factor=10 if dtype == torch.bfloat16 else 1)
    for ts in attn_o:
        for k in attn_o[ts]:
            for a, b in zip(attn_d[ts][k], attn_o[ts][k]):
                check(a.grad, b.grad, dtype, f"d loss / d map {ts} {k}", factor=2)
    # the skipped sample contributes exactly nothing
    a = attn_d["1"]["up_8"][0].grad.reshape(bs, heads, 8, 8, L)
    assert float(a[1].abs().max()) == 0.0 and float(a[0].abs().max()) > 0.0


def test_mask_loss_empty_cases(sim):
    """no attributes at all / every sample without masks -> zero losses, no kernel work, no error."""
    bs, heads, L = 2, 2, 9
    attn = {"1": _maps(bs, heads, (("up_8", 8),), L, 1, 3, torch.float32)}
    m = np.ones((1, 16, 16), dtype=bool)
    for masks, attrs in (([None, None], [[[1]], [[2]]]), ([m, m], [[], []])):
        tl, pl = losses.mask_loss(attn, masks, attrs, ("up_8",), bs, sim)
        assert float(tl) == 0.0 and float(pl) == 0.0
    tl, pl = losses.grounding_loss_by_layer(torch.zeros(0, 64), [], 8, attn["1"]["up_8"])
    assert float(tl) == 0.0 and float(pl) == 0.0


def test_resize_masks_matches_dense_operator():
    """host-side separable tap resize == the dense [res, H] operator form, for non-square sources and every level"""
    from comat_amd.resize import aa_taps, dense_from_taps
    rng = np.random.default_rng(0)
    m = rng.random((3, 96, 72)) > 0.97
    for res in (64, 32, 16, 8):
        ys, yw, _ = aa_taps(96, res, "bilinear")
        xs, xw, _ = aa_taps(72, res, "bilinear")
        ref = np.einsum("oh,nhw,pw->nop", dense_from_taps(ys, yw, 96), m.astype(np.float64), dense_from_taps(xs, xw, 72),
                        optimize=True)
        assert np.array_equal(losses.resize_masks(m, res), (ref > 0).astype(np.float32).reshape(3, -1))


def test_mask_loss_against_the_reference_assembly(dev):
    """the PRODUCT's attribute-concentration loss (gather kernel of the C ABI on its CPU simulator, host assembly of
    comat_amd/losses.py, noun / attribute lists of comat_amd/attr_index.py) against tests/golden/mask_loss.npz = the totals of
    the reference's own `get_mask_loss` (see tests/test_oracle.py::test_mask_loss_assembly_matches_reference)."""
    import os

    import numpy as np

    from comat_amd import attr_index
    from comat_amd.losses import mask_loss
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "mask_loss.npz"))
    bs, layers = int(d["bs"]), [str(s) for s in d["layers"]]
    subtrees = [[[2, 3], [6, 7]], [[2, 3]], [[1, 2], [4, 5], [8], [10, [11, 12]]]]
    pieces = [{1: "a", 2: "red", 3: "car", 4: "and", 5: "a", 6: "blue", 7: "dog"},
              {1: "a", 2: "tall", 3: "tree"},
              {1: "a", 2: "cat", 3: "a", 4: "big", 5: "cat", 6: "and", 7: "the", 8: "sky", 9: "a", 10: "green", 11: "skate", 12: "board"}]
    got = [attr_index.nouns_and_attributes(st, pc) for st, pc in zip(subtrees, pieces)]
    attn_dict = {}
    for key in d.files:
        if key.startswith("map:"):
            _, ts, place = key.split(":")
            attn_dict.setdefault(ts, {})[place] = [torch.from_numpy(m).to(dev) for m in d[key]]
    masks = [np.concatenate([d[f"mask:{n}"][0] for n in nouns]) if "tree" not in nouns else None for nouns, _ in got]
    tl, pl = mask_loss(attn_dict, masks, [a for _, a in got], layers, bs, dev)
    assert abs(float(tl) - float(d["token_loss"])) < 2e-5 * max(1.0, abs(float(d["token_loss"])))
    assert abs(float(pl) - float(d["pixel_loss"])) < 2e-5 * max(1.0, abs(float(d["pixel_loss"])))
