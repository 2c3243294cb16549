"""BLIP concept-matching loss: comat_amd.blip (HIP kernels / ABI simulator) against the golden vectors produced by
transformers' BlipForConditionalGeneration (tests/golden/blip_tiny.npz) and against the CPU oracle for the
crop + resize + normalise front end."""
import dataclasses
import os

import numpy as np
import pytest
import torch

from comat_amd import config
from comat_amd.blip import Blip
from helpers import check, tok
from oracle import blip as OB

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DTYPES = [torch.float32, torch.bfloat16]


def load_gold(dtype):
    d = np.load(os.path.join(GOLD, "blip_tiny.npz"))
    sd = {k[2:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("w:")}
    return d, sd


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("ls", [0.0, 0.1])
def test_caption_loss_matches_transformers_golden(dev, dtype, ls):
    d, sd = load_gold(dtype)
    tag = f"ls{int(ls * 10)}"
    model = Blip(config.TINY_BLIP, sd, dtype, dev)
    pv = torch.from_numpy(d["pixel_values"])
    B = pv.shape[0]
    ids, am, labels = (torch.from_numpy(d[k]) for k in ("input_ids", "attention_mask", "labels"))
    pd = tok(pv).to(dev, dtype).requires_grad_(True)
    loss, logits, logp = model.caption_loss(pd, B, ids, am, labels, label_smoothing=ls)
    loss.backward()
    f = 1.0 if dtype == torch.float32 else 3.0
    check(loss, torch.from_numpy(d[tag + "_loss"]), dtype, "loss", factor=f)
    T, V = ids.shape[1], config.TINY_BLIP.vocab_size
    check(logits.reshape(B, T, V), torch.from_numpy(d[tag + "_logits"]), dtype, "logits", factor=f)
    check(logp, torch.from_numpy(d[tag + "_logp"]), dtype, "token log-probs (concept scores)", factor=f)
    check(pd.grad, tok(torch.from_numpy(d[tag + "_dpv"])), dtype, "d loss / d pixel_values", factor=3 * f)


@pytest.mark.parametrize("dtype", DTYPES)
def test_score_with_crop_matches_oracle(dev, dtype):
    d, sd = load_gold(dtype)
    sdq = {k: v.to(dtype).float() for k, v in sd.items()}
    ocfg = OB.BlipConfig(**dataclasses.asdict(config.TINY_BLIP))
    B, H = 2, 44
    g = torch.Generator().manual_seed(3)
    img = (torch.rand(B, 3, H, H, generator=g) * 1.4 - 0.2).to(dtype).float()  # unclamped, like the VAE output
    ids, am = torch.from_numpy(d["input_ids"]), torch.from_numpy(d["attention_mask"])
    y0, x0, size = 1, 2, 41
    io = img.clone().requires_grad_(True)
    reward_o, lp_o = OB.score(sdq, ocfg, io[:, :, y0:y0 + size, x0:x0 + size], ids, am, label_smoothing=0.1)
    reward_o.backward()
    model = Blip(config.TINY_BLIP, sdq, dtype, dev)
    idv = tok(img).to(dev, dtype).requires_grad_(True)
    reward, lp = model.score(idv, B, H, H, ids, am, crop=(y0, x0, size, size), label_smoothing=0.1)
    reward.backward()
    f = 1.0 if dtype == torch.float32 else 3.0
    check(reward, reward_o, dtype, "reward", factor=f)
    check(lp, lp_o, dtype, "token log-probs", factor=f)
    check(idv.grad, tok(io.grad), dtype, "d reward / d image", factor=3 * f)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_qkv_vit_matches_split_projections(dev, dtype):
    """BLIP ViT with q/k/v as ONE GEMM + strided fused attention (opt-in) == the three-projection path: reward,
    token log-probs and the gradient that flows back into the image."""
    from comat_amd import config, weights
    from comat_amd.blip import Blip
    bsd = {k: v.to(dtype).float() for k, v in weights.make_blip_weights(config.TINY_BLIP, perturb_norms=True).items()}
    g = torch.Generator().manual_seed(3)
    B, S = 2, 40
    img = torch.rand(B * S * S, 3, generator=g).to(dtype)
    ids = torch.randint(1, config.TINY_BLIP.vocab_size, (B, 9), generator=g)
    outs = []
    for fused in (False, True):
        blip = Blip(config.TINY_BLIP, bsd, dtype, dev, fused_qkv=fused)
        assert blip.fused_qkv == fused
        x = img.clone().to(dev).requires_grad_(True)
        reward, logp = blip.score(x, B, S, S, ids, torch.ones_like(ids), crop=(1, 2, 36, 36), label_smoothing=0.1)
        reward.backward()
        outs.append((reward.detach().float().cpu(), logp.detach().float().cpu(), x.grad.float().cpu()))
    tol = 1e-5 if dtype == torch.float32 else 3e-2
    for a, b, name in zip(outs[0], outs[1], ("reward", "token log-probs", "d reward / d image")):
        assert (a - b).abs().max() <= tol * max(float(a.abs().max()), 1e-6), name


def test_caption_labels_match_the_reference_score():
    """tests/golden/blip_labels.json = the labels the reference's own `Blip.score` (concept_mat_utils/caption_blip.py:43-59,
    executed by tests/golden/make_blip_labels_golden.py with the processor / model calls recorded) hands to the captioner
    for given token ids: pads and the 'a photography of' prefix ignored; reward = -loss.  Product and oracle build the same."""
    import json
    import os

    from comat_amd.blip import Blip
    from oracle import blip as OBL
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "blip_labels.json")))
    ids = torch.tensor(g["input_ids"])
    want = torch.tensor(g["labels"])
    assert torch.equal(Blip.make_labels(ids, g["pad_token_id"], g["prompt_length"]), want)
    assert torch.equal(OBL.make_labels(ids, g["pad_token_id"], g["prompt_length"]), want)
    assert torch.equal(ids, torch.tensor(g["input_ids"]))  # the ids themselves stay untouched
    assert g["reward"] == -g["loss"] and g["text"][0] == "a photography of a red car"
    assert g["model_inputs"] == ["attention_mask", "input_ids", "labels", "pixel_values"]
