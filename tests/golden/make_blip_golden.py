"""Generates tests/golden/blip_tiny.npz: golden vectors from transformers' own BlipForConditionalGeneration
(tiny config, weights from comat_amd.weights.make_blip_weights, seed fixed).  Run in the build container only:
    python tests/golden/make_blip_golden.py
The fixture holds inputs + expected outputs (data); nothing of transformers ships."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from comat_amd import config, weights  # noqa: E402
from transformers import BlipConfig, BlipForConditionalGeneration  # noqa: E402


def main():
    cfg = config.TINY_BLIP
    sd = weights.make_blip_weights(cfg, seed=3456, perturb_norms=True)
    out = {}
    for ls in (0.0, 0.1):
        hf_cfg = BlipConfig(
            vision_config=dict(hidden_size=cfg.v_hidden, intermediate_size=cfg.v_mlp, num_hidden_layers=cfg.v_layers,
                               num_attention_heads=cfg.v_heads, image_size=cfg.image_size, patch_size=cfg.patch_size,
                               layer_norm_eps=cfg.v_eps, hidden_act="gelu"),
            text_config=dict(vocab_size=cfg.vocab_size, hidden_size=cfg.t_hidden, encoder_hidden_size=cfg.v_hidden,
                             intermediate_size=cfg.t_mlp, num_hidden_layers=cfg.t_layers,
                             num_attention_heads=cfg.t_heads, max_position_embeddings=cfg.max_pos,
                             layer_norm_eps=cfg.t_eps, hidden_act="gelu", is_decoder=True, label_smoothing=ls,
                             hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, pad_token_id=0,
                             bos_token_id=cfg.vocab_size - 2, sep_token_id=cfg.vocab_size - 1),
        )
        model = BlipForConditionalGeneration(hf_cfg).eval()
        full = dict(sd)
        full["text_decoder.cls.predictions.decoder.weight"] = sd["text_decoder.bert.embeddings.word_embeddings.weight"]
        full["text_decoder.cls.predictions.decoder.bias"] = sd["text_decoder.cls.predictions.bias"]
        missing, unexpected = model.load_state_dict(full, strict=False)
        missing = [m for m in missing if "position_ids" not in m]
        assert not missing and not unexpected, (missing, unexpected)
        g = torch.Generator().manual_seed(7)
        B, T = 2, 9
        pv = torch.randn(B, 3, cfg.image_size, cfg.image_size, generator=g)
        ids = torch.randint(1, cfg.vocab_size, (B, T), generator=g)
        ids[1, 7:] = 0  # padded second caption
        am = (ids != 0).long()
        labels = ids.masked_fill(ids == 0, -100)
        labels[:, :4] = -100
        pvr = pv.clone().requires_grad_(True)
        o = model(pixel_values=pvr, input_ids=ids, attention_mask=am, labels=labels)
        o.loss.backward()
        sl = o.logits[:, :-1]
        tgt = labels[:, 1:]
        lp = torch.log_softmax(sl, -1).gather(-1, tgt.clamp(min=0)[..., None])[..., 0] * (tgt != -100)
        tag = f"ls{int(ls * 10)}"
        out.update({f"{tag}_loss": o.loss.detach().numpy(), f"{tag}_logits": o.logits.detach().numpy(),
                    f"{tag}_logp": lp.detach().numpy(), f"{tag}_dpv": pvr.grad.numpy(),
                    f"{tag}_image_embeds": o.image_embeds.detach().numpy()})
        out.update(pixel_values=pv.numpy(), input_ids=ids.numpy(), attention_mask=am.numpy(), labels=labels.numpy())
    for k, v in sd.items():
        out["w:" + k] = v.numpy()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "blip_tiny.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
