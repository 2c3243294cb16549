"""ORACLE (test infrastructure, never imported by the product): CPU fp32 restatement of one CoMat optimisation step,
`training_script.py:556-694`, composed from oracle/sd.py, oracle/blip.py and oracle/losses.py.  PINNED by the reference's own
loop body executed on stand-in networks (tests/golden/step_body.npz, tests/test_oracle.py::
test_whole_step_matches_the_reference_loop_body): gradients, both parameter updates, logged loss terms.  This is also the
"reference CPU path" timed by bench.py's `cpu_baseline` leg (kind "port": the reference's own Python cannot run here —
diffusers / torchvision / weights are absent)."""
from __future__ import annotations

import torch

from . import blip as OB
from . import losses as OL
from . import sd as O


def g_loss_terms(W, batch, cfg, training_steps, crop, attrcon_steps=()):
    """W: dict(unet, vae, blip, lora, d_unet, d_lora, head_w, head_b, ucfg, vcfg, bcfg[, d_ucfg: the discriminator's
    UNet config when it differs from the generator's — SDXL generator + SD1.5 discriminator; fp8_unet: True = generator UNet
    forward on emulated fp8 operands, oracle/sd.py fp8_forward]).  cfg: the product's StepConfig
    (duck-typed).  Returns dict with loss, Blip (reward), G_loss, token_loss, pixel_loss, image, latents, logp."""
    kw = {}
    if cfg.attrcon:
        kw = dict(attrcon_steps=attrcon_steps, train_layer_ls=cfg.train_layer_ls, reses=cfg.attn_reses)
    if "pooled_prompt_embeds" in batch:  # SDXL generator (TrainableSDPipeline.py:657-846)
        kw["sdxl_cond"] = (batch["negative_pooled_prompt_embeds"], batch["pooled_prompt_embeds"],
                           torch.tensor([list(batch["add_time_ids"])], dtype=torch.float32))
    img, lat, attn_dict = O.sample_with_grad(W["unet"], W["ucfg"], W["vae"], W["vcfg"], W["lora"],
                                             batch["negative_prompt_embeds"], batch["prompt_embeds"],
                                             batch["latents"], batch["noises"], cfg.total_step, training_steps,
                                             cfg.cfg_scale, fp8_unet=bool(W.get("fp8_unet", False)), **kw)
    y0, x0, ch, cw = crop
    reward, logp = OB.score(W["blip"], W["bcfg"], img[:, :, y0:y0 + ch, x0:x0 + cw], batch["blip_input_ids"],
                            batch["blip_attention_mask"], label_smoothing=cfg.label_smoothing)
    out = dict(Blip=reward, token_logp=logp, image=img, latents=lat)
    loss = -reward
    if cfg.gan_loss:
        G = OL.gan_g_loss(W["d_unet"], W.get("d_ucfg", W["ucfg"]), W["d_lora"], W["head_w"], W["head_b"], lat, batch["gan_null_embeds"],
                          cfg.total_step)
        loss = loss + cfg.gan_loss_weight * G
        out["G_loss"] = G
    if cfg.attrcon:
        bs = batch["prompt_embeds"].shape[0]
        masks = [None if m is None else [torch.from_numpy(x)[None, None] for x in m] for m in batch["masks"]]
        tl, pl = OL.mask_loss(attn_dict, masks, batch["attributes"], cfg.train_layer_ls, bs)
        loss = loss + cfg.mask_token_loss_weight * tl + cfg.mask_pixel_loss_weight * pl
        out["token_loss"], out["pixel_loss"] = tl, pl
    out["loss"] = loss
    return out


def d_loss(W, batch, cfg, fake_latents):
    return OL.gan_d_loss(W["d_unet"], W.get("d_ucfg", W["ucfg"]), W["d_lora"], W["head_w"], W["head_b"], fake_latents,
                         batch["real_latents"], batch["gan_null_embeds"], cfg.total_step)


def train_step(W, batch, cfg, training_steps, crop, attrcon_steps=(), opt=None, opt_D=None):
    """Full step with torch optimizers on the LoRA leaves (training_script.py:658-664,689-694)."""
    g_params = list(W["lora"].values())
    d_params = list(W["d_lora"].values()) + [W["head_w"], W["head_b"]] if cfg.gan_loss else []
    for p in g_params + d_params:
        p.grad = None
    out = g_loss_terms(W, batch, cfg, training_steps, crop, attrcon_steps)
    out["loss"].backward()
    out["g_grads"] = {k: v.grad.clone() for k, v in W["lora"].items()}
    if opt is not None:
        torch.nn.utils.clip_grad_norm_(g_params, cfg.max_grad_norm)
        opt.step()
    if cfg.gan_loss:
        for p in d_params:
            p.grad = None
        D = d_loss(W, batch, cfg, out["latents"].detach())
        D.backward()
        out["D_loss"] = D
        out["d_grads"] = {k: v.grad.clone() for k, v in W["d_lora"].items()}
        out["head_grads"] = (W["head_w"].grad.clone(), W["head_b"].grad.clone())
        if opt_D is not None:
            torch.nn.utils.clip_grad_norm_(d_params, cfg.max_grad_norm_D)
            opt_D.step()
    return out
