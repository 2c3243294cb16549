"""ORACLE (test infrastructure, never imported by the product): CPU fp32 restatement of CoMat's loss heads.

  - attribute-concentration grounding loss: attn_utils/tc_loss_utils.py:66-173 (`get_grounding_loss_by_layer`),
    PINNED by tests/golden/grounding_loss.npz (outputs of the reference function itself, torchvision Resize shimmed —
    parity unpinned at that torchvision boundary: bool masks become `bilinear-antialias(mask) > 0`);
  - per-sample / per-timestep / per-layer assembly: attr_concen_utils/gsam_interface.py:140-228 (`get_mask_loss`),
    with the object masks (FastSAM + GroundingDINO, out of scope) and the attribute token lists as inputs; PINNED by
    tests/golden/mask_loss.npz (totals of the reference method itself, detector call replaced by prepared masks);
  - GAN fidelity discriminator: training_utils/gan_sdxl.py:50-132 (`D_sd.D_sd_pipeline_forward`, G and D sides),
    head `nn.Linear(4, 1)` on the NHWC-permuted UNet output + BCEWithLogitsLoss (:32-35); PINNED by
    tests/golden/gan_losses.npz (losses and gradients of the reference method itself on a stand-in UNet).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import sd as O


def resize_mask(mask, res):
    """tc_loss_utils.py:88-98: Resize((res,res), antialias=True) on a (1,1,H,W) bool mask, then `> 0`."""
    y = F.interpolate(mask.float(), size=(res, res), mode="bilinear", align_corners=False, antialias=True)
    return (y.squeeze(0) > 0.0).float()  # (1, res, res)


def grounding_loss_by_layer(gt_seg_list, word_token_idx_ls, res, attn_maps):
    """gt_seg_list: list[n_obj] of (1,1,H,W) masks; word_token_idx_ls: list[n_obj] of token index lists;
    attn_maps: list of (heads, res, res, L) maps of ONE sample.  Returns (token_loss, pixel_loss)."""
    if len(word_token_idx_ls) == 0:
        return torch.zeros(()), torch.zeros(())
    masks = [resize_mask(m, res) for m in gt_seg_list]
    token_loss = 0.0
    for a in attn_maps:
        b, H, W, _ = a.shape
        for i, idxs in enumerate(word_token_idx_ls):
            obj_loss = 0.0
            for pos in idxs:
                ca = a[:, :, :, pos].reshape(b, H, W)
                act = (ca * masks[i]).reshape(b, -1).sum(-1) / ca.reshape(b, -1).sum(-1)
                obj_loss = obj_loss + (1.0 - act.mean()) ** 2
            token_loss = token_loss + obj_loss / len(idxs)
    token_loss = token_loss / len(word_token_idx_ls)
    avg = torch.stack([a.reshape(-1, res, res, a.shape[-1]).mean(0) for a in attn_maps], 0)
    avg = (avg.sum(0) / avg.shape[0]).unsqueeze(0)  # (1, res, res, L)
    pixel_loss = 0.0
    for i, idxs in enumerate(word_token_idx_ls):
        word = torch.stack([avg[..., t] for t in idxs], 0).sum(0)
        pixel_loss = pixel_loss + F.binary_cross_entropy(word, masks[i])
    pixel_loss = pixel_loss / len(word_token_idx_ls)
    return token_loss, pixel_loss


def mask_loss(attn_dict, masks_per_sample, attributes_per_sample, train_layer_ls, bs):
    """gsam_interface.py:140-228 with the detector outputs as inputs.
    attn_dict: {timestep: {place_res: [ (bs*heads, res, res, L) ]}};  masks_per_sample[i]: list[n_obj] of
    (1,1,H,W) bool masks or None;  attributes_per_sample[i]: list[n_obj] of token lists."""
    token_loss = torch.zeros(())
    pixel_loss = torch.zeros(())
    for idx in range(bs):
        masks, attrs = masks_per_sample[idx], attributes_per_sample[idx]
        if masks is None or len(attrs) == 0:
            continue
        for ts in attn_dict:
            for layer in train_layer_ls:
                res = int(layer.split("_")[1])
                maps = [m.reshape(bs, m.shape[0] // bs, *m.shape[1:])[idx] for m in attn_dict[ts][layer]]
                tl, pl = grounding_loss_by_layer(masks, attrs, res, maps)
                token_loss = token_loss + tl
                pixel_loss = pixel_loss + pl
    return token_loss / bs, pixel_loss / bs


def disc_forward(d_unet_sd, ucfg, d_lora, head_w, head_b, latents, null_embed, total_steps):
    """UNet_D(latents, t_last, null) -> NHWC -> Linear(4,1): logits (B, h, w, 1).  gan_sdxl.py:68-83."""
    sched = O.DDPM()
    t_last = sched.set_timesteps(total_steps)[-1]
    eps = O.unet_forward(d_unet_sd, ucfg, latents, t_last, null_embed, d_lora, None)
    return F.linear(eps.permute(0, 2, 3, 1), head_w.reshape(1, 4), head_b.reshape(1))


def gan_g_loss(d_unet_sd, ucfg, d_lora, head_w, head_b, fake_latents, null_embed, total_steps):
    """G side (gan_sdxl.py:52-89): discriminator frozen, target 1."""
    d_lora = {k: v.detach() for k, v in d_lora.items()}
    pred = disc_forward(d_unet_sd, ucfg, d_lora, head_w.detach(), head_b.detach(), fake_latents, null_embed,
                        total_steps)
    return F.binary_cross_entropy_with_logits(pred, torch.ones_like(pred))


def gan_d_loss(d_unet_sd, ucfg, d_lora, head_w, head_b, fake_latents, real_latents, null_embed, total_steps):
    """D side (gan_sdxl.py:92-132): batch [fake.detach(); real], targets [0; 1]."""
    x = torch.cat([fake_latents.detach(), real_latents])
    pred = disc_forward(d_unet_sd, ucfg, d_lora, head_w, head_b, x, torch.cat([null_embed, null_embed]), total_steps)
    target = torch.ones_like(pred)
    target[: target.shape[0] // 2] = 0
    return F.binary_cross_entropy_with_logits(pred, target)
