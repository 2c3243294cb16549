"""GAN fidelity discriminator: a second SD1.5 UNet (own LoRA) + Linear(4,1) head per latent pixel + BCE-with-logits.

Mirrors `D_sd` / `D_sd.D_sd_pipeline_forward` (training_utils/gan_sdxl.py:7-132; factory
training_utils/gan_sd_model.py:8-14).  G side: discriminator frozen, target 1, gradient flows to the generated
latents.  D side: batch [fake.detach(); real], targets [0; 1], gradient to D's LoRA factors and the head.
The UNet output is already channels-last, so the reference's `permute(0,2,3,1)` disappears and the head + BCE is one
kernel (`comat_disc_head_*`).
"""
from __future__ import annotations

import torch

from . import ops
from .pipeline import DDPMScheduler
from .unet import LoRABank, UNet


class D_sd:
    def __init__(self, unet: UNet, bank: LoRABank, head_w: torch.Tensor, head_b: torch.Tensor):
        self.unet, self.bank = unet, bank
        dev = unet.device
        # head parameters live at the tail of one small flat fp32 buffer so that the optimizer/all-reduce see them
        self.head = torch.cat([head_w.reshape(4).float(), head_b.reshape(1).float()]).to(dev)
        self.head_grad = torch.zeros(5, dtype=torch.float32, device=dev)
        self.w = self.head[:4].requires_grad_(True)
        self.b = self.head[4:].requires_grad_(True)
        self.w.grad, self.b.grad = self.head_grad[:4], self.head_grad[4:]
        self.ori_scheduler = DDPMScheduler()
        self._targets = {}

    def _target(self, bs, side):
        key = (bs, side)
        if key not in self._targets:
            t = torch.ones(bs) if side == "G" else torch.cat([torch.zeros(bs), torch.ones(bs)])
            self._targets[key] = t.to(self.unet.device)
        return self._targets[key]

    def zero_grad(self):
        self.bank.zero_grad()
        self.head_grad.zero_()

    def set_D_sd_pipeline_lora(self, requires_grad=True):
        self.bank.set_requires_grad(requires_grad)
        self.w.requires_grad_(requires_grad)
        self.b.requires_grad_(requires_grad)

    def D_sd_pipeline_forward(self, training_latents, side="G", *, negative_prompt_embeds, num_inference_steps,
                              h, w, real_latents=None):
        """training_latents: fp32 channels-last tokens [bs*h*w, 4]; negative_prompt_embeds (bs, L, C) null embedding;
        real_latents: tokens [bs*h*w, 4] (D side: `batch['latents']`, gan_sdxl.py:46-48)."""
        u = self.unet
        T, dev = u.dtype, u.device
        bs, L, _ = negative_prompt_embeds.shape
        t_last = self.ori_scheduler.set_timesteps(num_inference_steps)[-1]
        null = negative_prompt_embeds.to(dev, torch.float32)
        if side == "G":
            self.set_D_sd_pipeline_lora(False)
            ctx = ops.cast(null.reshape(bs * L, -1).contiguous(), T)
            eps, _ = u(ops.cast_grad(training_latents, T), bs, h, w, t_last, ctx, L)
            return ops.disc_head_loss(eps, self.w, self.b, self._target(bs, "G"), h * w)
        if side == "D":
            self.set_D_sd_pipeline_lora(True)
            with torch.no_grad():
                x = ops.concat_rows(training_latents.detach(), real_latents.to(dev, torch.float32))
                x = ops.cast(x, T)
            ctx = ops.cast(torch.cat([null, null]).reshape(2 * bs * L, -1).contiguous(), T)
            eps, _ = u(x, 2 * bs, h, w, t_last, ctx, L)
            return ops.disc_head_loss(eps, self.w, self.b, self._target(bs, "D"), h * w)
        raise ValueError(side)
