"""K-of-N differentiable DDPM sampler — the MI355X-native counterpart of `TrainableSDPipeline.forward`
(TrainableSDPipeline.py:20-225) and of its attribute-concentration variant
(AttrConcenTrainableSDPipeline.py:38-279), with the same argument names and gradient topology:

  * UNet runs without grad on non-trained steps, and with grad on a NON-detached input on the K trained steps
    (`bp_on_trained=True`, TrainableSDPipeline.py:138-145);
  * the CFG combine + scheduler step runs with grad for every i >= min(training_timesteps) (:163-167), so the loss
    gradient reaches earlier trained steps through the (affine) DDPM chain;
  * on `attrcon_train_steps` the cross-attention maps of the COND half are handed out per timestep
    (`attn_dict[str(t)] = {place_res: [maps]}`, AttrConcenTrainableSDPipeline.py:239-279).  The reference runs the
    cond and uncond halves as two UNet calls there; here it stays one batched call and the cond half of the
    probability tensor is sliced — same arithmetic, half the launches.
Latents stay fp32 channels-last tokens for the whole loop; CFG + DDPM step is one fused kernel.
"""
from __future__ import annotations

import math

import torch

from . import ops
import os

from .unet import GraphedUNetForward, UNet, VAEDecoder, _capturing, regroup_maps


class DDPMScheduler:
    """DDPMScheduler with SD1.5's config (scaled_linear betas, steps_offset 1, leading spacing, epsilon prediction,
    fixed_small variance, clip_sample False) — SURVEY.md A.4; forced by training_utils/pipeline.py:50-59."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.num_train_timesteps = num_train_timesteps
        self.steps_offset = steps_offset
        self.num_inference_steps = None
        self.timesteps = []

    def set_timesteps(self, num_inference_steps):
        self.num_inference_steps = num_inference_steps
        ratio = self.num_train_timesteps // num_inference_steps
        self.timesteps = [i * ratio + self.steps_offset for i in range(num_inference_steps)][::-1]
        return self.timesteps

    def scale_model_input(self, sample, t):
        return sample  # identity for DDPM

    def step_coefficients(self, t):
        """prev_sample = cx * sample + ce * eps + sigma * z (x0 eliminated from DDPMScheduler.step)."""
        prev_t = t - self.num_train_timesteps // self.num_inference_steps
        a_t = float(self.alphas_cumprod[t])
        a_prev = float(self.alphas_cumprod[prev_t]) if prev_t >= 0 else 1.0
        beta_t, beta_prev = 1.0 - a_t, 1.0 - a_prev
        cur_alpha = a_t / a_prev
        cur_beta = 1.0 - cur_alpha
        c_x0 = math.sqrt(a_prev) * cur_beta / beta_t
        c_xt = math.sqrt(cur_alpha) * beta_prev / beta_t
        sigma = math.sqrt(max(beta_prev / beta_t * cur_beta, 1e-20)) if t > 0 else 0.0
        sa, sb = math.sqrt(a_t), math.sqrt(beta_t)
        return c_xt + c_x0 / sa, -c_x0 * sb / sa, sigma


class TrainableSDPipeline:
    is_sdxl = False

    def __init__(self, unet: UNet, vae: VAEDecoder, scheduler: DDPMScheduler | None = None):
        self.unet, self.vae = unet, vae
        self.scheduler = scheduler or DDPMScheduler()
        self.dtype, self.device = unet.dtype, unet.device
        self.attn_dict = {}
        # hipGraph replay for the untrained (no-grad) denoise steps; COMAT_GRAPHS=0 disables it (COMAT_GRAPHS_ATTRCON=0
        # only for forwards that also capture attention maps).  History: an early build raised a hardware exception
        # in the second optimisation step when graphs and attribute-concentration steps were combined; it has not
        # reproduced since the attention-map gather kernel was rewritten (fixed-order, no atomics) — see DESIGN.md.
        # SDXL: the prompt-dependent half of the time embedding is an extra graph input (validated on MI355X in round 2;
        # COMAT_SDXL_GRAPHS=0 disables).  One graph per timestep: capture them up front with prepare_graphs() — a
        # capture costs ~3 eager forwards, and lazily captured graphs only pay off after every timestep has been seen
        # (only a real UNet is replayed from graphs: a stand-in callable - the reference-code fixtures of tests/ drive this
        # class with a small torch "UNet" on the device - has no static launch topology to rely on)
        use_graphs = (torch.device(self.device).type == "cuda" and os.environ.get("COMAT_GRAPHS", "1") != "0"
                      and isinstance(unet, UNet)
                      and (not unet.cfg.addition_embed or os.environ.get("COMAT_SDXL_GRAPHS", "1") != "0"))
        self.graphed = GraphedUNetForward(unet) if use_graphs else None
        # hook of segments.SegmentedStep: runs a TRAINED UNet call (slot = its rank among the trained steps) from a pair
        # of replayable graphs instead of eager launches; None = eager
        self.trained_runner = None
        # the text key / value projections are shared by the denoise steps of one sampler call (UNet.__call__); replayed
        # segments recompute them per call, and so does the eager path when asked to match them bit for bit
        self.share_text_kv = True

    def prepare_graphs(self, batch_size, height, width, L, num_inference_steps):
        """Capture the no-grad UNet forward graph of every timestep up front (before training starts), so that no
        capture — and none of the allocator housekeeping it triggers — happens in the middle of a step."""
        if self.graphed is None:
            return 0
        h, w = height // 8, width // 8
        B = 2 * batch_size
        x = torch.zeros((B * h * w, self.unet.cfg.in_channels), dtype=self.dtype, device=self.device)
        ctx = torch.zeros((B * L, self.unet.cfg.cross_attention_dim), dtype=self.dtype, device=self.device)
        added = None
        if self.unet.cfg.addition_embed:
            added = torch.zeros((B, self.unet.cfg.time_embed_dim), dtype=self.dtype, device=self.device)
        for t in self.scheduler.set_timesteps(num_inference_steps):
            self.graphed(x, B, h, w, int(t), ctx, L, added=added)
        torch.cuda.synchronize()
        return len(self.graphed.graphs)

    def fp8_calibrate(self, prompt_embeds, negative_prompt_embeds, height, width, num_inference_steps, guidance_scale=7.5,
                      latents=None, noises=None, **sdxl_kw):
        """fp8 forward with delayed scaling (ops.set_fp8_scaling('delayed')): the scales of the FIRST optimisation step.  One
        eager no-grad sampler pass over all `num_inference_steps` denoise steps of this prompt in which every quantisation site
        quantises under its own abs-max and records it; the maxima over the pass become the scales (ops.fp8_end_of_step), as
        they will after every later step.  Captured graphs are not touched (they hold the delayed-scaling launches, which read
        the scale words at replay time).  -> True when a calibration ran."""
        if not getattr(self.unet, "fp8", False) or ops.fp8_scaling() != "delayed":
            return False
        graphed, runner = self.graphed, self.trained_runner
        self.graphed = self.trained_runner = None
        try:
            with torch.no_grad(), ops.fp8_calibration():
                self.forward(prompt_embeds, negative_prompt_embeds, height=height, width=width, training_timesteps=(),
                             num_inference_steps=num_inference_steps, guidance_scale=guidance_scale, latents=latents,
                             noises=noises, output_type="latent", **sdxl_kw)
        finally:
            self.graphed, self.trained_runner = graphed, runner
        ops.fp8_end_of_step()
        return True

    def prepare_latents(self, batch_size, height, width, generator=None, latents=None):
        """(bs,4,h/8,w/8) NCHW fp32 -> channels-last tokens [bs*h*w, 4] fp32 on the device."""
        h, w = height // 8, width // 8
        if latents is None:
            latents = torch.randn((batch_size, 4, h, w), generator=generator, dtype=torch.float32)
        latents = latents.to(self.device, torch.float32)
        return ops.nchw_to_tokens(latents), h, w

    def forward(self, prompt_embeds, negative_prompt_embeds, height=512, width=512, training_timesteps=(),
                num_inference_steps=50, guidance_scale=7.5, latents=None, generator=None, noises=None,
                detach_gradient=True, bp_on_trained=True, early_exit=False, double_laststep=False,
                fast_training=False, return_latents=False, attrcon_train_steps=(), train_layer_ls=(),
                attn_reses=(64, 32, 16, 8), output_type="image", pooled_prompt_embeds=None,
                negative_pooled_prompt_embeds=None, add_time_ids=None):
        """prompt_embeds / negative_prompt_embeds: (bs, L, cross_dim) text-encoder outputs (the CLIP text encoder is
        a no-grad preprocessing step outside this path).  Returns image/2+0.5 as (bs,3,H,W) [output_type 'image'] or
        as channels-last tokens ([bs*H*W,3], H, W) ['tokens'], plus the final latents when `return_latents`;
        output_type 'latent' returns the final latents (bs,4,h,w) fp32 without decoding."""
        if early_exit or double_laststep or fast_training or not (detach_gradient and bp_on_trained):
            raise NotImplementedError("only the trainer's flag set (training_script.py:558-567) is supported")
        if guidance_scale <= 1.0:
            raise NotImplementedError("classifier-free guidance is always on in the CoMat trainer")
        bs, L, _ = prompt_embeds.shape
        dev, T = self.device, self.dtype
        ctx = torch.cat([negative_prompt_embeds, prompt_embeds]).to(dev, torch.float32)
        ctx = ops.cast(ctx.reshape(2 * bs * L, -1).contiguous(), T)
        added = None
        if self.is_sdxl:  # added_cond_kwargs of TrainableSDPipeline.py:772-784,807
            if add_time_ids is None:
                add_time_ids = (height, width, 0, 0, height, width)  # original_size + crop_top_left + target_size
            text_embeds = torch.cat([negative_pooled_prompt_embeds, pooled_prompt_embeds]).to(dev, torch.float32)
            # prompt-level constant: computed once here, shared by every denoise step (and a graph input)
            added = self.unet.added_embedding(text_embeds, [list(add_time_ids)] * (2 * bs))
        timesteps = self.scheduler.set_timesteps(num_inference_steps)
        lat, h, w = self.prepare_latents(bs, height, width, generator, latents)
        training_timesteps = list(training_timesteps)
        tmin = min(training_timesteps) if training_timesteps else 0
        places = sorted({s.split("_")[0] for s in train_layer_ls})
        self.attn_dict = {}
        if self.graphed is not None:
            self.graphed.new_sampler_call()  # this call's context / LoRA factors: its text keys / values are projected once
        # text key / value projections shared by the denoise steps of this call (see UNet.__call__)
        kv_cache = {} if (self.share_text_kv and self.trained_runner is None) else None
        wanted = {(s_.split("_")[0], int(s_.split("_")[1])) for s_ in train_layer_ls}
        slot_of = {i: j for j, i in enumerate(sorted(set(training_timesteps)))}
        for i, t in enumerate(timesteps):
            train = i in training_timesteps
            with torch.set_grad_enabled(len(training_timesteps) == 0 or i > tmin):
                x2 = ops.concat_rows(lat, lat)
            with torch.set_grad_enabled(train):
                # SDXL forward detaches the UNet input on every step (`detach_gradient=True`, no bp_on_trained
                # exception: TrainableSDPipeline.py:805-806, AttrConcenTrainableSDXLPipeline.py:393-410)
                xin = x2 if (train and not self.is_sdxl) else x2.detach()
                xin = ops.cast_grad(xin, T)
                cap = places if (train and i in attrcon_train_steps) else ()
                graph_ok = not attrcon_train_steps or os.environ.get("COMAT_GRAPHS_ATTRCON", "1") != "0"
                if (not train and self.graphed is not None and graph_ok
                        and not torch.cuda.is_current_stream_capturing()):
                    eps2, maps = self.graphed(xin, 2 * bs, h, w, int(t), ctx, L, added=added), {}
                elif train and self.trained_runner is not None and not _capturing(dev):
                    eps2, maps = self.trained_runner(slot_of[i], xin, 2 * bs, h, w, int(t), ctx, L, cap, added, wanted)
                else:
                    eps2, maps = self.unet(xin, 2 * bs, h, w, int(t), ctx, L, capture_places=cap, added=added,
                                           kv_cache=kv_cache)
                if cap:
                    cond = {p: [m[bs:] for m in lst] for p, lst in maps.items()}
                    self.attn_dict[str(int(t))] = regroup_maps(cond, reses=attn_reses)
            if noises is not None:
                z = ops.nchw_to_tokens(noises[i].to(dev, torch.float32))
            else:
                z = torch.randn(lat.shape, generator=generator, dtype=torch.float32,
                                device=dev if generator is None else generator.device).to(dev)
            dbg = os.environ.get("COMAT_DEBUG_SYNC")
            if dbg == "1" or (dbg == "train" and train) or (dbg == "nograd" and not train):
                torch.cuda.synchronize()
                print(f"[comat] denoise step {i} (t={int(t)}, train={train}, capture={bool(cap)}) ok", flush=True)
            cx, ce, sigma = self.scheduler.step_coefficients(int(t))
            with torch.set_grad_enabled(len(training_timesteps) == 0 or i >= tmin):
                lat = ops.cfg_ddpm_step(lat, eps2, z, guidance_scale, cx, ce, sigma)
        if output_type == "latent":  # TrainableSDPipeline.py:224-225: the final latents, no decode
            return ops.tokens_to_nchw(lat, bs, h, w)
        if output_type == "latent_tokens":  # the same as channels-last tokens [bs*h*w, 4] fp32 (decode_tokens follows)
            return lat
        img, H, W = self.decode_tokens(lat, bs, h, w, return_latents)
        if output_type == "tokens":
            out = (img, H, W)
        else:
            out = ops.tokens_to_nchw(img, bs, H, W)
        if return_latents:
            return out, (lat if output_type == "tokens" else ops.tokens_to_nchw(lat, bs, h, w))
        return out


    def decode_tokens(self, lat, bs, h, w, return_latents=False):
        """`vae.decode(latents / scaling_factor)` (+ `/2 + 0.5`, TrainableSDPipeline.py:219-223) on channels-last latent
        tokens -> (image tokens [bs*H*W, 3], H, W)"""
        z0 = ops.cast_grad(ops.affine(lat, 1.0 / self.vae.cfg.scaling_factor, 0.0), getattr(self.vae, "dtype", self.dtype))
        img, H, W = self.vae(z0, bs, h, w)
        if not (self.is_sdxl and return_latents):  # SDXL + return_latents returns the raw decode (:838-840)
            img = ops.affine(img, 0.5, 0.5)
        return img, H, W


class TrainableSDXLPipeline(TrainableSDPipeline):
    """SDXL variant (TrainableSDPipeline.py:657-846, AttrConcenTrainableSDXLPipeline.py:234-496): pooled text
    embedding + size/crop ids as additional conditioning, UNet input always detached, raw VAE decode returned with
    `return_latents`.  Use with an SDXL_UNET-style UNetConfig and VAEConfig(scaling_factor=0.13025)."""
    is_sdxl = True
