"""ORACLE (test infrastructure, never imported by the product): CPU fp32 restatement of the Stable-Diffusion part of
the CoMat hot path in plain PyTorch ops, NCHW like the reference stack.

PARITY UNPINNED FOR THE UNET'S LAYER ARITHMETIC (everything else in this file is pinned, see below): the arithmetic of
UNet2DConditionModel, AutoencoderKL.decode, DDPMScheduler, LoRALinearLayer and Attention lives in `diffusers>=0.22.1`
(requirements.txt:6), which is neither vendored under /root/reference nor installable here, and no reference test pins
its outputs.  The restatement follows the published architecture and is anchored on the
reference's own call sites:
  - UNet call + CFG + scheduler step: TrainableSDPipeline.py:132-167
  - attention math as patched by the reference (naive softmax(scale QK^T) V, no mask, probs exposed to a
    controller when they require grad): attn_utils/tc_attn_utils.py:104-161
  - LoRA on to_q/to_k/to_v/to_out[0] of every attention, scale 1: training_utils/pipeline.py:84-115
  - VAE decode of latents / scaling_factor, then /2 + 0.5: TrainableSDPipeline.py:219-223
The sampler LOOP (gradient gates, detach rule, CFG, latents chain, image / 2 + 0.5) is pinned by the reference's own
`TrainableSDPipeline.forward` executed on stand-in UNet / VAE / scheduler objects (tests/golden/sampler_loop.npz,
tests/test_oracle.py::test_sampler_loop_matches_reference).
The scheduler constants are pinned by the known answers of SURVEY.md §8(c) (tests/test_oracle.py); the layer set,
state-dict names and shapes this file consumes are pinned by public totals (SD1.5 UNet 859,520,964 parameters in 686
tensors, SDXL UNet 2,567,463,684 in 1,680, VAE decoder 49,490,179: tests/test_architectures.py).  The VAE decoder's
arithmetic and wiring are pinned by an independent implementation of the same architecture (the latent-diffusion decoder
inside transformers' Janus VQ-VAE: tests/test_oracle.py::test_vae_decoder_matches_a_third_party_ldm_decoder); the
arithmetic of the UNet's layers is what remains unpinned.
Weights arrive as a flat dict with diffusers state-dict names (conv weights OIHW, linear weights [out, in]).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------------------------
# configs
# ----------------------------------------------------------------------------------------------------------------
@dataclass
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: tuple = (320, 640, 1280, 1280)
    down_attn: tuple = (True, True, True, False)   # CrossAttnDownBlock2D x3 + DownBlock2D
    layers_per_block: int = 2
    num_heads: int = 8                              # SD1.5: `attention_head_dim: 8` is used as the head COUNT
    cross_attention_dim: int = 768
    norm_groups: int = 32
    lora_rank: int = 128
    heads_per_level: tuple = ()          # SDXL: (5, 10, 20)
    transformer_layers: tuple = ()       # SDXL: (1, 2, 10)
    linear_projection: bool = False      # SDXL: proj_in / proj_out are nn.Linear on tokens
    addition_embed: bool = False         # SDXL: addition_embed_type "text_time"
    addition_time_embed_dim: int = 256
    pooled_dim: int = 1280

    @property
    def time_embed_dim(self):
        return self.block_out_channels[0] * 4

    @property
    def up_attn(self):
        return tuple(reversed(self.down_attn))

    def heads(self, level):
        return self.heads_per_level[level] if self.heads_per_level else self.num_heads

    def depth(self, level):
        return self.transformer_layers[level] if self.transformer_layers else 1


@dataclass
class VAEConfig:
    latent_channels: int = 4
    out_channels: int = 3
    block_out_channels: tuple = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_groups: int = 32
    scaling_factor: float = 0.18215


SD15_UNET = UNetConfig()
SD15_VAE = VAEConfig()
TINY_UNET = UNetConfig(block_out_channels=(32, 64, 64), down_attn=(True, True, False), layers_per_block=1,
                       num_heads=2, cross_attention_dim=24, norm_groups=8, lora_rank=4)
TINY_SDXL_UNET = UNetConfig(block_out_channels=(32, 64, 64), down_attn=(False, True, True), layers_per_block=1,
                            heads_per_level=(2, 2, 4), transformer_layers=(1, 2, 3), cross_attention_dim=24,
                            norm_groups=8, lora_rank=4, linear_projection=True, addition_embed=True,
                            addition_time_embed_dim=8, pooled_dim=16)
TINY_VAE = VAEConfig(block_out_channels=(8, 16, 16, 32), layers_per_block=1, norm_groups=8)


# ----------------------------------------------------------------------------------------------------------------
# DDPM scheduler (SD1.5 scheduler config: scaled_linear betas, steps_offset 1, leading spacing,
# epsilon prediction, fixed_small variance, no clipping)
# ----------------------------------------------------------------------------------------------------------------
class DDPM:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.T = num_train_timesteps
        self.steps_offset = steps_offset
        self.num_inference_steps = None
        self.timesteps = None

    def set_timesteps(self, n):
        self.num_inference_steps = n
        ratio = self.T // n
        ts = [i * ratio + self.steps_offset for i in range(n)][::-1]
        self.timesteps = ts
        return ts

    def coefficients(self, t):
        """x_prev = c_x0 * x0 + c_xt * x + sigma * z with x0 = (x - sqrt(1-abar_t) eps)/sqrt(abar_t).
        Returns (c_x0, c_xt, sigma, sqrt_abar_t, sqrt_1m_abar_t)."""
        prev_t = t - self.T // self.num_inference_steps
        a_t = float(self.alphas_cumprod[t])
        a_prev = float(self.alphas_cumprod[prev_t]) if prev_t >= 0 else 1.0
        beta_t, beta_prev = 1.0 - a_t, 1.0 - a_prev
        cur_alpha = a_t / a_prev
        cur_beta = 1.0 - cur_alpha
        c_x0 = math.sqrt(a_prev) * cur_beta / beta_t
        c_xt = math.sqrt(cur_alpha) * beta_prev / beta_t
        var = max(beta_prev / beta_t * cur_beta, 1e-20)
        sigma = math.sqrt(var) if t > 0 else 0.0
        return c_x0, c_xt, sigma, math.sqrt(a_t), math.sqrt(beta_t)

    def affine(self, t):
        """x_prev = cx * x + ce * eps + sigma * z (x0 eliminated)."""
        c_x0, c_xt, sigma, sa, sb = self.coefficients(t)
        return c_xt + c_x0 / sa, -c_x0 * sb / sa, sigma

    def step(self, eps, t, x, z):
        c_x0, c_xt, sigma, sa, sb = self.coefficients(t)
        x0 = (x - sb * eps) / sa
        return c_x0 * x0 + c_xt * x + sigma * z


# ----------------------------------------------------------------------------------------------------------------
# UNet
# ----------------------------------------------------------------------------------------------------------------
def timestep_embedding(t, dim, max_period=10000.0):
    """diffusers get_timestep_embedding with flip_sin_to_cos=True, downscale_freq_shift=0: [cos | sin]."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = torch.as_tensor(t, dtype=torch.float32).reshape(-1, 1) * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


# fp8 forward of the generator UNet (BASELINE.json configs[4]; arithmetic: oracle/fp8.py).  Inside `fp8_forward()` the
# frozen product of every block Linear / conv whose contraction length per tap is a multiple of 64 takes the VALUE of the
# product of the per-tensor e4m3-quantised activation and weight, and the GRADIENT of the unquantised product (the
# backward runs in the storage dtype on the saved activations: comat_amd/ops.py).  Embeddings, conv_in and conv_out are
# excluded; LoRA, attention and norms are untouched.
_FP8 = False
_NO_FP8 = ("time_emb", "add_embedding", "conv_in", "conv_out")


class fp8_forward:
    def __init__(self, flag=True):
        self.flag = flag

    def __enter__(self):
        global _FP8
        self.prev, _FP8 = _FP8, self.flag

    def __exit__(self, *exc):
        global _FP8
        _FP8 = self.prev
        return False


def _fp8_value(y, name, k_inner, fn, x, w):
    if not _FP8 or k_inner % 64 or any(t in name for t in _NO_FP8):
        return y
    if _FP8 is not True and not any(t in name for t in _FP8):  # a tuple of name fragments: only those layers
        return y
    from .fp8 import fake_quant
    with torch.no_grad():
        yq = fn(fake_quant(x), fake_quant(w))
    return y + (yq - y).detach()


def _lin(sd, name, x, lora=None):
    w, b = sd[name + ".weight"], sd.get(name + ".bias")
    y = _fp8_value(F.linear(x, w, b), name, w.shape[1], lambda xq, wq: F.linear(xq, wq, b), x, w)
    if lora is not None and (name + ".lora.down.weight") in lora:
        y = y + F.linear(F.linear(x, lora[name + ".lora.down.weight"]), lora[name + ".lora.up.weight"])
    return y


def _conv(sd, name, x, stride=1, padding=1):
    w, b = sd[name + ".weight"], sd.get(name + ".bias")
    return _fp8_value(F.conv2d(x, w, b, stride=stride, padding=padding), name, w.shape[1],
                      lambda xq, wq: F.conv2d(xq, wq, b, stride=stride, padding=padding), x, w)


def _gn(sd, name, x, groups, eps):
    return F.group_norm(x, groups, sd[name + ".weight"], sd[name + ".bias"], eps=eps)


def _ln(sd, name, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps=eps)


def attention(sd, name, x, ctx, heads, lora, capture, place, residual=None):
    """The reference's patched Attention.forward (attn_utils/tc_attn_utils.py:104-161) for token inputs."""
    is_cross = ctx is not None
    src = ctx if is_cross else x
    q = _lin(sd, name + ".to_q", x, lora)
    k = _lin(sd, name + ".to_k", src, lora)
    v = _lin(sd, name + ".to_v", src, lora)
    B, N, C = q.shape
    L = k.shape[1]
    d = C // heads

    def split(t, n):
        return t.reshape(B, n, heads, d).permute(0, 2, 1, 3).reshape(B * heads, n, d)
    q, k, v = split(q, N), split(k, L), split(v, L)
    probs = torch.softmax(torch.baddbmm(torch.zeros(B * heads, N, L), q, k.transpose(1, 2), beta=0, alpha=d ** -0.5),
                          dim=-1)
    if capture is not None and probs.requires_grad:
        probs = capture(probs, is_cross, place)  # controller protocol (tc_attn_utils.py:33-42,142-143)
    o = torch.bmm(probs, v).reshape(B, heads, N, d).permute(0, 2, 1, 3).reshape(B, N, C)
    o = _lin(sd, name + ".to_out.0", o, lora)
    return o if residual is None else o + residual


def resnet(sd, name, x, temb, groups, eps=1e-5):
    h = F.silu(_gn(sd, name + ".norm1", x, groups, eps))
    h = _conv(sd, name + ".conv1", h)
    if temb is not None:
        h = h + F.linear(F.silu(temb), sd[name + ".time_emb_proj.weight"], sd[name + ".time_emb_proj.bias"])[:, :, None, None]
    h = F.silu(_gn(sd, name + ".norm2", h, groups, eps))
    h = _conv(sd, name + ".conv2", h)
    if (name + ".conv_shortcut.weight") in sd:
        x = _conv(sd, name + ".conv_shortcut", x, padding=0)
    return x + h


def transformer(sd, name, x, ctx, cfg: UNetConfig, lora, capture, place, level=0):
    B, C, H, W = x.shape
    res = x
    heads = cfg.heads(level)
    h = _gn(sd, name + ".norm", x, cfg.norm_groups, 1e-6)
    if cfg.linear_projection:
        h = _lin(sd, name + ".proj_in", h.permute(0, 2, 3, 1).reshape(B, H * W, C))
    else:
        h = _conv(sd, name + ".proj_in", h, padding=0).permute(0, 2, 3, 1).reshape(B, H * W, C)
    for k in range(cfg.depth(level)):
        blk = f"{name}.transformer_blocks.{k}"
        h = attention(sd, blk + ".attn1", _ln(sd, blk + ".norm1", h), None, heads, lora, capture, place, h)
        h = attention(sd, blk + ".attn2", _ln(sd, blk + ".norm2", h), ctx, heads, lora, capture, place, h)
        f = _lin(sd, blk + ".ff.net.0.proj", _ln(sd, blk + ".norm3", h))
        a, gate = f.chunk(2, dim=-1)
        h = _lin(sd, blk + ".ff.net.2", a * F.gelu(gate)) + h
    if cfg.linear_projection:
        h = _lin(sd, name + ".proj_out", h).reshape(B, H, W, C).permute(0, 3, 1, 2)
    else:
        h = _conv(sd, name + ".proj_out", h.reshape(B, H, W, C).permute(0, 3, 1, 2), padding=0)
    return h + res


def unet_forward(sd, cfg: UNetConfig, sample, t, ctx, lora=None, capture=None, added_cond=None, fp8=False):
    with fp8_forward(fp8):
        return _unet_forward(sd, cfg, sample, t, ctx, lora, capture, added_cond)


def _unet_forward(sd, cfg: UNetConfig, sample, t, ctx, lora=None, capture=None, added_cond=None):
    """sample (B,4,h,w), t int, ctx (B,77,cross_dim) -> eps (B,4,h,w).  `capture(probs, is_cross, place)` is the
    AttentionControl protocol; places are 'down' / 'mid' / 'up'.  SDXL: added_cond = (text_embeds (B,pooled),
    time_ids (B,6)) — `added_cond_kwargs` of TrainableSDPipeline.py:807."""
    B = sample.shape[0]
    g = cfg.norm_groups
    temb = timestep_embedding([t] * B, cfg.block_out_channels[0])
    temb = F.linear(temb, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])
    temb = F.linear(F.silu(temb), sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])
    if cfg.addition_embed:
        text_embeds, time_ids = added_cond
        tid = timestep_embedding(time_ids.reshape(-1), cfg.addition_time_embed_dim).reshape(B, -1)
        add = torch.cat([text_embeds, tid], dim=-1)
        add = F.linear(add, sd["add_embedding.linear_1.weight"], sd["add_embedding.linear_1.bias"])
        temb = temb + F.linear(F.silu(add), sd["add_embedding.linear_2.weight"], sd["add_embedding.linear_2.bias"])
    h = _conv(sd, "conv_in", sample)
    skips = [h]
    nb = len(cfg.block_out_channels)
    for i in range(nb):
        for j in range(cfg.layers_per_block):
            h = resnet(sd, f"down_blocks.{i}.resnets.{j}", h, temb, g)
            if cfg.down_attn[i]:
                h = transformer(sd, f"down_blocks.{i}.attentions.{j}", h, ctx, cfg, lora, capture, "down", i)
            skips.append(h)
        if i < nb - 1:
            h = _conv(sd, f"down_blocks.{i}.downsamplers.0.conv", h, stride=2, padding=1)
            skips.append(h)
    h = resnet(sd, "mid_block.resnets.0", h, temb, g)
    h = transformer(sd, "mid_block.attentions.0", h, ctx, cfg, lora, capture, "mid", nb - 1)
    h = resnet(sd, "mid_block.resnets.1", h, temb, g)
    for i in range(nb):
        for j in range(cfg.layers_per_block + 1):
            h = torch.cat([h, skips.pop()], dim=1)
            h = resnet(sd, f"up_blocks.{i}.resnets.{j}", h, temb, g)
            if cfg.up_attn[i]:
                h = transformer(sd, f"up_blocks.{i}.attentions.{j}", h, ctx, cfg, lora, capture, "up", nb - 1 - i)
        if i < nb - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(sd, f"up_blocks.{i}.upsamplers.0.conv", h)
    h = F.silu(_gn(sd, "conv_norm_out", h, g, 1e-5))
    return _conv(sd, "conv_out", h)


# ----------------------------------------------------------------------------------------------------------------
# VAE decoder (AutoencoderKL.decode)
# ----------------------------------------------------------------------------------------------------------------
def vae_decode(sd, cfg: VAEConfig, z):
    """z = latents / scaling_factor (the caller divides, TrainableSDPipeline.py:220) -> image (B,3,8h,8w)."""
    g = cfg.norm_groups
    h = _conv(sd, "post_quant_conv", z, padding=0)
    h = _conv(sd, "decoder.conv_in", h)
    h = resnet(sd, "decoder.mid_block.resnets.0", h, None, g, eps=1e-6)
    # single-head attention over H*W tokens with group-norm and residual
    B, C, H, W = h.shape
    a = "decoder.mid_block.attentions.0"
    hn = F.group_norm(h.reshape(B, C, H * W), g, sd[a + ".group_norm.weight"], sd[a + ".group_norm.bias"], eps=1e-6)
    tok = hn.transpose(1, 2)
    q, k, v = (_lin(sd, f"{a}.to_{n}", tok) for n in "qkv")
    p = torch.softmax(q @ k.transpose(1, 2) * C ** -0.5, dim=-1)
    o = _lin(sd, a + ".to_out.0", p @ v)
    h = o.transpose(1, 2).reshape(B, C, H, W) + h
    h = resnet(sd, "decoder.mid_block.resnets.1", h, None, g, eps=1e-6)
    nb = len(cfg.block_out_channels)
    for i in range(nb):
        for j in range(cfg.layers_per_block + 1):
            h = resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}", h, None, g, eps=1e-6)
        if i < nb - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(sd, f"decoder.up_blocks.{i}.upsamplers.0.conv", h)
    h = F.silu(_gn(sd, "decoder.conv_norm_out", h, g, 1e-6))
    return _conv(sd, "decoder.conv_out", h)


# ----------------------------------------------------------------------------------------------------------------
# attention store + regrouping (restates attn_utils/tc_attn_utils.py:53-94,198-217; pinned by golden vectors
# generated from the reference file itself, tests/golden/make_attn_golden.py)
# ----------------------------------------------------------------------------------------------------------------
class AttentionStore:
    def __init__(self, train_layer_ls):
        self.train_layer_place = sorted({s.split("_")[0] for s in train_layer_ls})
        self.reset()

    def reset(self):
        self.step_store = {f"{p}_{k}": [] for p in ("down", "mid", "up") for k in ("cross", "self")}

    def __call__(self, probs, is_cross, place):
        if is_cross and place in self.train_layer_place:
            self.step_store[f"{place}_cross"].append(probs.clone())
        return probs

    def maps(self, reses=(64, 32, 16, 8), poses=("down", "mid", "up")):
        out = {}
        for pos in poses:
            for res in reses:
                lst = [m.reshape(-1, res, res, m.shape[-1]) for m in self.step_store[f"{pos}_cross"]
                       if m.shape[1] == res * res]
                if lst:
                    out[f"{pos}_{res}"] = lst
        return out


# ----------------------------------------------------------------------------------------------------------------
# K-of-N differentiable sampler (TrainableSDPipeline.forward, TrainableSDPipeline.py:94-225 and the attrcon branch
# AttrConcenTrainableSDPipeline.py:159-175,239-279) with the trainer's fixed flags: detach_gradient=True,
# bp_on_trained=True, early_exit=False, double_laststep=False, fast_training=False (training_script.py:558-567).
# ----------------------------------------------------------------------------------------------------------------
def sample_with_grad(unet_sd, ucfg, vae_sd, vcfg, lora, ctx_uncond, ctx_cond, latents, noises, total_steps,
                     training_steps, guidance=7.5, attrcon_steps=(), train_layer_ls=(), reses=(64, 32, 16, 8),
                     sdxl_cond=None, fp8_unet=False):
    """fp8_unet: the generator UNet's forward in fp8 (fp8_forward above; per-tensor scales on the joint CFG batch, also on
    attribute-concentration steps).  Returns (image/2+0.5 (B,3,H,W), final latents, attn_dict {str(t): {place_res: [maps]}}).
    sdxl_cond = (neg_pooled, pooled, time_ids (1,6)) selects the SDXL variants (TrainableSDPipeline.py:657-846,
    AttrConcenTrainableSDXLPipeline.py:234-447): the UNet input is ALWAYS detached (`detach_gradient=True`, no
    bp_on_trained exception) and, with return_latents, the decoded image is returned WITHOUT /2+0.5 (:838-840)."""
    sched = DDPM()
    timesteps = sched.set_timesteps(total_steps)
    ctx = torch.cat([ctx_uncond, ctx_cond])
    bs = latents.shape[0]
    attn_dict = {}
    added = None
    if sdxl_cond is not None:
        neg_pooled, pooled, time_ids = sdxl_cond
        added = (torch.cat([neg_pooled, pooled]), torch.cat([time_ids, time_ids]).repeat(bs, 1))
    tmin = min(training_steps) if len(training_steps) else 0
    for i, t in enumerate(timesteps):
        with torch.set_grad_enabled(len(training_steps) == 0 or i > tmin):
            x_in = torch.cat([latents] * 2)
        train = i in training_steps
        with torch.set_grad_enabled(train):
            inp = x_in if (train and sdxl_cond is None) else x_in.detach()
            half = lambda a, lo, hi: None if a is None else (a[0][lo:hi], a[1][lo:hi])
            if train and i in attrcon_steps and fp8_unet:
                # fp8 (no reference counterpart: BASELINE.json configs[4]): per-tensor activation scales are defined on the
                # joint CFG batch, so the capturing step is ONE batched call as well and the cond half of every map is
                # taken afterwards (what the product does in every configuration, comat_amd/pipeline.py)
                store = AttentionStore(train_layer_ls)
                eps2 = unet_forward(unet_sd, ucfg, inp, t, ctx, lora, store, added, fp8=True)
                attn_dict[str(t)] = {k: [m[m.shape[0] // 2:] for m in v] for k, v in store.maps(reses).items()}
            elif train and i in attrcon_steps:
                store = AttentionStore(train_layer_ls)
                e_c = unet_forward(unet_sd, ucfg, inp[bs:], t, ctx[bs:], lora, store, half(added, bs, 2 * bs))
                attn_dict[str(t)] = store.maps(reses)
                e_u = unet_forward(unet_sd, ucfg, inp[:bs], t, ctx[:bs], lora, None, half(added, 0, bs))
                eps2 = torch.cat([e_u, e_c])
            else:
                eps2 = unet_forward(unet_sd, ucfg, inp, t, ctx, lora, None, added, fp8=fp8_unet)
            e_u, e_c = eps2.chunk(2)
            eps = e_u + guidance * (e_c - e_u)
        with torch.set_grad_enabled(len(training_steps) == 0 or i >= tmin):
            latents = sched.step(eps, t, latents, noises[i])
    image = vae_decode(vae_sd, vcfg, latents / vcfg.scaling_factor)
    if sdxl_cond is not None:
        return image, latents, attn_dict
    return image / 2 + 0.5, latents, attn_dict
