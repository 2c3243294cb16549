"""SD UNet (UNet2DConditionModel, SD1.5 topology) and the VAE decoder assembled from the HIP operators.

Replaces the third-party model code behind `self.unet(latent_model_input, t, encoder_hidden_states=...)`
(TrainableSDPipeline.py:144-150) and `self.vae.decode(...)` (:220).  MI355X-first choices:
  * activations are channels-last token matrices [B*H*W, C]: 1x1 convs and attention projections are plain GEMMs on
    the same buffer (the reference's view/transpose between NCHW and tokens disappears), 3x3 convs are implicit
    GEMMs, skip concat is a column concat;
  * frozen weights are stored in both orientations (forward + data-gradient) — 288 GB of HBM make the second
    copy free — so fwd and dgrad of every conv/linear share one k-contiguous MFMA kernel;
  * time-embedding adds, biases and residual adds ride in GEMM/conv epilogues; GroupNorm+SiLU is one kernel pair;
  * cross-attention probabilities are written once as [B, heads, N, 77] and handed out as the "captured map"
    (the reference clones them in AttentionStore.forward, attn_utils/tc_attn_utils.py:60-68);
  * no activation checkpointing: all K trained UNet calls keep their activations resident (SD1.5, bs 1: < 40 GB).
Module / parameter names follow the diffusers state dict so that real checkpoints map 1:1.
"""
from __future__ import annotations

import math
import os

import numpy as np
import torch

from . import ops
from .config import UNetConfig, VAEConfig
from .weights import attention_names


def timestep_embedding(t, dim, batch, max_period=10000.0):
    """[cos | sin] sinusoid (flip_sin_to_cos=True, freq_shift=0), computed on the host: t is a host integer."""
    half = dim // 2
    freqs = np.exp(-math.log(max_period) * np.arange(half, dtype=np.float32) / half).astype(np.float32)
    args = np.float32(t) * freqs
    emb = np.concatenate([np.cos(args), np.sin(args)]).astype(np.float32)
    return torch.from_numpy(np.tile(emb[None], (batch, 1)))


# projections of one Attention that read the same input: (group key, members)
ATTN1_GROUPS = (("qkv", ("to_q", "to_k", "to_v")), ("out", ("to_out.0",)))           # self-attention
ATTN2_GROUPS = (("q", ("to_q",)), ("kv", ("to_k", "to_v")), ("out", ("to_out.0",)))  # cross-attention


def attention_groups(path):
    return ATTN1_GROUPS if path.endswith("attn1") else ATTN2_GROUPS


def _capturing(device):
    return torch.device(device).type == "cuda" and torch.cuda.is_current_stream_capturing()


class LoRABank(ops.LoRAStore):
    """The LoRA factors of one UNet (training_utils/pipeline.py:84-144: rank-r LoRALinearLayers on to_q / to_k / to_v
    / to_out.0 of every Attention; the same parameter set, keyed by the same names) in the flat-buffer store of
    ops.LoRAStore.  Projections that read the same activation form one group (see ops.LoRAGroup); `self.group[(path,
    key)]` with key in {"qkv", "out"} (attn1) or {"q", "kv", "out"} (attn2)."""

    def __init__(self, cfg: UNetConfig, lora_sd: dict, dtype, device):
        spec, keys = [], []
        for path, _, _, _ in attention_names(cfg):
            for key, projs in attention_groups(path):
                spec.append([(f"{path}.{p}.lora.down.weight", f"{path}.{p}.lora.up.weight",
                              lora_sd[f"{path}.{p}.lora.down.weight"], lora_sd[f"{path}.{p}.lora.up.weight"])
                             for p in projs])
                keys.append((path, key))
        super().__init__(spec, dtype, device)
        self.group = dict(zip(keys, self.groups))


class _Mods:
    """Weight containers built from a diffusers-named state dict."""

    NO_FP8 = ("time_emb", "add_embedding", "conv_in", "conv_out")  # embeddings and the first / last conv stay exact

    def __init__(self, sd, dtype, device, fp8=False):
        """fp8: False, True (every eligible block layer) or a tuple of name fragments (only layers whose state-dict name
        contains one of them: per-category parity tests)"""
        self.sd, self.dtype, self.device = sd, dtype, device
        self.fp8, self.made, self.groups = fp8, [], []

    def _tag(self, obj, name):
        obj.allow_fp8 = (bool(self.fp8) and not any(s in name for s in self.NO_FP8)
                         and (self.fp8 is True or any(s in name for s in self.fp8)))
        self.made.append(obj)
        return obj

    def lin(self, name):
        return self._tag(ops.FrozenLinear(self.sd[name + ".weight"], self.sd.get(name + ".bias"), self.dtype, self.device),
                         name)

    def geglu_lin(self, name):
        """`ff.net.0.proj` with the GEGLU behind it (ops.FrozenGegluLinear: rows interleaved for the fused epilogue)"""
        return self._tag(ops.FrozenGegluLinear(self.sd[name + ".weight"], self.sd.get(name + ".bias"), self.dtype, self.device),
                         name)

    def lin_group(self, names):
        """projections that read the same input: one co-allocated weight buffer when their shapes agree"""
        ws = [self.sd[n + ".weight"] for n in names]
        if len(names) > 1 and all(tuple(w.shape) == tuple(ws[0].shape) for w in ws):
            grp = ops.frozen_linear_group(ws, [self.sd.get(n + ".bias") for n in names], self.dtype, self.device)
            grp = [self._tag(g, n) for g, n in zip(grp, names)]
            self.groups.append(grp)
            return grp
        return [self.lin(n) for n in names]

    def conv(self, name, stride=1, pad=1):
        return self._tag(ops.FrozenConv(self.sd[name + ".weight"], self.sd.get(name + ".bias"), self.dtype, self.device,
                                        stride=stride, pad=pad), name)

    def conv1x1(self, name):
        """1x1 conv (weight [Co,Ci,1,1]) or nn.Linear (weight [Co,Ci]) — the same GEMM on channels-last tokens."""
        w = self.sd[name + ".weight"]
        return self._tag(ops.FrozenLinear(w.reshape(w.shape[0], w.shape[1]), self.sd.get(name + ".bias"), self.dtype,
                                          self.device), name)

    def norm(self, name):
        f = lambda t: t.to(device=self.device, dtype=torch.float32).contiguous()
        return f(self.sd[name + ".weight"]), f(self.sd[name + ".bias"])


class ResBlock:
    def __init__(self, m: _Mods, name, groups, eps, has_temb=True):
        self.n1, self.n2 = m.norm(name + ".norm1"), m.norm(name + ".norm2")
        self.c1, self.c2 = m.conv(name + ".conv1"), m.conv(name + ".conv2")
        self.temb = m.lin(name + ".time_emb_proj") if has_temb else None
        self.short = m.conv1x1(name + ".conv_shortcut") if (name + ".conv_shortcut.weight") in m.sd else None
        self.groups, self.eps = groups, eps

    def __call__(self, x, B, H, W, temb_act):
        HW = H * W
        # x feeds the norm AND the shortcut: the fork hands both gradients to one GroupNorm-backward launch
        h, x = ops.group_norm_fork(x, *self.n1, B, HW, G=self.groups, eps=self.eps, silu=True, fp8_for=self.c1)
        b2 = None
        if self.temb is not None:
            b2 = temb_act.get(id(self))
            if b2 is None:
                with torch.no_grad():  # depends on t and frozen weights only: computed once per timestep value
                    b2 = temb_act[id(self)] = ops.linear(temb_act["silu_temb"], self.temb, out_dtype=torch.float32)
        h = ops.conv2d(h, self.c1, B, H, W, bias2=b2)
        h = ops.group_norm(h, *self.n2, B, HW, G=self.groups, eps=self.eps, silu=True, fp8_for=self.c2)
        sc = x if self.short is None else ops.linear(x, self.short)
        return ops.conv2d(h, self.c2, B, H, W, residual=sc)


class CrossAttnBlock:
    """Transformer2DModel: GroupNorm, proj_in, `depth` BasicTransformerBlocks (self-attn, cross-attn to the text
    context, GEGLU FFN), proj_out, residual.  SD1.5: depth 1; SDXL: 1 / 2 / 10."""

    def __init__(self, m: _Mods, name, cfg: UNetConfig, lora: LoRABank | None, level=0):
        self.cfg = cfg
        self.heads = cfg.heads(level)
        self.norm = m.norm(name + ".norm")
        self.proj_in, self.proj_out = m.conv1x1(name + ".proj_in"), m.conv1x1(name + ".proj_out")
        self.layers = []
        for k in range(cfg.depth(level)):
            b = f"{name}.transformer_blocks.{k}"
            att = {}
            for a in ("attn1", "attn2"):
                for key, projs in attention_groups(a):
                    att[(a, key)] = (m.lin_group([f"{b}.{a}.{p}" for p in projs]),
                                     lora.group[(f"{b}.{a}", key)] if lora is not None else None)
            self.layers.append(dict(ln=[m.norm(f"{b}.norm{i}") for i in (1, 2, 3)], att=att,
                                    ff1=m.geglu_lin(f"{b}.ff.net.0.proj"), ff2=m.lin(f"{b}.ff.net.2")))

    def _self_attn(self, att, x, B, N):
        q, k, v = ops.lora_group_linear(x, *att[("attn1", "qkv")])
        return ops.attention(q, k, v, B, N, N, self.heads, q.shape[1] // self.heads, need_probs=False,
                             fp8_for=att[("attn1", "out")][0][0])[0]

    def text_kv(self, att, ctx, kv_cache):
        """(k, v) = the text key / value projections of one layer.  They depend on the context and the LoRA factors only:
        within one sampler call (one optimisation step) every denoise step shares them through `kv_cache`, and autograd
        sums their gradients before ONE backward through the projection"""
        if kv_cache is None:
            return ops.lora_group_linear(ctx, *att[("attn2", "kv")])
        key = (id(att), ctx.data_ptr(), tuple(ctx.shape), torch.is_grad_enabled())
        kv = kv_cache.get(key)
        if kv is None:
            kv = kv_cache[key] = ops.lora_group_linear(ctx, *att[("attn2", "kv")])
        return kv

    def _cross_attn(self, att, x, ctx, B, N, L, need_probs, kv_cache):
        (q,) = ops.lora_group_linear(x, *att[("attn2", "q")])
        k, v = self.text_kv(att, ctx, kv_cache)
        return ops.attention(q, k, v, B, N, L, self.heads, q.shape[1] // self.heads, need_probs=need_probs,
                             fp8_for=att[("attn2", "out")][0][0])

    def __call__(self, x, B, H, W, ctx, L, want_probs, kv_cache=None):
        """returns (tokens, [cross-attention probabilities of every transformer layer] or None)"""
        N = H * W
        h, x = ops.group_norm_fork(x, *self.norm, B, N, G=self.cfg.norm_groups, eps=1e-6, silu=False, fp8_for=self.proj_in)
        h = ops.linear(h, self.proj_in)
        probs_all = [] if want_probs else None
        for Lr in self.layers:
            att, ln = Lr["att"], Lr["ln"]
            # (norm, residual alias): one LayerNorm-backward launch per fork; fp8_for: the layer the norm feeds (ops.group_norm)
            y, h = ops.layer_norm_fork(h, *ln[0], fp8_for=att[("attn1", "qkv")][0][0])
            o = self._self_attn(att, y, B, N)
            h = ops.lora_group_linear(o, *att[("attn1", "out")], residual=h)[0]
            y, h = ops.layer_norm_fork(h, *ln[1], fp8_for=att[("attn2", "q")][0][0])
            o, probs = self._cross_attn(att, y, ctx, B, N, L, want_probs, kv_cache)
            h = ops.lora_group_linear(o, *att[("attn2", "out")], residual=h)[0]
            y, h = ops.layer_norm_fork(h, *ln[2], fp8_for=Lr["ff1"])
            h = ops.geglu_feed_forward(y, Lr["ff1"], Lr["ff2"], residual=h)  # fwd: 2 launches; bwd: GEGLU' in ff2's dgrad epilogue
            if want_probs:
                probs_all.append(probs)
        return ops.linear(h, self.proj_out, residual=x), probs_all


class UNet:
    def __init__(self, cfg: UNetConfig, sd: dict, dtype=torch.bfloat16, device="cuda", lora: LoRABank | None = None,
                 fp8_forward=False):
        """fp8_forward: BASELINE.json configs[4] - the forward products of the frozen block layers (resnet convs,
        down / upsamplers, proj_in / proj_out, attention projections, feed-forward) run on the fp8 MFMA with per-tensor
        scales (ops.fp8_forward); embeddings, conv_in / conv_out, LoRA, attention and every backward product do not.
        A tuple of name fragments restricts it to the layers whose state-dict name contains one of them."""
        self.cfg, self.dtype, self.device, self.lora = cfg, dtype, device, lora
        self.fp8 = bool(fp8_forward)
        m = _Mods(sd, dtype, device, fp8=fp8_forward if isinstance(fp8_forward, tuple) else self.fp8)
        g = cfg.norm_groups
        self.t1, self.t2 = m.lin("time_embedding.linear_1"), m.lin("time_embedding.linear_2")
        if cfg.addition_embed:
            self.a1, self.a2 = m.lin("add_embedding.linear_1"), m.lin("add_embedding.linear_2")
        self.conv_in = m.conv("conv_in")
        nb = len(cfg.block_out_channels)
        self.down, self.up = [], []
        for i in range(nb):
            res = [ResBlock(m, f"down_blocks.{i}.resnets.{j}", g, 1e-5) for j in range(cfg.layers_per_block)]
            att = [CrossAttnBlock(m, f"down_blocks.{i}.attentions.{j}", cfg, lora, i)
                   for j in range(cfg.layers_per_block)] if cfg.down_attn[i] else None
            ds = m.conv(f"down_blocks.{i}.downsamplers.0.conv", stride=2, pad=1) if i < nb - 1 else None
            self.down.append((res, att, ds))
        self.mid = (ResBlock(m, "mid_block.resnets.0", g, 1e-5),
                    CrossAttnBlock(m, "mid_block.attentions.0", cfg, lora, nb - 1),
                    ResBlock(m, "mid_block.resnets.1", g, 1e-5))
        for i in range(nb):
            res = [ResBlock(m, f"up_blocks.{i}.resnets.{j}", g, 1e-5) for j in range(cfg.layers_per_block + 1)]
            att = [CrossAttnBlock(m, f"up_blocks.{i}.attentions.{j}", cfg, lora, nb - 1 - i)
                   for j in range(cfg.layers_per_block + 1)] if cfg.up_attn[i] else None
            us = m.conv(f"up_blocks.{i}.upsamplers.0.conv") if i < nb - 1 else None
            self.up.append((res, att, us))
        self.norm_out = m.norm("conv_norm_out")
        self.conv_out = m.conv("conv_out")
        self._temb_cache = {}
        self._te_cache = {}
        if self.fp8:  # quantise the frozen weights now (once), not inside the first step or a graph capture
            ops.fp8_state(device)  # ... and give every activation site its scale / running-maximum words (delayed scaling)
            for grp in m.groups:  # q / k / v (k / v) of one attention: their bytes in one buffer (one batched fp8 product)
                if all(ops.fp8_eligible(o, o.in_features) for o in grp):
                    ops.fp8_weight_group(grp)
            for o in m.made:
                if ops.fp8_eligible(o, o.cin if isinstance(o, ops.FrozenConv) else o.in_features):
                    ops.fp8_weight(o)
                    ops._fp8_site(o, device)

    def added_embedding(self, text_embeds, time_ids):
        """SDXL `text_time` conditioning (TrainableSDPipeline.py:772-784,807): add_embedding([pooled text |
        sinusoid(time_ids)]) -> [B, time_embed_dim] in the compute dtype.  It depends on the prompt only, not on the
        timestep: the sampler computes it ONCE per call and hands it to every UNet call (and to the captured graphs
        as an input buffer).  text_embeds: [B, pooled] tensor; time_ids: [B, 6] host values."""
        cfg = self.cfg
        B = text_embeds.shape[0]
        with torch.no_grad():
            tid = np.concatenate([timestep_embedding(float(v), cfg.addition_time_embed_dim, 1).numpy()
                                  for v in np.asarray(time_ids, dtype=np.float32).reshape(-1)], axis=1)
            tid = torch.from_numpy(tid.reshape(B, -1)).to(self.device)
            add = ops.concat_cols(ops.cast(text_embeds.to(self.device), self.dtype), ops.cast(tid, self.dtype))
            return ops.linear(ops.linear(add, self.a1, act=ops.ACT_SILU), self.a2)

    def time_sinusoid(self, t, B):
        """[B, C0] fp32 sinusoid of timestep t on the device (memoised: at most one host->device copy per timestep)"""
        key = ("sin", int(t), B)
        te = self._te_cache.get(key)
        if te is None:
            te = self._te_cache[key] = timestep_embedding(t, self.cfg.block_out_channels[0], B).to(self.device)
        return te

    def _time_embedding(self, t, B, added):
        """{"silu_temb": SiLU(emb)} plus, lazily, each ResBlock's projection of it.  SD1.5: depends on (t, batch)
        and frozen weights only -> memoised.  SDXL: emb = temb(t) + added_embedding(prompt): the timestep half is
        memoised, the sum and the ResBlock projections are recomputed per call (they depend on the prompt)."""
        cfg = self.cfg
        if torch.is_tensor(t):
            # the sinusoid itself as a device tensor (time_sinusoid): nothing is memoised and nothing comes from the
            # host, so ONE captured graph serves every timestep (comat_amd/segments.py).  Same arithmetic as below.
            with torch.no_grad():
                te = ops.linear(ops.cast(t, self.dtype), self.t1, act=ops.ACT_SILU)
                if not cfg.addition_embed:
                    return {"silu_temb": ops.linear(te, self.t2, act=ops.ACT_SILU)}
                aug = added if torch.is_tensor(added) else self.added_embedding(*added)
                return {"silu_temb": ops.silu(ops.linear(te, self.t2, residual=aug))}
        key = (int(t), B)
        if not cfg.addition_embed:
            temb_act = self._temb_cache.get(key)
            if temb_act is not None:
                return temb_act
        with torch.no_grad():
            te = self._te_cache.get(key) if cfg.addition_embed else None
            if te is None:
                te = timestep_embedding(t, cfg.block_out_channels[0], B).to(self.device)
                te = ops.linear(ops.cast(te, self.dtype), self.t1, act=ops.ACT_SILU)
                if cfg.addition_embed and len(self._te_cache) < 128 and not _capturing(self.device):
                    self._te_cache[key] = te
            if not cfg.addition_embed:
                temb_act = {"silu_temb": ops.linear(te, self.t2, act=ops.ACT_SILU)}
                if len(self._temb_cache) < 128:
                    self._temb_cache[key] = temb_act
                return temb_act
            aug = added if torch.is_tensor(added) else self.added_embedding(*added)
            emb = ops.linear(te, self.t2, residual=aug)
            return {"silu_temb": ops.silu(emb)}

    def cross_attention_blocks(self):
        """every CrossAttnBlock in call order (down, mid, up)"""
        out = []
        for _, att, _ in self.down:
            out += att or []
        out.append(self.mid[1])
        for _, att, _ in self.up:
            out += att or []
        return out

    def project_text_kv(self, ctx, kv_cache):
        """fill `kv_cache` with the cross-attention key / value projections of `ctx` for every transformer layer (what
        the layers would compute at their first use): the no-grad graphs read them instead of projecting per call"""
        for blk in self.cross_attention_blocks():
            for Lr in blk.layers:
                blk.text_kv(Lr["att"], ctx, kv_cache)

    def __call__(self, x, B, H, W, t, ctx, L, capture_places=(), added=None, kv_cache=None):
        """x: [B*H*W, 4] tokens (compute dtype), ctx: [B*L, cross_dim]; t: host integer timestep, or its sinusoid as a
        device tensor (time_sinusoid) when the call is being captured for replay at any timestep.  Returns (eps tokens [B*H*W, 4],
        maps {place: [probs [B, heads, N, L], ...]}) — maps only for `capture_places` ⊆ {'down','mid','up'}.
        SDXL: added = (text_embeds [B, pooled], time_ids [B, 6]), or the precomputed `added_embedding(...)` tensor.
        kv_cache: a dict owned by the caller for ONE sampler invocation (LoRA factors and `ctx` must not change while
        it lives): the cross-attention key / value projections of `ctx` are computed once and shared by its calls."""
        with ops.fp8_forward(self.fp8):
            return self._forward(x, B, H, W, t, ctx, L, capture_places, added, kv_cache)

    def _forward(self, x, B, H, W, t, ctx, L, capture_places, added, kv_cache):
        cfg = self.cfg
        temb_act = self._time_embedding(t, B, added)
        maps = {p: [] for p in capture_places}
        h = ops.conv2d(x, self.conv_in, B, H, W)
        skips = [h]
        hh, ww = H, W
        for res, att, ds in self.down:
            for j, r in enumerate(res):
                h = r(h, B, hh, ww, temb_act)
                if att is not None:
                    h, p = att[j](h, B, hh, ww, ctx, L, "down" in maps, kv_cache)
                    if p is not None:
                        maps["down"].extend(p)
                skips.append(h)
            if ds is not None:
                h = ops.conv2d(h, ds, B, hh, ww)
                hh, ww = ops.conv_out_hw(ds, hh, ww)
                skips.append(h)
        h = self.mid[0](h, B, hh, ww, temb_act)
        h, p = self.mid[1](h, B, hh, ww, ctx, L, "mid" in maps, kv_cache)
        if p is not None:
            maps["mid"].extend(p)
        h = self.mid[2](h, B, hh, ww, temb_act)
        for res, att, us in self.up:
            for j, r in enumerate(res):
                h = ops.concat_cols(h, skips.pop())
                h = r(h, B, hh, ww, temb_act)
                if att is not None:
                    h, p = att[j](h, B, hh, ww, ctx, L, "up" in maps, kv_cache)
                    if p is not None:
                        maps["up"].extend(p)
            if us is not None:
                h = ops.conv2d(h, us, B, hh, ww, ups=2)
                hh, ww = hh * 2, ww * 2
        h = ops.group_norm(h, *self.norm_out, B, hh * ww, G=cfg.norm_groups, eps=1e-5, silu=True)
        return ops.conv2d(h, self.conv_out, B, hh, ww), maps


class GraphedUNetForward:
    """hipGraph replay of the NO-GRAD UNet forward (the N-K untrained denoise steps of a CoMat step are ~700 small
    launches each and host-bound when issued one by one).  One graph per (t, batch, H, W, L): the time-embedding
    projections are baked per timestep; latents and text context go through static input buffers; LoRA factors are
    read from the bank's flat compute copy at replay time, so optimizer updates are seen without re-capture.
    The cross-attention key / value projections of the text context are the same for every denoise step of a sampler call
    (same context, same LoRA factors): they live in ONE more graph that is replayed once per sampler call
    (`new_sampler_call()` marks the boundary: CALL IT whenever the text context changes - the pipeline does at the start
    of every `forward`; an update of the LoRA factors is noticed by itself) and writes fixed-address tensors the
    per-timestep graphs read - 2 launches per cross-attention layer and step less (SD1.5: 32 of ~700, SDXL: 140 of ~2 000)."""

    def __init__(self, unet: "UNet"):
        self.unet = unet
        self.graphs = {}
        self.timing = None
        self.pool = None  # one memory pool for all timesteps: the graphs never run concurrently, only `out` stays alive
        self.static = {}  # (B, H, W, L) -> static inputs + the text K/V graph
        self.share_text_kv = os.environ.get("COMAT_NOGRAD_TEXT_KV", "1") != "0"

    def new_sampler_call(self):
        """the text context (or the LoRA factors) may have changed: the next replay refreshes the key / value tensors"""
        for st in self.static.values():
            st["kv_fresh"] = False

    def _capture_kwargs(self):
        import torch.distributed as dist
        return {"capture_error_mode": "thread_local"} if dist.is_available() and dist.is_initialized() else {}

    def _static(self, x, B, H, W, ctx, L, added):
        key = (B, H, W, L)
        st = self.static.get(key)
        if st is None:
            u = self.unet
            sx, sc = torch.empty_like(x), torch.empty_like(ctx)
            sa = None if added is None else torch.empty_like(added)
            sx.copy_(x)
            sc.copy_(ctx)
            if sa is not None:
                sa.copy_(added)
            st = self.static[key] = dict(sx=sx, sc=sc, sa=sa, kv=None, kv_graph=None, kv_fresh=False)
            if self.share_text_kv:
                with torch.no_grad():
                    if u.lora is not None:
                        u.lora.ensure_compute_copy()
                    u.project_text_kv(sc, {})  # eager warm-up (workspaces, LoRA copies)
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    kv = {}
                    with ops.graph_capture(g, pool=self.pool, stream=ops.capture_stream(u.device), **self._capture_kwargs()):
                        u.project_text_kv(sc, kv)
                    if self.pool is None:
                        self.pool = g.pool()
                st["kv"], st["kv_graph"] = kv, g
        return st

    def __call__(self, x, B, H, W, t, ctx, L, added=None):
        """`added`: SDXL only — the precomputed UNet.added_embedding(...) tensor (a graph input like x and ctx)."""
        st = self._static(x, B, H, W, ctx, L, added)
        sx, sc, sa = st["sx"], st["sc"], st["sa"]
        key = (int(t), B, H, W, L)
        ent = self.graphs.get(key)
        if ent is None:
            u = self.unet
            with torch.no_grad():
                u(sx, B, H, W, t, sc, L, added=sa, kv_cache=st["kv"])  # eager warm-up: temb memo, LoRA compute copy
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                # the package's capture stream: its workspaces exist (zeroed) before any capture begins
                with ops.graph_capture(g, pool=self.pool, stream=ops.capture_stream(u.device), **self._capture_kwargs()):
                    out, _ = u(sx, B, H, W, t, sc, L, added=sa, kv_cache=st["kv"])
                if self.pool is None:
                    self.pool = g.pool()
            ent = self.graphs[key] = (g, out)
        g, out = ent
        k = ops.kernels()
        k.unary(ops.UN_COPY, x, sx, x.numel())
        if sa is not None:
            k.unary(ops.UN_COPY, added, sa, added.numel())
        if self.unet.lora is not None:
            self.unet.lora.ensure_compute_copy()
            if st.get("lora_epoch") != getattr(self.unet.lora, "epoch", 0):  # the factors changed: so do the projections
                st["kv_fresh"], st["lora_epoch"] = False, getattr(self.unet.lora, "epoch", 0)
        if not st["kv_fresh"] or st["kv_graph"] is None:
            k.unary(ops.UN_COPY, ctx, sc, ctx.numel())
            if st["kv_graph"] is not None:
                st["kv_graph"].replay()
            st["kv_fresh"] = True
        if self.timing is None:
            g.replay()
        else:  # diagnostics (bench.py): HIP events around the replay
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            g.replay()
            e.record()
            self.timing.append((s, e))
        return out


def regroup_maps(maps: dict, reses=(64, 32, 16, 8), poses=("down", "mid", "up")):
    """Same key/shape schema as the reference's get_cross_attn_map_from_unet (attn_utils/tc_attn_utils.py:198-217):
    {f"{pos}_{res}": [Tensor(B*heads, res, res, L), ...]} — views of the stored probabilities, no copy."""
    out = {}
    for pos in poses:
        for res in reses:
            lst = [p.reshape(-1, res, res, p.shape[-1]) for p in maps.get(pos, []) if p.shape[2] == res * res]
            if lst:
                out[f"{pos}_{res}"] = lst
    return out


class VAEDecoder:
    """AutoencoderKL.decode (post_quant_conv + decoder); frozen, gradient flows to the latents only."""

    def __init__(self, cfg: VAEConfig, sd: dict, dtype=torch.bfloat16, device="cuda"):
        self.cfg, self.dtype, self.device = cfg, dtype, device
        m = _Mods(sd, dtype, device)
        g = cfg.norm_groups
        self.pq = m.conv1x1("post_quant_conv")
        self.conv_in = m.conv("decoder.conv_in")
        self.mid0 = ResBlock(m, "decoder.mid_block.resnets.0", g, 1e-6, has_temb=False)
        a = "decoder.mid_block.attentions.0"
        self.a_norm = m.norm(a + ".group_norm")
        self.a_q, self.a_k, self.a_v, self.a_o = (m.lin(f"{a}.{n}") for n in ("to_q", "to_k", "to_v", "to_out.0"))
        self.mid1 = ResBlock(m, "decoder.mid_block.resnets.1", g, 1e-6, has_temb=False)
        self.up = []
        nb = len(cfg.block_out_channels)
        for i in range(nb):
            res = [ResBlock(m, f"decoder.up_blocks.{i}.resnets.{j}", g, 1e-6, has_temb=False)
                   for j in range(cfg.layers_per_block + 1)]
            us = m.conv(f"decoder.up_blocks.{i}.upsamplers.0.conv") if i < nb - 1 else None
            self.up.append((res, us))
        self.norm_out = m.norm("decoder.conv_norm_out")
        self.conv_out = m.conv("decoder.conv_out")

    def __call__(self, z, B, H, W):
        """z: [B*H*W, 4] latent tokens already divided by the scaling factor -> ([B*H'*W', 3], H', W')."""
        g = self.cfg.norm_groups
        h = ops.linear(z, self.pq)
        h = ops.conv2d(h, self.conv_in, B, H, W)
        h = self.mid0(h, B, H, W, None)
        N = H * W
        hn, h = ops.group_norm_fork(h, *self.a_norm, B, N, G=g, eps=1e-6, silu=False)
        q, k, v = ops.linear(hn, self.a_q), ops.linear(hn, self.a_k), ops.linear(hn, self.a_v)
        o, _ = ops.attention(q, k, v, B, N, N, 1, q.shape[1], need_probs=False)
        h = ops.linear(o, self.a_o, residual=h)
        h = self.mid1(h, B, H, W, None)
        hh, ww = H, W
        for res, us in self.up:
            for r in res:
                h = r(h, B, hh, ww, None)
            if us is not None:
                h = ops.conv2d(h, us, B, hh, ww, ups=2)
                hh, ww = hh * 2, ww * 2
        h = ops.group_norm(h, *self.norm_out, B, hh * ww, G=g, eps=1e-6, silu=True)
        return ops.conv2d(h, self.conv_out, B, hh, ww), hh, ww
