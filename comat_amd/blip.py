"""BLIP captioner loss ("concept matching" reward) on the HIP operators.

Mirrors `Blip.score` (concept_mat_utils/caption_blip.py:43-59) and the `BlipForConditionalGeneration` forward it
calls: random-crop + antialiased bicubic resize to 384 + CLIP normalise (one fused resampling kernel), ViT-L/16
patch embedding as patch-gather + MFMA GEMM, pre-LN encoder layers, BERT-style causal decoder with cross-attention
to the 577 image tokens, tied LM head and shifted cross-entropy with `ignore_index=-100`.  All weights are frozen
(caption_blip.py:20-21): only data-gradients flow, back to the decoded image.  The fused `qkv` projection of the
vision attention is split into three frozen projections at load time so that q/k/v are plain token matrices.
Tokenisation is outside the hot path: `score` takes input ids.
"""
from __future__ import annotations

import os

import torch
from torch.autograd import Function

from . import ops
from .config import BlipConfig
from .resize import resize_tables

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


class _PrependClassToken(Function):
    """[B*N, d] patch tokens -> [B*(N+1), d] with the class embedding in row 0 of every sample."""

    @staticmethod
    def forward(ctx, patches, cls, B):
        patches = ops._c(patches)
        N, d = patches.shape[0] // B, patches.shape[1]
        out = patches.new_empty((B * (N + 1), d))
        k = ops.kernels()
        k.copy2d(patches, N * d, out[1:], (N + 1) * d, B, N * d)
        for b in range(B):
            k.unary(ops.UN_COPY, cls, out[b * (N + 1): b * (N + 1) + 1], d)
        ctx.cfg = (B, N, d)
        return out

    @staticmethod
    def backward(ctx, g):
        B, N, d = ctx.cfg
        g = ops._c(g)
        gp = g.new_empty((B * N, d))
        ops.kernels().copy2d(g[1:], (N + 1) * d, gp, N * d, B, N * d)
        return gp, None, None


class Blip:
    def __init__(self, cfg: BlipConfig, sd: dict, dtype=torch.bfloat16, device="cuda", fused_qkv=None):
        self.cfg, self.dtype, self.device = cfg, dtype, device
        T = dtype
        # ViT q/k/v as one GEMM + strided fused attention (ops.fused_qkv_attention), as the checkpoint stores the
        # projection; validated on MI355X in round 2 (COMAT_BLIP_FUSED_QKV=0 restores three separate projections)
        if fused_qkv is None:
            fused_qkv = os.environ.get("COMAT_BLIP_FUSED_QKV", "1") != "0"
        hd = cfg.v_hidden // cfg.v_heads
        self.fused_qkv = bool(fused_qkv) and ops.flash_ok(hd, dtype)

        def lin(name):
            return ops.FrozenLinear(sd[name + ".weight"], sd.get(name + ".bias"), T, device)

        def norm(name):
            f = lambda t: t.to(device=device, dtype=torch.float32).contiguous()
            return f(sd[name + ".weight"]), f(sd[name + ".bias"])

        v = "vision_model."
        d = cfg.v_hidden
        pw = sd[v + "embeddings.patch_embedding.weight"]  # [d, 3, P, P] -> [d, P, P, 3] rows (ky, kx, c)
        self.patch = ops.FrozenLinear(pw.permute(0, 2, 3, 1).reshape(d, -1), sd[v + "embeddings.patch_embedding.bias"],
                                      T, device)
        self.cls = sd[v + "embeddings.class_embedding"].reshape(1, d).to(device=device, dtype=T).contiguous()
        self.pos = sd[v + "embeddings.position_embedding"].reshape(-1).to(device=device, dtype=T).contiguous()
        self.vlayers = []
        for i in range(cfg.v_layers):
            L = f"{v}encoder.layers.{i}."
            wqkv, bqkv = sd[L + "self_attn.qkv.weight"], sd[L + "self_attn.qkv.bias"]
            if self.fused_qkv:  # one Linear(d -> 3d), as the checkpoint stores it
                qkv = ops.FrozenLinear(wqkv, bqkv, T, device)
            else:
                qkv = [ops.FrozenLinear(wqkv[j * d:(j + 1) * d], bqkv[j * d:(j + 1) * d], T, device) for j in range(3)]
            self.vlayers.append(dict(ln1=norm(L + "layer_norm1"), qkv=qkv, proj=lin(L + "self_attn.projection"),
                                     ln2=norm(L + "layer_norm2"), fc1=lin(L + "mlp.fc1"), fc2=lin(L + "mlp.fc2")))
        self.post_ln = norm(v + "post_layernorm")
        t = "text_decoder.bert."
        self.word = sd[t + "embeddings.word_embeddings.weight"].to(device=device, dtype=T).contiguous()
        self.tpos = sd[t + "embeddings.position_embeddings.weight"].to(device=device, dtype=T).contiguous()
        self.emb_ln = norm(t + "embeddings.LayerNorm")
        self.tlayers = []
        for i in range(cfg.t_layers):
            L = f"{t}encoder.layer.{i}."
            self.tlayers.append(dict(
                sq=lin(L + "attention.self.query"), sk=lin(L + "attention.self.key"), sv=lin(L + "attention.self.value"),
                so=lin(L + "attention.output.dense"), sln=norm(L + "attention.output.LayerNorm"),
                cq=lin(L + "crossattention.self.query"), ck=lin(L + "crossattention.self.key"),
                cv=lin(L + "crossattention.self.value"), co=lin(L + "crossattention.output.dense"),
                cln=norm(L + "crossattention.output.LayerNorm"),
                fi=lin(L + "intermediate.dense"), fo=lin(L + "output.dense"), fln=norm(L + "output.LayerNorm")))
        c = "text_decoder.cls.predictions."
        self.head_dense = lin(c + "transform.dense")
        self.head_ln = norm(c + "transform.LayerNorm")
        self.lm_head = ops.FrozenLinear(sd[t + "embeddings.word_embeddings.weight"], sd[c + "bias"], T, device)
        std = torch.tensor(CLIP_STD)
        self.norm_scale = (1.0 / std).to(device)
        self.norm_shift = (-torch.tensor(CLIP_MEAN) / std).to(device)
        self._tables = {}
        self.static_tables = None
        self._static_key = None

    # ---- image preprocessing -----------------------------------------------------------------------------------
    def _crop_tables(self, H, W, crop):
        key = (H, W, tuple(crop))
        if key not in self._tables:
            fwd, bwd = resize_tables(H, W, crop, (self.cfg.image_size, self.cfg.image_size), "bicubic")
            self._tables[key] = ops.ResampleTables(fwd, bwd, H, W, self.cfg.image_size, self.cfg.image_size,
                                                   self.device)
        return self._tables[key]

    def tables(self, H, W, crop):
        """the crop + resize operator of (H, W, crop).  Once fixed-address tables are installed (captured graphs of the
        step read them: step.GraphedStep, segments.SegmentedStep) every caller gets THOSE, loaded with the operator it
        asked for - so an eager call after a captured one can never preprocess with a stale crop.  Inside a capture the
        tables are handed out as they are (the replaying code loads them before each replay)."""
        st = self.static_tables
        if st is None:
            return self._crop_tables(H, W, crop)
        if (H, W) != (st.Hin, st.Win):
            raise ValueError(f"static resampling tables were installed for {st.Hin}x{st.Win} images, got {H}x{W}")
        capturing = torch.device(self.device).type == "cuda" and torch.cuda.is_current_stream_capturing()
        if not capturing and self._static_key != (H, W, tuple(crop)):
            st.load(self._crop_tables(H, W, crop))
            self._static_key = (H, W, tuple(crop))
        return st

    def install_static_tables(self, H, W, crop):
        """fixed-address tables (with spare taps for any other crop of the same size), loaded with `crop`'s operator"""
        if self.static_tables is None:
            self.static_tables = self._crop_tables(H, W, crop).static_copy()
            self._static_key = (H, W, tuple(crop))
        return self.tables(H, W, crop)

    def preprocess(self, img_tokens, B, H, W, crop=None):
        """crop (y0, x0, h, w) + Resize(bicubic, antialias) + Normalize in one kernel; -> [B*S*S, 3]."""
        crop = crop or (0, 0, H, W)
        y0, x0, ch, cw = crop
        if not (0 <= y0 and 0 <= x0 and ch > 0 and cw > 0 and y0 + ch <= H and x0 + cw <= W):
            raise ValueError(f"crop {crop} does not fit the {H}x{W} image (training_script.py:606-609 keeps it inside)")
        return ops.resample(img_tokens, self.tables(H, W, crop), B, 3, scale=self.norm_scale, shift=self.norm_shift,
                            out_dtype=self.dtype)  # (the image may come from a VAE of another storage type)

    # ---- vision encoder ----------------------------------------------------------------------------------------
    def vision(self, pixel_tokens, B):
        cfg = self.cfg
        S, P, d, nh = cfg.image_size, cfg.patch_size, cfg.v_hidden, cfg.v_heads
        N = (S // P) ** 2 + 1
        h = ops.linear(ops.patchify(pixel_tokens, B, S, S, 3, P), self.patch)
        h = _PrependClassToken.apply(h, self.cls, B)
        h = ops.add_rowvec(h.reshape(B, N * d), self.pos[: N * d]).reshape(B * N, d)
        for Lr in self.vlayers:
            x, h = ops.layer_norm_fork(h, *Lr["ln1"], eps=cfg.v_eps)
            if self.fused_qkv:
                o = ops.fused_qkv_attention(x, Lr["qkv"], B, N, nh)
            else:
                q, k, v = (ops.linear(x, w) for w in Lr["qkv"])
                o, _ = ops.attention(q, k, v, B, N, N, nh, d // nh, need_probs=False)
            h = ops.linear(o, Lr["proj"], residual=h)
            x, h = ops.layer_norm_fork(h, *Lr["ln2"], eps=cfg.v_eps)
            h = ops.linear(ops.gelu(ops.linear(x, Lr["fc1"])), Lr["fc2"], residual=h)
        return ops.layer_norm(h, *self.post_ln, eps=cfg.v_eps), N

    # ---- text decoder ------------------------------------------------------------------------------------------
    def decoder_logits(self, input_ids, attention_mask, image_embeds, B, N_img):
        cfg = self.cfg
        T = input_ids.shape[1]
        hdim, nh = cfg.t_hidden, cfg.t_heads
        ids = input_ids.to(self.device).reshape(-1)
        h = ops.embedding(ids, self.word)
        h = ops.add_rowvec(h.reshape(B, T * hdim), self.tpos[:T].reshape(-1)).reshape(B * T, hdim)
        h = ops.layer_norm(h, *self.emb_ln, eps=cfg.t_eps)
        km = attention_mask.to(device=self.device, dtype=torch.int8).contiguous()
        for Lr in self.tlayers:
            q, k, v = ops.linear(h, Lr["sq"]), ops.linear(h, Lr["sk"]), ops.linear(h, Lr["sv"])
            o, _ = ops.attention(q, k, v, B, T, T, nh, hdim // nh, causal=True, key_mask=km)
            h = ops.layer_norm(ops.linear(o, Lr["so"], residual=h), *Lr["sln"], eps=cfg.t_eps)
            q = ops.linear(h, Lr["cq"])
            k, v = ops.linear(image_embeds, Lr["ck"]), ops.linear(image_embeds, Lr["cv"])
            o, _ = ops.attention(q, k, v, B, T, N_img, nh, hdim // nh, need_probs=False)
            h = ops.layer_norm(ops.linear(o, Lr["co"], residual=h), *Lr["cln"], eps=cfg.t_eps)
            f = ops.gelu(ops.linear(h, Lr["fi"]))
            h = ops.layer_norm(ops.linear(f, Lr["fo"], residual=h), *Lr["fln"], eps=cfg.t_eps)
        x = ops.layer_norm(ops.gelu(ops.linear(h, self.head_dense)), *self.head_ln, eps=cfg.t_eps)
        return ops.linear(x, self.lm_head)  # [B*T, V]

    def caption_loss(self, pixel_tokens, B, input_ids, attention_mask, labels, label_smoothing=None):
        """mean shifted CE (`BlipForConditionalGeneration(...).loss`); returns (loss, logits [B*T,V], logp [B,T-1])."""
        ls = self.cfg.label_smoothing if label_smoothing is None else label_smoothing
        emb, N = self.vision(pixel_tokens, B)
        logits = self.decoder_logits(input_ids, attention_mask, emb, B, N)
        T = input_ids.shape[1]
        labels = labels.to(self.device)
        shifted = torch.cat([labels[:, 1:], torch.full((B, 1), -100, dtype=torch.int64, device=self.device)], dim=1)
        loss, logp = ops.cross_entropy(logits, shifted.reshape(-1).contiguous(), -100, ls)
        return loss, logits, logp.reshape(B, T)[:, :-1]

    @staticmethod
    def make_labels(input_ids, pad_token_id=0, prompt_length=4):
        """caption_blip.py:51-54: pads and the 'a photography of' prefix are ignored."""
        labels = input_ids.masked_fill(input_ids == pad_token_id, -100)
        labels[:, :prompt_length] = -100  # masked_fill returned a copy: the caller's ids are untouched
        return labels

    def score(self, images, B, H, W, input_ids, attention_mask, crop=None, pad_token_id=0, prompt_length=4,
              label_smoothing=None):
        """`Blip.score`: images are channels-last tokens [B*H*W, 3] in [0,1] (unclamped, TrainableSDPipeline.py:223).
        Returns (reward = -loss, token log-probs [B, T-1])."""
        pv = self.preprocess(images, B, H, W, crop)
        labels = self.make_labels(input_ids, pad_token_id, prompt_length)
        loss, _, logp = self.caption_loss(pv, B, input_ids, attention_mask, labels, label_smoothing)
        return -loss, logp
