"""Seeded synthetic weights at the real layer shapes, keyed by the upstream state-dict names (diffusers UNet /
AutoencoderKL, transformers BLIP) so that real checkpoints can be dropped in later.
No pretrained weights exist offline; BASELINE.md §3 fixes: fan-in-scaled normal init (activations stay O(1)),
LoRA down ~ N(0, 1/r), LoRA up ~ N(0, 0.02^2) (non-zero so LoRA gradients are non-trivial)."""
from __future__ import annotations

import torch

from .config import BlipConfig, UNetConfig, VAEConfig


class _Init:
    def __init__(self, seed, perturb_norms=False):
        self.g = torch.Generator().manual_seed(seed if seed is not None else 0)
        self.sd = {}
        self.perturb = perturb_norms
        self.shapes_only = seed is None  # seed=None: meta tensors (layer shapes / parameter counts without memory)

    def randn(self, *shape, std=1.0):
        if self.shapes_only:
            return torch.empty(*shape, device="meta")
        return torch.randn(*shape, generator=self.g) * std

    def linear(self, name, fin, fout, bias=True):
        self.sd[name + ".weight"] = self.randn(fout, fin, std=fin ** -0.5)
        if bias:
            self.sd[name + ".bias"] = self.randn(fout, std=0.02)

    def conv(self, name, cin, cout, k):
        self.sd[name + ".weight"] = self.randn(cout, cin, k, k, std=(cin * k * k) ** -0.5)
        self.sd[name + ".bias"] = self.randn(cout, std=0.02)

    def norm(self, name, c):
        if self.perturb:
            self.sd[name + ".weight"] = 1.0 + self.randn(c, std=0.1)
            self.sd[name + ".bias"] = self.randn(c, std=0.1)
        else:
            dev = "meta" if self.shapes_only else "cpu"
            self.sd[name + ".weight"] = torch.ones(c, device=dev)
            self.sd[name + ".bias"] = torch.zeros(c, device=dev)


def attention_names(cfg: UNetConfig):
    """(module path, query dim, kv dim, inner dim) of every Attention in creation order of diffusers'
    `unet.attn_processors` (down, up, mid as registered; we use down -> mid -> up consistently)."""
    out = []
    nb = len(cfg.block_out_channels)

    def blk(prefix, c, level):
        for k in range(cfg.depth(level)):
            out.append((f"{prefix}.transformer_blocks.{k}.attn1", c, c, c))
            out.append((f"{prefix}.transformer_blocks.{k}.attn2", c, cfg.cross_attention_dim, c))
    for i in range(nb):
        if cfg.down_attn[i]:
            for j in range(cfg.layers_per_block):
                blk(f"down_blocks.{i}.attentions.{j}", cfg.block_out_channels[i], i)
    blk("mid_block.attentions.0", cfg.block_out_channels[-1], nb - 1)
    rev = list(reversed(cfg.block_out_channels))
    for i in range(nb):
        if cfg.up_attn[i]:
            for j in range(cfg.layers_per_block + 1):
                blk(f"up_blocks.{i}.attentions.{j}", rev[i], nb - 1 - i)
    return out


def make_unet_weights(cfg: UNetConfig, seed=1234, perturb_norms=False):
    it = _Init(seed, perturb_norms)
    c0 = cfg.block_out_channels[0]
    ted = cfg.time_embed_dim
    it.linear("time_embedding.linear_1", c0, ted)
    it.linear("time_embedding.linear_2", ted, ted)
    if cfg.addition_embed:
        it.linear("add_embedding.linear_1", cfg.pooled_dim + 6 * cfg.addition_time_embed_dim, ted)
        it.linear("add_embedding.linear_2", ted, ted)
    it.conv("conv_in", cfg.in_channels, c0, 3)

    def resnet(name, cin, cout):
        it.norm(name + ".norm1", cin)
        it.conv(name + ".conv1", cin, cout, 3)
        it.linear(name + ".time_emb_proj", ted, cout)
        it.norm(name + ".norm2", cout)
        it.conv(name + ".conv2", cout, cout, 3)
        if cin != cout:
            it.conv(name + ".conv_shortcut", cin, cout, 1)

    def transformer(name, c, level):
        it.norm(name + ".norm", c)
        if cfg.linear_projection:
            it.linear(name + ".proj_in", c, c)
        else:
            it.conv(name + ".proj_in", c, c, 1)
        for k in range(cfg.depth(level)):
            b = f"{name}.transformer_blocks.{k}"
            for a, kv in (("attn1", c), ("attn2", cfg.cross_attention_dim)):
                it.linear(f"{b}.{a}.to_q", c, c, bias=False)
                it.linear(f"{b}.{a}.to_k", kv, c, bias=False)
                it.linear(f"{b}.{a}.to_v", kv, c, bias=False)
                it.linear(f"{b}.{a}.to_out.0", c, c)
            for n in ("norm1", "norm2", "norm3"):
                it.norm(f"{b}.{n}", c)
            it.linear(f"{b}.ff.net.0.proj", c, 8 * c)
            it.linear(f"{b}.ff.net.2", 4 * c, c)
        if cfg.linear_projection:
            it.linear(name + ".proj_out", c, c)
        else:
            it.conv(name + ".proj_out", c, c, 1)

    nb = len(cfg.block_out_channels)
    ch = c0
    skip_ch = [c0]
    for i in range(nb):
        cout = cfg.block_out_channels[i]
        for j in range(cfg.layers_per_block):
            resnet(f"down_blocks.{i}.resnets.{j}", ch, cout)
            ch = cout
            if cfg.down_attn[i]:
                transformer(f"down_blocks.{i}.attentions.{j}", cout, i)
            skip_ch.append(ch)
        if i < nb - 1:
            it.conv(f"down_blocks.{i}.downsamplers.0.conv", ch, ch, 3)
            skip_ch.append(ch)
    resnet("mid_block.resnets.0", ch, ch)
    transformer("mid_block.attentions.0", ch, nb - 1)
    resnet("mid_block.resnets.1", ch, ch)
    rev = list(reversed(cfg.block_out_channels))
    for i in range(nb):
        cout = rev[i]
        for j in range(cfg.layers_per_block + 1):
            resnet(f"up_blocks.{i}.resnets.{j}", ch + skip_ch.pop(), cout)
            ch = cout
            if cfg.up_attn[i]:
                transformer(f"up_blocks.{i}.attentions.{j}", cout, nb - 1 - i)
        if i < nb - 1:
            it.conv(f"up_blocks.{i}.upsamplers.0.conv", ch, ch, 3)
    it.norm("conv_norm_out", ch)
    it.conv("conv_out", ch, cfg.out_channels, 3)
    return it.sd


def make_lora_weights(cfg: UNetConfig, seed=4321):
    """{'<attn>.to_q.lora.down.weight': [r, in], '<attn>.to_q.lora.up.weight': [out, r], ...} in the order of
    training_utils/pipeline.py:123-144 (q, k, v, out per attention)."""
    g = torch.Generator().manual_seed(seed)
    r = cfg.lora_rank
    out = {}
    for path, qd, kvd, inner in attention_names(cfg):
        for proj, fin, fout in (("to_q", qd, inner), ("to_k", kvd, inner), ("to_v", kvd, inner),
                                ("to_out.0", inner, qd)):
            out[f"{path}.{proj}.lora.down.weight"] = torch.randn(r, fin, generator=g) * (1.0 / r) ** 0.5
            out[f"{path}.{proj}.lora.up.weight"] = torch.randn(fout, r, generator=g) * 0.02
    return out


def make_vae_weights(cfg: VAEConfig, seed=2345, perturb_norms=False):
    it = _Init(seed, perturb_norms)
    it.conv("post_quant_conv", cfg.latent_channels, cfg.latent_channels, 1)
    rev = list(reversed(cfg.block_out_channels))
    ch = rev[0]
    it.conv("decoder.conv_in", cfg.latent_channels, ch, 3)

    def resnet(name, cin, cout):
        it.norm(name + ".norm1", cin)
        it.conv(name + ".conv1", cin, cout, 3)
        it.norm(name + ".norm2", cout)
        it.conv(name + ".conv2", cout, cout, 3)
        if cin != cout:
            it.conv(name + ".conv_shortcut", cin, cout, 1)
    resnet("decoder.mid_block.resnets.0", ch, ch)
    a = "decoder.mid_block.attentions.0"
    it.norm(a + ".group_norm", ch)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        it.linear(f"{a}.{n}", ch, ch)
    resnet("decoder.mid_block.resnets.1", ch, ch)
    for i, cout in enumerate(rev):
        for j in range(cfg.layers_per_block + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", ch, cout)
            ch = cout
        if i < len(rev) - 1:
            it.conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", ch, ch, 3)
    it.norm("decoder.conv_norm_out", ch)
    it.conv("decoder.conv_out", ch, cfg.out_channels, 3)
    return it.sd


def make_blip_weights(cfg: BlipConfig, seed=3456, perturb_norms=False):
    """transformers BlipForConditionalGeneration state-dict names (modeling_blip.py / modeling_blip_text.py)."""
    it = _Init(seed, perturb_norms)
    d, P = cfg.v_hidden, cfg.patch_size
    n_pos = (cfg.image_size // P) ** 2 + 1
    v = "vision_model."
    it.sd[v + "embeddings.class_embedding"] = it.randn(1, 1, d, std=0.02)
    it.sd[v + "embeddings.position_embedding"] = it.randn(1, n_pos, d, std=0.02)
    it.conv(v + "embeddings.patch_embedding", 3, d, P)
    for i in range(cfg.v_layers):
        L = f"{v}encoder.layers.{i}."
        it.norm(L + "layer_norm1", d)
        it.linear(L + "self_attn.qkv", d, 3 * d)
        it.linear(L + "self_attn.projection", d, d)
        it.norm(L + "layer_norm2", d)
        it.linear(L + "mlp.fc1", d, cfg.v_mlp)
        it.linear(L + "mlp.fc2", cfg.v_mlp, d)
    it.norm(v + "post_layernorm", d)
    t = "text_decoder.bert."
    h = cfg.t_hidden
    it.sd[t + "embeddings.word_embeddings.weight"] = it.randn(cfg.vocab_size, h, std=0.05)
    it.sd[t + "embeddings.position_embeddings.weight"] = it.randn(cfg.max_pos, h, std=0.05)
    it.norm(t + "embeddings.LayerNorm", h)
    for i in range(cfg.t_layers):
        L = f"{t}encoder.layer.{i}."
        for n in ("query", "key", "value"):
            it.linear(L + "attention.self." + n, h, h)
        it.linear(L + "attention.output.dense", h, h)
        it.norm(L + "attention.output.LayerNorm", h)
        it.linear(L + "crossattention.self.query", h, h)
        it.linear(L + "crossattention.self.key", d, h)
        it.linear(L + "crossattention.self.value", d, h)
        it.linear(L + "crossattention.output.dense", h, h)
        it.norm(L + "crossattention.output.LayerNorm", h)
        it.linear(L + "intermediate.dense", h, cfg.t_mlp)
        it.linear(L + "output.dense", cfg.t_mlp, h)
        it.norm(L + "output.LayerNorm", h)
    c = "text_decoder.cls.predictions."
    it.linear(c + "transform.dense", h, h)
    it.norm(c + "transform.LayerNorm", h)
    it.sd[c + "bias"] = it.randn(cfg.vocab_size, std=0.02)
    # decoder.weight is tied to the word embeddings
    return it.sd
