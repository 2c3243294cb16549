"""Model / step configurations of the CoMat hot path."""
from __future__ import annotations

from dataclasses import dataclass


@dataclass
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: tuple = (320, 640, 1280, 1280)
    down_attn: tuple = (True, True, True, False)
    layers_per_block: int = 2
    num_heads: int = 8
    cross_attention_dim: int = 768
    norm_groups: int = 32
    lora_rank: int = 128
    # SDXL extensions: per-level head counts / transformer depths, linear proj_in/out, text_time
    # additional embedding (pooled text embedding + 6 size/crop ids)
    heads_per_level: tuple = ()
    transformer_layers: tuple = ()
    linear_projection: bool = False
    addition_embed: bool = False
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


@dataclass
class BlipConfig:
    # vision (ViT)
    image_size: int = 384
    patch_size: int = 16
    v_hidden: int = 1024
    v_layers: int = 24
    v_heads: int = 16
    v_mlp: int = 4096
    v_eps: float = 1e-5
    # text decoder (BERT-style)
    vocab_size: int = 30524
    t_hidden: int = 768
    t_layers: int = 12
    t_heads: int = 12
    t_mlp: int = 3072
    t_eps: float = 1e-12
    max_pos: int = 512
    label_smoothing: float = 0.1  # transformers==4.31.0 hard-codes 0.1 in the BLIP decoder loss


SD15_UNET = UNetConfig()
SDXL_UNET = UNetConfig(block_out_channels=(320, 640, 1280), down_attn=(False, True, True), layers_per_block=2,
                       heads_per_level=(5, 10, 20), transformer_layers=(1, 2, 10), cross_attention_dim=2048,
                       linear_projection=True, addition_embed=True)
SDXL_VAE = VAEConfig(scaling_factor=0.13025)
SD15_VAE = VAEConfig()
BLIP_LARGE = BlipConfig()

TINY_UNET = UNetConfig(block_out_channels=(32, 64, 64), down_attn=(True, True, False), layers_per_block=1,
                       num_heads=2, cross_attention_dim=24, norm_groups=8, lora_rank=4)
TINY_SDXL_UNET = UNetConfig(block_out_channels=(32, 64, 64), down_attn=(False, True, True), layers_per_block=1,
                            heads_per_level=(2, 2, 4), transformer_layers=(1, 2, 3), cross_attention_dim=24,
                            norm_groups=8, lora_rank=4, linear_projection=True, addition_embed=True,
                            addition_time_embed_dim=8, pooled_dim=16)
TINY_VAE = VAEConfig(block_out_channels=(8, 16, 16, 32), layers_per_block=1, norm_groups=8)
TINY_BLIP = BlipConfig(image_size=32, patch_size=8, v_hidden=32, v_layers=2, v_heads=2, v_mlp=64, vocab_size=97,
                       t_hidden=24, t_layers=2, t_heads=2, t_mlp=48, max_pos=32)
