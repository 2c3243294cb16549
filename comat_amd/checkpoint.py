"""Checkpoint wire format of the hot path: what `Trainer.save_model_hook`
(training_script.py:390-426) writes and `load_model_hook` (:170-196) reads, so adapters trained here load into a stock
diffusers pipeline (`pipe.load_lora_weights(dir)`) and adapters trained by the reference resume here.

    {dir}/pytorch_lora_weights.safetensors     keys  unet.{attention path}.{to_q|to_k|to_v|to_out.0}.lora.{down|up}.weight
                                               (unet_lora_state_dict, training_script.py:49-64; written by
                                               LoraLoaderMixin.save_lora_weights, safetensors, metadata format=pt)
    {dir}/D_sd/pytorch_lora_weights.safetensors   the discriminator UNet's LoRA factors, same key scheme (:413-424)
    {dir}/D_sd/mlp.pt                          torch.save(nn.Sequential(nn.Linear(4, 1)).state_dict())  (:426;
                                               gan_sdxl.py:32-35) -> keys "0.weight" [1, 4], "0.bias" [1]

Frozen base weights are not part of a checkpoint; `load_safetensors` reads the upstream repositories' own files
(`unet/diffusion_pytorch_model.safetensors`, `vae/...`, BLIP `model.safetensors`): the model classes of this package
take those state dicts under their upstream names.
"""
from __future__ import annotations

import os

import torch
from safetensors.torch import load_file, save_file

LORA_WEIGHT_NAME = "pytorch_lora_weights.safetensors"
PREFIX = "unet."


def lora_state_dict(bank) -> dict:
    """{'unet.<module>.lora.down.weight': fp32 CPU tensor, ...} in the bank's parameter order."""
    return {PREFIX + n: p.detach().to("cpu", torch.float32).contiguous() for n, p in bank.params.items()}


def save_lora_weights(save_directory: str, bank, weight_name: str = LORA_WEIGHT_NAME) -> str:
    os.makedirs(save_directory, exist_ok=True)
    path = os.path.join(save_directory, weight_name)
    save_file(lora_state_dict(bank), path, metadata={"format": "pt"})
    return path


def load_lora_state_dict(path: str) -> dict:
    """Reads a LoRA safetensors file (or the directory holding `pytorch_lora_weights.safetensors`) and returns the
    UNet factors keyed WITHOUT the 'unet.' prefix (the key scheme of LoRABank / weights.make_lora_weights).
    Text-encoder LoRA entries (prefix 'text_encoder.') are not on this path and are ignored."""
    if os.path.isdir(path):
        path = os.path.join(path, LORA_WEIGHT_NAME)
    sd = load_file(path)
    return {k[len(PREFIX):]: v.float() for k, v in sd.items() if k.startswith(PREFIX)}


def load_lora_into_bank(bank, sd: dict):
    """Copies factors into the bank's flat fp32 buffer (shapes and names must match: same rank, same attention set)."""
    missing = [n for n in bank.names if n not in sd]
    extra = [n for n in sd if n not in bank.params]
    if missing or extra:
        raise KeyError(f"LoRA state dict does not match the bank: missing {missing[:3]}..., unexpected {extra[:3]}...")
    with torch.no_grad():
        for n, p in bank.params.items():
            if tuple(sd[n].shape) != tuple(p.shape):
                raise ValueError(f"{n}: shape {tuple(sd[n].shape)} != {tuple(p.shape)}")
            p.copy_(sd[n].to(p.device, torch.float32))
    bank.mark_updated()


def save_checkpoint(output_dir: str, bank, disc=None):
    """training_script.py:390-426 for the LoRA configuration (no full fine-tuning, frozen VAE / text encoder)."""
    save_lora_weights(output_dir, bank)
    if disc is not None:
        d = os.path.join(output_dir, "D_sd")
        save_lora_weights(d, disc.bank)
        torch.save({"0.weight": disc.w.detach().reshape(1, 4).cpu().clone(),
                    "0.bias": disc.b.detach().reshape(1).cpu().clone()}, os.path.join(d, "mlp.pt"))


def load_checkpoint(load_dir: str, bank, disc=None):
    """training_script.py:170-196"""
    load_lora_into_bank(bank, load_lora_state_dict(load_dir))
    if disc is not None:
        d = os.path.join(load_dir, "D_sd")
        load_lora_into_bank(disc.bank, load_lora_state_dict(d))
        mlp = torch.load(os.path.join(d, "mlp.pt"), map_location="cpu")
        with torch.no_grad():
            disc.w.copy_(mlp["0.weight"].reshape(4).to(disc.w.device, torch.float32))
            disc.b.copy_(mlp["0.bias"].reshape(1).to(disc.b.device, torch.float32))


def load_safetensors(path: str) -> dict:
    """Frozen base weights under their upstream names (e.g. runwayml/stable-diffusion-v1-5
    unet/diffusion_pytorch_model.safetensors, Salesforce/blip-image-captioning-large model.safetensors)."""
    return load_file(path)
