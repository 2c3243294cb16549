"""LoRA / discriminator checkpoint wire format (training_script.py:49-64, 390-426, 170-196)."""
import os

import torch
from safetensors import safe_open

from comat_amd import checkpoint, config, weights
from comat_amd.gan import D_sd
from comat_amd.unet import LoRABank, UNet


def _bank(dev, seed):
    lsd = weights.make_lora_weights(config.TINY_UNET, seed=seed)
    return LoRABank(config.TINY_UNET, lsd, torch.bfloat16, dev), lsd


def test_lora_wire_format_round_trip(dev, tmp_path):
    sim = dev  # runs on the ABI simulator (CPU) and, marked gpu, on the real kernels / device buffers
    bank, lsd = _bank(sim, 1)
    path = checkpoint.save_lora_weights(str(tmp_path), bank)
    assert os.path.basename(path) == "pytorch_lora_weights.safetensors"
    with safe_open(path, framework="pt") as f:
        keys = sorted(f.keys())
        assert f.metadata() == {"format": "pt"}
    # the reference's key scheme: f"unet.{module name}.lora.{down|up}.weight" for every to_q/to_k/to_v/to_out.0
    assert len(keys) == len(lsd) == 8 * len(weights.attention_names(config.TINY_UNET))
    assert all(k.startswith("unet.") and (k.endswith(".lora.down.weight") or k.endswith(".lora.up.weight")) for k in keys)
    assert "unet.mid_block.attentions.0.transformer_blocks.0.attn2.to_out.0.lora.up.weight" in keys
    back = checkpoint.load_lora_state_dict(str(tmp_path))
    assert sorted(back) == sorted(lsd)
    for k in lsd:
        assert back[k].dtype == torch.float32 and torch.equal(back[k], lsd[k])
    # into a bank initialised differently: parameters, and the derived compute copies, follow
    other, _ = _bank(sim, 2)
    other.ensure_compute_copy()
    checkpoint.load_lora_into_bank(other, back)
    assert torch.equal(other.flat.cpu(), bank.flat.cpu())
    other.ensure_compute_copy()
    assert torch.equal(other.flat_c.cpu(), bank.flat.to(torch.bfloat16).cpu())
    g = other.groups[0]
    dc, ucs, dct, uts = g.compute_copies()
    assert torch.equal(dct.cpu(), dc.t().cpu())
    assert all(torch.equal(ut.cpu(), uc.t().cpu()) for ut, uc in zip(uts, ucs))


def test_checkpoint_with_discriminator(dev, tmp_path):
    sim = dev
    bank, _ = _bank(sim, 3)
    dbank, _ = _bank(sim, 4)
    usd = weights.make_unet_weights(config.TINY_UNET, seed=5)
    disc = D_sd(UNet(config.TINY_UNET, usd, torch.bfloat16, sim, dbank), dbank, torch.tensor([0.1, -0.2, 0.3, 0.4]),
                torch.tensor([0.05]))
    checkpoint.save_checkpoint(str(tmp_path), bank, disc)
    assert sorted(os.listdir(tmp_path / "D_sd")) == ["mlp.pt", "pytorch_lora_weights.safetensors"]
    # the head file is a state dict of the reference's nn.Sequential(nn.Linear(4, 1)) (gan_sdxl.py:32-35)
    mlp = torch.nn.Sequential(torch.nn.Linear(4, 1))
    mlp.load_state_dict(torch.load(tmp_path / "D_sd" / "mlp.pt"))
    assert torch.allclose(mlp[0].weight.detach().reshape(-1), torch.tensor([0.1, -0.2, 0.3, 0.4]))
    bank2, _ = _bank(sim, 6)
    dbank2, _ = _bank(sim, 7)
    disc2 = D_sd(UNet(config.TINY_UNET, usd, torch.bfloat16, sim, dbank2), dbank2, torch.zeros(4), torch.zeros(1))
    checkpoint.load_checkpoint(str(tmp_path), bank2, disc2)
    assert torch.equal(bank2.flat, bank.flat) and torch.equal(dbank2.flat, dbank.flat)
    assert torch.equal(disc2.head, disc.head)


def test_upstream_format_base_weights_load(dev, tmp_path):
    """Frozen base weights travel as safetensors files under the upstream (diffusers / transformers) parameter names
    (training_script.py:170-196 loads them through the upstream classes): a UNet and a VAE decoder built from files written
    by a generator give the same outputs as ones built from the in-memory state dicts; names and shapes of the full-size
    dicts are pinned by tests/test_architectures.py."""
    from safetensors.torch import save_file

    from comat_amd.unet import VAEDecoder
    dtype = torch.bfloat16
    usd = weights.make_unet_weights(config.TINY_UNET, seed=11)
    vsd = weights.make_vae_weights(config.TINY_VAE, seed=12)
    save_file({k: v.contiguous() for k, v in usd.items()}, str(tmp_path / "unet.safetensors"))
    save_file({k: v.contiguous() for k, v in vsd.items()}, str(tmp_path / "vae.safetensors"))
    usd2 = checkpoint.load_safetensors(str(tmp_path / "unet.safetensors"))
    vsd2 = checkpoint.load_safetensors(str(tmp_path / "vae.safetensors"))
    assert sorted(usd2) == sorted(usd) and all(torch.equal(usd2[k], usd[k]) for k in usd)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2 * 8 * 8, 4, generator=g).to(dev, dtype)
    ctx = torch.randn(2 * 7, config.TINY_UNET.cross_attention_dim, generator=g).to(dev, dtype)
    with torch.no_grad():
        a, _ = UNet(config.TINY_UNET, usd, dtype, dev)(x, 2, 8, 8, 500, ctx, 7)
        b, _ = UNet(config.TINY_UNET, usd2, dtype, dev)(x, 2, 8, 8, 500, ctx, 7)
        va, _, _ = VAEDecoder(config.TINY_VAE, vsd, dtype, dev)(x[:64], 1, 8, 8)
        vb, _, _ = VAEDecoder(config.TINY_VAE, vsd2, dtype, dev)(x[:64], 1, 8, 8)
    assert torch.equal(a, b) and torch.equal(va, vb)
