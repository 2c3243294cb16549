"""LoRA / discriminator checkpoint wire format (training_script.py:49-64, 390-426, 170-196)."""
import os

import torch
from safetensors import safe_open

from comat_amd import checkpoint, config, weights
from comat_amd.gan import D_sd
from comat_amd.unet import LoRABank, UNet


def _bank(sim, seed):
    lsd = weights.make_lora_weights(config.TINY_UNET, seed=seed)
    return LoRABank(config.TINY_UNET, lsd, torch.bfloat16, sim), lsd


def test_lora_wire_format_round_trip(sim, tmp_path):
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
    dc, ucs, dct = g.compute_copies()
    assert torch.equal(dct, dc.t())


def test_checkpoint_with_discriminator(sim, tmp_path):
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
