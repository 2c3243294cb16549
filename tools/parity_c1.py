"""BASELINE config C1 at FULL size — SD1.5, 1 prompt, 2 denoise steps (both trained), concept-matching loss only, fp32:
the product (HIP kernels, exact-f32 MFMA) against the CPU oracle on the same seeded weights and inputs.  Reports what
the north star asks for: relative L2 error of the generator-LoRA gradients (bar: 1e-3) and the token-level concept
scores (BLIP per-token log-probs).  The oracle needs a few minutes of host time (12.6 TFLOP fp32 on the CPU).

    python tools/parity_c1.py                 # on the GPU box (writes one JSON line)
    python tools/parity_c1.py --tiny --sim    # plumbing check anywhere (tiny shapes, ABI simulator instead of the GPU)
"""
import argparse
import dataclasses
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from comat_amd import config, ops, weights  # noqa: E402
from comat_amd.blip import Blip  # noqa: E402
from comat_amd.pipeline import TrainableSDPipeline  # noqa: E402
from comat_amd.step import CoMatTrainer, StepConfig  # noqa: E402
from comat_amd.unet import LoRABank, UNet, VAEDecoder  # noqa: E402
from oracle import blip as OB  # noqa: E402
from oracle import sd as O  # noqa: E402
from oracle import step as OS  # noqa: E402


def rel_l2(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiny", action="store_true")
    ap.add_argument("--sim", action="store_true", help="ABI simulator on the CPU instead of the HIP library")
    args = ap.parse_args()
    if args.sim:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from sim_backend import SimKernels
        ops.set_kernel_backend(SimKernels())
        dev = torch.device("cpu")
    else:
        from comat_amd import _hip
        ops.set_kernel_backend(_hip.HipKernels())
        dev = torch.device("cuda:0")
    ucfg, vcfg, bcfg = ((config.TINY_UNET, config.TINY_VAE, config.TINY_BLIP) if args.tiny
                        else (config.SD15_UNET, config.SD15_VAE, config.BLIP_LARGE))
    res = 64 if args.tiny else 512
    dtype = torch.float32
    usd = weights.make_unet_weights(ucfg, seed=1234)
    vsd = weights.make_vae_weights(vcfg, seed=2345)
    bsd = weights.make_blip_weights(bcfg, seed=3456)
    lsd = weights.make_lora_weights(ucfg, seed=4321)
    scfg = StepConfig(resolution=res, total_step=2, K=2, gan_loss=False, attrcon=False)
    g = torch.Generator().manual_seed(1000)
    L, T, h = (7, 9, res // 8) if args.tiny else (77, 16, res // 8)
    ids = torch.randint(1000 if not args.tiny else 1, bcfg.vocab_size - 2, (1, T), generator=g)
    batch = dict(prompt_embeds=torch.randn(1, L, ucfg.cross_attention_dim, generator=g),
                 negative_prompt_embeds=torch.randn(1, L, ucfg.cross_attention_dim, generator=g),
                 latents=torch.randn(1, 4, h, h, generator=torch.Generator().manual_seed(42)),
                 noises=[torch.randn(1, 4, h, h, generator=torch.Generator().manual_seed(100 + i)) for i in range(2)],
                 blip_input_ids=ids, blip_attention_mask=torch.ones_like(ids))
    off = res // 224  # crop geometry of training_script.py:606-609 (512 -> offsets in [0, 2], size 510)
    ts, crop = [0, 1], (min(1, off), min(1, off), res - off, res - off)

    bank = LoRABank(ucfg, lsd, dtype, dev)
    pipe = TrainableSDPipeline(UNet(ucfg, usd, dtype, dev, bank), VAEDecoder(vcfg, vsd, dtype, dev))
    trainer = CoMatTrainer(pipe, bank, Blip(bcfg, bsd, dtype, dev), None, scfg, seed=0)
    t0 = time.time()
    bank.set_requires_grad(True)
    bank.zero_grad()
    out = trainer.compute_losses(batch, training_steps=ts, crop=crop)
    out["loss"].backward()
    if dev.type == "cuda":
        torch.cuda.synchronize()
    t_prod = time.time() - t0

    W = dict(unet=usd, vae=vsd, blip=bsd, ucfg=O.UNetConfig(**dataclasses.asdict(ucfg)),
             vcfg=O.VAEConfig(**dataclasses.asdict(vcfg)), bcfg=OB.BlipConfig(**dataclasses.asdict(bcfg)),
             lora={k: v.clone().requires_grad_(True) for k, v in lsd.items()})
    t0 = time.time()
    ref = OS.g_loss_terms(W, batch, scfg, ts, crop)
    ref["loss"].backward()
    t_ref = time.time() - t0
    g_ref = torch.cat([W["lora"][n].grad.reshape(-1) for n in bank.names])
    worst = max(rel_l2(bank.params[n].grad, W["lora"][n].grad) for n in bank.names)
    print(json.dumps({
        "config": "C1" + (" (tiny)" if args.tiny else " (SD1.5 full size)"), "backend": "sim" if args.sim else "hip",
        "dtype": "f32", "loss": float(out["loss"]), "oracle_loss": float(ref["loss"]),
        "lora_grad_rel_l2_flat": rel_l2(bank.flat_grad, g_ref), "lora_grad_rel_l2_worst_tensor": worst,
        "token_logp_max_abs_diff": float((out["token_logp"].detach().cpu() - ref["token_logp"].detach()).abs().max()),
        "image_rel_l2": rel_l2(ops.tokens_to_nchw(out["image"][0], 1, out["image"][1], out["image"][2]), ref["image"]),
        "product_s": round(t_prod, 2), "oracle_s": round(t_ref, 1), "host_threads": torch.get_num_threads()}))


if __name__ == "__main__":
    main()
