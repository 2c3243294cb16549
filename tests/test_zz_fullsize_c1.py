"""BASELINE config C1 at FULL size on the GPU (SD1.5, 1 prompt, 2 trained denoise steps, concept-matching loss, fp32
exact-f32 MFMA) against tests/golden/c1_full.npz = the CPU oracle's result on the same seeded weights and inputs
(tests/golden/make_c1_golden.py).  The acceptance numbers of the north star at the real model size: LoRA gradients
within 1e-3 relative, identical token-level concept scores.  The 25.5 M-element gradient is compared through the stored
functionals: per-tensor norms and 8 Rademacher inner products per tensor (the RMS of their differences estimates the
error norm of that tensor's gradient).

Sorted last on purpose (30 s; a failure here should not mask the rest of the suite).  First passed on an MI355X in
round 2 (profiles/r02_a_call1_acceptance_and_variants.txt); it runs by default under `-m gpu`."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

pytestmark = [pytest.mark.gpu]
BF16_FULL_GRAD_LIMIT = 0.042  # 2 x the 2.1e-2 measured on an MI355X (profiles/r04_z_bf16_errors.txt; 1.7e-2 in round 3: the figure moves with every change of a summation order)


def test_c1_full_size_bf16_error_is_bounded(hip):
    """The same C1 step with bf16 storage (the bench's dtype) against the fp32 golden: there is no 1e-3 claim in bf16 -
    this test RECORDS the error of the flat LoRA gradient at the real model size ($COMAT_TEST_REPORT) and bounds it, so a
    kernel change that degrades bf16 numerics (accumulation dtype, a dropped rounding step) shows up at full size."""
    from make_c1_golden import c1_inputs, rademacher

    from comat_amd.blip import Blip
    from comat_amd.pipeline import TrainableSDPipeline
    from comat_amd.step import CoMatTrainer
    from comat_amd.unet import LoRABank, UNet, VAEDecoder
    gold = np.load(os.path.join(HERE, "golden", "c1_full.npz"))
    (ucfg, vcfg, bcfg), sd, batch, scfg, ts, crop = c1_inputs()
    dtype = torch.bfloat16
    bank = LoRABank(ucfg, sd["lora"], dtype, hip)
    pipe = TrainableSDPipeline(UNet(ucfg, sd["unet"], dtype, hip, bank), VAEDecoder(vcfg, sd["vae"], dtype, hip))
    trainer = CoMatTrainer(pipe, bank, Blip(bcfg, sd["blip"], dtype, hip), None, scfg, seed=0)
    bank.set_requires_grad(True)
    bank.zero_grad()
    out = trainer.compute_losses(batch, training_steps=ts, crop=crop)
    out["loss"].backward()
    torch.cuda.synchronize()
    names = [str(n) for n in gold["names"]]
    total_ref = float(np.sqrt((gold["grad_norm"] ** 2).sum()))
    err_sq, norm_sq = 0.0, 0.0
    for i, n in enumerate(names):
        gr = bank.params[n].grad.detach().double().cpu().reshape(-1)
        p = (rademacher(n, gr.numel()).double() @ gr).numpy()
        err_sq += float(np.mean((p - gold["grad_proj"][i]) ** 2))
        norm_sq += float(gr.norm()) ** 2
    rel = float(np.sqrt(err_sq)) / total_ref
    loss_rel = abs(float(out["loss"]) - float(gold["loss"])) / abs(float(gold["loss"]))
    logp_err = float(np.abs(out["token_logp"].detach().float().cpu().numpy() - gold["token_logp"]).max())
    path = os.environ.get("COMAT_TEST_REPORT")
    if path:
        with open(path, "a") as f:
            f.write(f"c1_full bfloat16 cuda grad_rel_err={rel:.3e} grad_norm_ratio={np.sqrt(norm_sq) / total_ref:.4f} "
                    f"loss_rel={loss_rel:.3e} token_logp_maxerr={logp_err:.3e}\n")
    assert np.isfinite(rel) and rel < BF16_FULL_GRAD_LIMIT, f"bf16 flat LoRA gradient: estimated rel. error {rel:.3e}"
    assert abs(np.sqrt(norm_sq) / total_ref - 1.0) < BF16_FULL_GRAD_LIMIT
    assert loss_rel < 2e-2 and logp_err < 0.25


def test_c1_full_size_matches_oracle_golden(hip):
    from make_c1_golden import NPROJ, c1_inputs, rademacher

    from comat_amd import ops
    from comat_amd.blip import Blip
    from comat_amd.pipeline import TrainableSDPipeline
    from comat_amd.step import CoMatTrainer
    from comat_amd.unet import LoRABank, UNet, VAEDecoder
    gold = np.load(os.path.join(HERE, "golden", "c1_full.npz"))
    (ucfg, vcfg, bcfg), sd, batch, scfg, ts, crop = c1_inputs()
    dtype = torch.float32
    bank = LoRABank(ucfg, sd["lora"], dtype, hip)
    pipe = TrainableSDPipeline(UNet(ucfg, sd["unet"], dtype, hip, bank), VAEDecoder(vcfg, sd["vae"], dtype, hip))
    trainer = CoMatTrainer(pipe, bank, Blip(bcfg, sd["blip"], dtype, hip), None, scfg, seed=0)
    bank.set_requires_grad(True)
    bank.zero_grad()
    out = trainer.compute_losses(batch, training_steps=ts, crop=crop)
    out["loss"].backward()
    torch.cuda.synchronize()
    # scalars and the token-level concept scores
    assert abs(float(out["loss"]) - float(gold["loss"])) < 2e-4 * abs(float(gold["loss"]))
    assert np.abs(out["token_logp"].detach().cpu().numpy() - gold["token_logp"]).max() < 2e-3
    img = ops.tokens_to_nchw(out["image"][0], 1, out["image"][1], out["image"][2]).detach().cpu()
    assert abs(float(img.double().norm()) - float(gold["image_norm"])) < 1e-4 * float(gold["image_norm"])
    assert np.abs(img[0, :, ::64, ::64].numpy() - gold["image_samples"]).max() < 1e-3 * np.abs(gold["image_samples"]).max()
    assert abs(float(out["training_latents"].double().norm()) - float(gold["latents_norm"])) < 1e-4 * float(gold["latents_norm"])
    # the LoRA gradient, tensor by tensor
    names = [str(n) for n in gold["names"]]
    assert sorted(bank.names) == names
    total_ref = float(np.sqrt((gold["grad_norm"] ** 2).sum()))
    err_sq = 0.0
    for i, n in enumerate(names):
        gr = bank.params[n].grad.detach().double().cpu().reshape(-1)
        nref = float(gold["grad_norm"][i])
        assert abs(float(gr.norm()) - nref) <= 1e-3 * nref + 1e-6 * total_ref, f"{n}: gradient norm"
        p = (rademacher(n, gr.numel()).double() @ gr).numpy()
        est = float(np.sqrt(np.mean((p - gold["grad_proj"][i]) ** 2)))  # ~ |g - g_ref| for this tensor
        assert est <= 3e-3 * nref + 3e-6 * total_ref, f"{n}: estimated gradient error {est:.3e} vs norm {nref:.3e}"
        err_sq += est ** 2
    # measured 1.4e-5 (profiles/r04_z_grad_shrink_bisect.txt, row fff): an order of magnitude inside the north star's 1e-3
    assert np.sqrt(err_sq) <= 1e-4 * total_ref, f"flat LoRA gradient: estimated rel. error {np.sqrt(err_sq) / total_ref:.3e}"
    assert NPROJ == gold["grad_proj"].shape[1]


def test_c2_full_size_gan_leg_matches_oracle_golden(hip):
    """The GAN leg at FULL size ("full-size parity exists for C1's loss set only"): C2's loss set - concept matching +
    generator-side discriminator loss, then the discriminator step on [fake.detach(); real] - on the SD1.5 generator and the SD1.5
    discriminator in fp32 (exact-f32 MFMA) against tests/golden/c2_full.npz = the CPU oracle on the same seeded world
    (tests/golden/make_c2_golden.py).  Scalars, the generator's and the discriminator's LoRA gradients (per-tensor norms + 8
    Rademacher projections each) within the north star's 1e-3, the discriminator head's gradient element by element."""
    from make_c2_golden import c2_inputs
    from make_c1_golden import rademacher

    from comat_amd.blip import Blip
    from comat_amd.gan import D_sd
    from comat_amd.pipeline import TrainableSDPipeline
    from comat_amd.step import CoMatTrainer
    from comat_amd.unet import LoRABank, UNet, VAEDecoder
    path = os.path.join(HERE, "golden", "c2_full.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/c2_full.npz not generated")
    gold = np.load(path)
    (ucfg, vcfg, bcfg), sd, batch, scfg, ts, crop = c2_inputs()
    dtype = torch.float32
    bank = LoRABank(ucfg, sd["lora"], dtype, hip)
    pipe = TrainableSDPipeline(UNet(ucfg, sd["unet"], dtype, hip, bank), VAEDecoder(vcfg, sd["vae"], dtype, hip))
    dbank = LoRABank(ucfg, sd["d_lora"], dtype, hip)
    disc = D_sd(UNet(ucfg, sd["d_unet"], dtype, hip, dbank), dbank, sd["head_w"], sd["head_b"])
    trainer = CoMatTrainer(pipe, bank, Blip(bcfg, sd["blip"], dtype, hip), disc, scfg, seed=0)
    logs = trainer.train_step(batch, training_steps=ts, crop=crop)  # the gradients stay in the flat buffers after the update
    torch.cuda.synchronize()
    for key, gk in (("step_loss", "loss"), ("Blip", "blip_reward"), ("G_loss", "G_loss"), ("D_loss", "D_loss")):
        assert abs(float(logs[key]) - float(gold[gk])) < 2e-4 * max(abs(float(gold[gk])), 1e-3), (key, float(logs[key]), float(gold[gk]))

    def flat_error(bk, names, norms, projs, what):
        total_ref = float(np.sqrt((norms ** 2).sum()))
        err_sq = 0.0
        for i, n in enumerate(names):
            gr = bk.params[n].grad.detach().double().cpu().reshape(-1)
            nref = float(norms[i])
            assert abs(float(gr.norm()) - nref) <= 1e-3 * nref + 1e-6 * total_ref, f"{what} {n}: gradient norm"
            p = (rademacher(n, gr.numel()).double() @ gr).numpy()
            est = float(np.sqrt(np.mean((p - projs[i]) ** 2)))
            assert est <= 3e-3 * nref + 3e-6 * total_ref, f"{what} {n}: estimated gradient error {est:.3e} vs norm {nref:.3e}"
            err_sq += est ** 2
        return float(np.sqrt(err_sq)) / total_ref

    g_err = flat_error(bank, [str(n) for n in gold["names"]], gold["grad_norm"], gold["grad_proj"], "generator")
    d_err = flat_error(dbank, [str(n) for n in gold["d_names"]], gold["d_grad_norm"], gold["d_grad_proj"], "discriminator")
    path = os.environ.get("COMAT_TEST_REPORT")
    if path:
        with open(path, "a") as f:
            f.write(f"c2_full float32 cuda g_grad_rel_err={g_err:.3e} d_grad_rel_err={d_err:.3e}\n")
    # measured on an MI355X: generator 1.2e-5, discriminator 1.5e-6 (profiles/r05_f_fullsize_c2.txt) - bounded an order of
    # magnitude inside the north star's 1e-3, as the C1 check above
    assert g_err <= 1e-4 and d_err <= 1e-4, f"flat LoRA gradients: generator {g_err:.3e}, discriminator {d_err:.3e}"
    hg = trainer.D.head_grad.detach().double().cpu().numpy()
    assert np.abs(hg - gold["head_grad"]).max() <= 3e-3 * np.abs(gold["head_grad"]).max(), (hg, gold["head_grad"])


def test_c3_full_size_attribute_concentration_leg_matches_oracle_golden(hip):
    """The attribute-concentration leg at FULL size: concept matching + token-level and pixel-level concentration losses on the
    cross-attention maps captured at one of two trained denoise steps (mid_8, up_16, up_32, up_64), SD1.5 generator, fp32, against
    tests/golden/c3_full.npz = the CPU oracle on the same seeded world (tests/golden/make_c3_golden.py): loss terms and the LoRA
    gradient (per-tensor norms + 8 Rademacher projections each)."""
    from make_c1_golden import rademacher
    from make_c3_golden import c3_inputs

    from comat_amd.blip import Blip
    from comat_amd.pipeline import TrainableSDPipeline
    from comat_amd.step import CoMatTrainer
    from comat_amd.unet import LoRABank, UNet, VAEDecoder
    path = os.path.join(HERE, "golden", "c3_full.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/c3_full.npz not generated")
    gold = np.load(path)
    (ucfg, vcfg, bcfg), sd, batch, scfg, ts, crop, acs = c3_inputs()
    dtype = torch.float32
    bank = LoRABank(ucfg, sd["lora"], dtype, hip)
    pipe = TrainableSDPipeline(UNet(ucfg, sd["unet"], dtype, hip, bank), VAEDecoder(vcfg, sd["vae"], dtype, hip))
    trainer = CoMatTrainer(pipe, bank, Blip(bcfg, sd["blip"], dtype, hip), None, scfg, seed=0)
    bank.set_requires_grad(True)
    bank.zero_grad()
    out = trainer.compute_losses(batch, training_steps=ts, crop=crop, attrcon_steps=acs)
    out["loss"].backward()
    torch.cuda.synchronize()
    for key, gk in (("loss", "loss"), ("Blip", "blip_reward"), ("token_loss", "token_loss"), ("pixel_loss", "pixel_loss")):
        assert abs(float(out[key].detach()) - float(gold[gk])) < 2e-4 * max(abs(float(gold[gk])), 1e-3), (key, float(out[key].detach()), float(gold[gk]))
    names = [str(n) for n in gold["names"]]
    total_ref = float(np.sqrt((gold["grad_norm"] ** 2).sum()))
    err_sq = 0.0
    for i, n in enumerate(names):
        gr = bank.params[n].grad.detach().double().cpu().reshape(-1)
        nref = float(gold["grad_norm"][i])
        assert abs(float(gr.norm()) - nref) <= 1e-3 * nref + 1e-6 * total_ref, f"{n}: gradient norm"
        p = (rademacher(n, gr.numel()).double() @ gr).numpy()
        est = float(np.sqrt(np.mean((p - gold["grad_proj"][i]) ** 2)))
        assert est <= 3e-3 * nref + 3e-6 * total_ref, f"{n}: estimated gradient error {est:.3e} vs norm {nref:.3e}"
        err_sq += est ** 2
    rel = float(np.sqrt(err_sq)) / total_ref
    path = os.environ.get("COMAT_TEST_REPORT")
    if path:
        with open(path, "a") as f:
            f.write(f"c3_full float32 cuda grad_rel_err={rel:.3e}\n")
    # measured on an MI355X: 1.0e-5 (profiles/r05_f_fullsize_c2.txt)
    assert rel <= 1e-4, f"flat LoRA gradient: estimated rel. error {rel:.3e}"



def test_c4_full_size_sdxl_leg_matches_oracle_golden(hip):
    """The SDXL leg at FULL size: AttrConcenTrainableSDXLPipeline.forward with the real SDXL UNet layout (head
    dim 64, Linear proj_in / proj_out, 1 / 2 / 10-deep transformers, text_time conditioning), 512 x 512, N = 2 / K = 1 (the last step
    trained), concept matching + token-level and pixel-level concentration losses on the maps of mid_16 / up_16 / up_32, fp32, against
    tests/golden/c4_full.npz = the CPU oracle on the same seeded world (tests/golden/make_c4_golden.py): loss terms and the LoRA
    gradient (185.8 M values: per-tensor norms + 8 Rademacher projections each)."""
    from make_c1_golden import rademacher
    from make_c4_golden import c4_inputs

    from comat_amd.blip import Blip
    from comat_amd.pipeline import TrainableSDXLPipeline
    from comat_amd.step import CoMatTrainer
    from comat_amd.unet import LoRABank, UNet, VAEDecoder
    path = os.path.join(HERE, "golden", "c4_full.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/c4_full.npz not generated")
    gold = np.load(path)
    (ucfg, vcfg, bcfg), sd, batch, scfg, ts, crop, acs = c4_inputs()
    dtype = torch.float32
    bank = LoRABank(ucfg, sd["lora"], dtype, hip)
    pipe = TrainableSDXLPipeline(UNet(ucfg, sd["unet"], dtype, hip, bank), VAEDecoder(vcfg, sd["vae"], dtype, hip))
    trainer = CoMatTrainer(pipe, bank, Blip(bcfg, sd["blip"], dtype, hip), None, scfg, seed=0)
    bank.set_requires_grad(True)
    bank.zero_grad()
    out = trainer.compute_losses(batch, training_steps=ts, crop=crop, attrcon_steps=acs)
    out["loss"].backward()
    torch.cuda.synchronize()
    for key, gk in (("loss", "loss"), ("Blip", "blip_reward"), ("token_loss", "token_loss"), ("pixel_loss", "pixel_loss")):
        assert abs(float(out[key].detach()) - float(gold[gk])) < 2e-4 * max(abs(float(gold[gk])), 1e-3), (key, float(out[key].detach()), float(gold[gk]))
    names = [str(n) for n in gold["names"]]
    total_ref = float(np.sqrt((gold["grad_norm"] ** 2).sum()))
    err_sq = 0.0
    for i, n in enumerate(names):
        gr = bank.params[n].grad.detach().double().cpu().reshape(-1)
        nref = float(gold["grad_norm"][i])
        assert abs(float(gr.norm()) - nref) <= 1e-3 * nref + 1e-6 * total_ref, f"{n}: gradient norm"
        p = (rademacher(n, gr.numel()).double() @ gr).numpy()
        est = float(np.sqrt(np.mean((p - gold["grad_proj"][i]) ** 2)))
        assert est <= 3e-3 * nref + 3e-6 * total_ref, f"{n}: estimated gradient error {est:.3e} vs norm {nref:.3e}"
        err_sq += est ** 2
    rel = float(np.sqrt(err_sq)) / total_ref
    path = os.environ.get("COMAT_TEST_REPORT")
    if path:
        with open(path, "a") as f:
            f.write(f"c4_full float32 cuda grad_rel_err={rel:.3e}\n")
    assert rel <= 1e-4, f"flat LoRA gradient: estimated rel. error {rel:.3e}"
