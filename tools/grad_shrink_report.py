"""Where does the bf16 run of the full-size C1 step lose gradient norm?  (grad_norm_ratio 0.976 at 2.9e-2 error)

Runs config C1 at full size (SD1.5, 2 trained denoise steps, concept matching) in bf16 on the GPU and compares every LoRA
tensor's gradient with the fp32 CPU oracle's (tests/golden/c1_full.npz: per-tensor norm + 8 Rademacher projections),
grouped by UNet level, attention kind, projection and factor.  For every group: norm ratio bf16 / fp32, the estimated
relative error, and the COSINE between the two gradients estimated from the projections - a ratio below one with cosine
near one is a systematic attenuation, a ratio near sqrt(1 - err^2)-ish with lower cosine is noise.

    python tools/grad_shrink_report.py > gpurun_out/grad_shrink.txt
    python tools/grad_shrink_report.py --bisect >> gpurun_out/grad_shrink.txt

--bisect (round 4): the same step with ONE component at a time in fp32 storage (exact-f32 MFMA) and the others in
bf16 - UNet, VAE decoder, BLIP - and all three in fp32: which link of the chain loss -> BLIP -> resample -> VAE -> latents -> UNet
carries the uniform attenuation of the bf16 gradient (|g_bf16| / |g_fp32| = 0.9958 in every group)."""
import os
import sys
from collections import defaultdict

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from make_c1_golden import c1_inputs, rademacher  # noqa: E402

from comat_amd import _hip, ops  # noqa: E402
from comat_amd.blip import Blip  # noqa: E402
from comat_amd.pipeline import TrainableSDPipeline  # noqa: E402
from comat_amd.step import CoMatTrainer  # noqa: E402
from comat_amd.unet import LoRABank, UNet, VAEDecoder  # noqa: E402

dev = torch.device("cuda:0")
ops.set_kernel_backend(_hip.HipKernels())
gold = np.load(os.path.join(ROOT, "tests", "golden", "c1_full.npz"))
(ucfg, vcfg, bcfg), sd, batch, scfg, ts, crop = c1_inputs()
names = [str(n) for n in gold["names"]]
DT = {"b": torch.bfloat16, "f": torch.float32}


def run(mix):
    """mix = storage dtype of (UNet, VAE decoder, BLIP), 'b' or 'f' each -> (rows, loss)"""
    du, dv, db = (DT[c] for c in mix)
    bank = LoRABank(ucfg, sd["lora"], du, dev)
    pipe = TrainableSDPipeline(UNet(ucfg, sd["unet"], du, dev, bank), VAEDecoder(vcfg, sd["vae"], dv, dev))
    trainer = CoMatTrainer(pipe, bank, Blip(bcfg, sd["blip"], db, dev), None, scfg, seed=0)
    bank.set_requires_grad(True)
    bank.zero_grad()
    out = trainer.compute_losses(batch, training_steps=ts, crop=crop)
    out["loss"].backward()
    torch.cuda.synchronize()
    rows = []
    for i, n in enumerate(names):
        g = bank.params[n].grad.detach().double().cpu().reshape(-1)
        p = (rademacher(n, g.numel()).double() @ g).numpy()          # 8 projections of this run's gradient
        pr = gold["grad_proj"][i]                                     # ... of the fp32 oracle gradient
        rows.append((n, float(g.norm()), float(gold["grad_norm"][i]), p, pr))
    loss = float(out["loss"].detach())
    del trainer, pipe, bank, out
    torch.cuda.empty_cache()
    return rows, loss


if "--bisect" in sys.argv:
    print("# storage dtype of (UNet, VAE decoder, BLIP): b = bf16, f = fp32 (exact-f32 MFMA); against the fp32 CPU oracle")
    print("# mix   |g| / |g_fp32|   est. rel err   loss (fp32 oracle %.5f)" % float(gold["loss"]))
    for mix in ("bbb", "fbb", "bfb", "bbf", "bff", "fff"):
        rows, loss = run(mix)
        nb = np.sqrt(sum(r[1] ** 2 for r in rows))
        nr = np.sqrt(sum(r[2] ** 2 for r in rows))
        err = np.sqrt(sum(float(np.mean((r[3] - r[4]) ** 2)) for r in rows))
        print(f"  {mix}   {nb / nr:9.4f}   {err / nr:11.3e}   {loss:.5f}", flush=True)
    sys.exit(0)

rows, loss_b = run("bbb")
out = {"loss": loss_b}


def level(n):
    parts = n.split(".")
    if parts[0] == "mid_block":
        return "mid (8x8)"
    res = {"down_blocks": [64, 32, 16, 8], "up_blocks": [8, 16, 32, 64]}[parts[0]][int(parts[1])]
    return f"{parts[0].split('_')[0]} {res}x{res}"


def summarise(key_fn, title):
    grp = defaultdict(list)
    for r in rows:
        grp[key_fn(r[0])].append(r)
    print(f"\n# by {title}:  group  tensors  |g_fp32|  ratio |g_bf16|/|g_fp32|  est. rel err  est. cosine")
    for k in sorted(grp):
        rs = grp[k]
        nb = np.sqrt(sum(r[1] ** 2 for r in rs))
        nr = np.sqrt(sum(r[2] ** 2 for r in rs))
        # E[p_a p_b] over Rademacher vectors = <a, b>: 8 samples per tensor, summed over the group's tensors
        dot = sum(float(np.mean(r[3] * r[4])) for r in rs)
        err = np.sqrt(sum(float(np.mean((r[3] - r[4]) ** 2)) for r in rs))
        print(f"  {k:28s} {len(rs):4d}  {nr:10.3e}  {nb / nr:7.4f}  {err / nr:9.3e}  {dot / (nb * nr):7.4f}")


tot_b = np.sqrt(sum(r[1] ** 2 for r in rows))
tot_r = np.sqrt(sum(r[2] ** 2 for r in rows))
print(f"C1 full size, bf16 vs fp32 oracle: |g_bf16| / |g_fp32| = {tot_b / tot_r:.4f} over {len(rows)} LoRA tensors; "
      f"loss bf16 {float(out['loss']):.5f} vs fp32 {float(gold['loss']):.5f}")
summarise(level, "UNet level")
summarise(lambda n: "attn1 (self)" if ".attn1." in n else "attn2 (cross)", "attention kind")
summarise(lambda n: n.split(".lora.")[0].split(".")[-1].replace("0", "to_out") + "." + n.split(".lora.")[1].split(".")[0], "projection.factor")
summarise(lambda n: level(n) + (" attn1" if ".attn1." in n else " attn2"), "level x attention kind")
