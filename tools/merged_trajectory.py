"""a short TRAINING TRAJECTORY from the standard LoRA initialisation (up factors = 0) with the trained calls on merged
weights W + s U D (COMAT_TRAIN_MERGED=1, the default since round 5) against the low-rank form (0): the early steps are where a
LoRA delta below half an ulp of W could vanish from the merged forward.  Full SD1.5 size, the C2 step (bf16), the same batch every
step; prints per step the loss terms and the norm of the generator's LoRA gradient, then the relative distance of the two runs'
LoRA parameters after the last step.
    python tools/merged_trajectory.py [steps]      (GPU box; runs both forms in this process, one after the other)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
os.environ["COMAT_STEP_GRAPH"] = "0"
import bench  # noqa: E402
from comat_amd import _hip, ops  # noqa: E402


def run(merged, steps, dev):
    ops.set_train_merged(merged)
    trainer, batch, fixed, scfg, _, _ = bench.build_world(dev, torch.bfloat16, 0, "c2")
    st = trainer.bank
    for name, p in st.params.items():  # the reference's LoRA init: down ~ N(0, 1 / r), up = 0 (training_utils/pipeline.py:84-115)
        if name.endswith("up.weight"):
            p.data.zero_()
    st.mark_updated()
    rows = []
    for i in range(steps):
        logs = trainer.train_step(batch, **fixed)
        torch.cuda.synchronize()
        rows.append((float(logs["step_loss"]), float(logs["Blip"]), float(logs["G_loss"]), float(logs["D_loss"]),
                     float(logs["grad_norm_sq"]) ** 0.5))
    flat = st.flat.detach().clone()
    del trainer, batch
    torch.cuda.empty_cache()
    return rows, flat


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda:0")
    ops.set_kernel_backend(_hip.HipKernels())
    out = {}
    for merged in (True, False):
        out[merged] = run(merged, steps, dev)
    print("# step : loss, BLIP reward, G loss, D loss, |LoRA grad|   merged (COMAT_TRAIN_MERGED=1) | low-rank (0)")
    for i in range(steps):
        a, b = out[True][0][i], out[False][0][i]
        print(f"{i:2d} : " + " ".join(f"{v:11.5f}" for v in a) + "  |  " + " ".join(f"{v:11.5f}" for v in b), flush=True)
    fa, fb = out[True][1], out[False][1]
    print(f"# LoRA parameters after {steps} steps: |merged - low-rank| / |low-rank| = {float((fa - fb).norm() / fb.norm()):.3e}; "
          f"|update| / |init| = {float((fb - 0).norm()):.3e} (norm of all factors)")


if __name__ == "__main__":
    main()
