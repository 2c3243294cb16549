"""Ground-truth latent producer for the GAN fidelity term: the counterpart of
tools/gan_gt_generate.py:171-193.  A frozen pipeline samples `num_inference_steps` DDPM steps with classifier-free
guidance under no_grad (the hipGraph-replayed UNet forward of this package) and returns the final latents; each
sample is stored as an fp32 (4, h, w) `.pt` tensor and indexed by one JSON line {"prompt", "file_path"} — the records
`Gan_Dataset.__getitem__` reads back as `batch['latents']` (training_utils/gan_dataset.py:56-74) and the step consumes
as `real_latents`.
"""
from __future__ import annotations

import json
import os
import uuid

import torch


def generate_gt_latents(pipeline, prompt_embeds, negative_prompt_embeds, height=512, width=512,
                        num_inference_steps=50, guidance_scale=7.5, generator=None, latents=None, noises=None, **kw):
    """One batch of final latents (bs, 4, height/8, width/8) fp32, as `pipeline(..., output_type='latent').images`
    of the reference.  `pipeline` should be built without a LoRA bank (or with frozen factors): nothing is trained."""
    with torch.no_grad():
        return pipeline.forward(prompt_embeds, negative_prompt_embeds, height=height, width=width,
                                training_timesteps=(), num_inference_steps=num_inference_steps,
                                guidance_scale=guidance_scale, generator=generator, latents=latents, noises=noises,
                                output_type="latent", **kw)


def write_gt_records(latents: torch.Tensor, prompts, save_dir: str, index_path: str):
    """gan_gt_generate.py:183-193: latents/{uid}.pt (fp32, CPU, one sample each) + appended jsonl index."""
    lat_dir = os.path.join(save_dir, "latents")
    os.makedirs(lat_dir, exist_ok=True)
    lines = []
    for i, prompt in enumerate(prompts):
        path = os.path.join(lat_dir, f"{uuid.uuid4().hex[:22]}.pt")
        torch.save(latents[i].detach().to("cpu", torch.float32).clone(), path)
        lines.append(json.dumps({"prompt": prompt, "file_path": path}))
    os.makedirs(os.path.dirname(os.path.abspath(index_path)), exist_ok=True)
    with open(index_path, "a") as f:
        f.write("\n".join(lines) + "\n")
    return [json.loads(line)["file_path"] for line in lines]


def read_gt_record(line: str):
    """What Gan_Dataset.__getitem__ extracts from one index line: {'text', 'latents'}."""
    ann = json.loads(line)
    return {"text": ann["prompt"], "latents": torch.load(ann["file_path"], map_location="cpu")}
