"""Golden vectors for ONE whole optimisation step from the reference's OWN loop body, run in the build container.

    python tests/golden/make_step_body_golden.py      # writes tests/golden/step_body.npz

The statements of `Trainer.train`'s inner loop from `total_step = args.total_step` to the end of the discriminator update
(training_script.py:553-694) are taken from the file's syntax tree and executed as they are.  What they call is the reference's
own code as well - `TrainableSDPipeline.forward` (tests/golden/make_sampler_golden.py) and `D_sd.D_sd_pipeline_forward`
(make_gan_golden.py) - on stand-ins for the networks: a small differentiable generator "UNet" with one trainable matrix (its
"LoRA"), a 1x1 "VAE", a caption model whose reward is a fixed differentiable function of the cropped image, a discriminator
"UNet" with one trainable matrix plus the real 4 -> 1 head; `accelerate`'s object is a pass-through (backward, clip, gather);
the optimizers are `torch.optim.AdamW` with the hyper-parameters the reference's parser gives scripts/sd15.sh
(tests/golden/recipes.json).  What the vectors pin: the ORDER of a step (sampler -> reward on the crop -> generator-side GAN term
-> backward -> clip -> generator update, then the discriminator loss on the DETACHED latents of the same step -> backward -> clip
-> discriminator update), which parameters each update moves, and the logged scalars - through the (clipped) gradients left in
the parameters and the parameters after the step."""
import ast
import contextlib
import os
import random
import sys
import textwrap
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_gan_golden as gan  # noqa: E402
import make_sampler_golden as samp  # noqa: E402


def loop_body():
    src = open(os.path.join(REF, "training_script.py")).read()
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.For) and isinstance(node.iter, ast.Call) and "train_dataloader" in ast.get_source_segment(src, node.iter):
            stmts = [st for st in node.body if st.lineno >= 553 and st.end_lineno <= 694]
            assert (stmts[0].lineno, stmts[-1].end_lineno) == (553, 694), [(s.lineno, s.end_lineno) for s in stmts]
            code = "\n".join(textwrap.dedent(" " * st.col_offset + ast.get_source_segment(src, st)) for st in stmts)
            return compile(code, "training_script.py:553-694", "exec")
    raise SystemExit("loop not found")


def caption_reward(crop):
    """stand-in caption model: a differentiable function of the CROPPED image (so the crop offsets matter)"""
    ramp = torch.linspace(0.5, 1.5, crop.shape[-1]).reshape(1, 1, 1, -1) * torch.linspace(1.2, 0.8, crop.shape[-2]).reshape(1, 1, -1, 1)
    return -((crop * ramp) ** 2).mean(dim=(1, 2, 3))


def main():
    body = loop_body()
    forward = samp.reference_forward()
    d_forward, set_lora, get_gt = gan.reference_methods("D_sd_pipeline_forward", "set_D_sd_pipeline_lora", "get_D_gt_noise")
    recipes = __import__("json").load(open(os.path.join(HERE, "recipes.json")))["sd15"]
    g = torch.Generator().manual_seed(77)
    bs, h, w, L, C, N, K = 2, 28, 28, 6, 8, 5, 2      # resolution 224: offset range 1, crop size 223
    W0, V = torch.randn(4, 4, generator=g) * 0.6, torch.randn(3, 4, generator=g) * 0.5
    mix0 = torch.randn(4, 4, generator=g) * 0.7
    head_w0, head_b0 = torch.randn(1, 4, generator=g) * 0.8, torch.randn(1, generator=g) * 0.3
    lat0 = torch.randn(bs, 4, h, w, generator=g)
    noises = [torch.randn(bs, 4, h, w, generator=g) for _ in range(N)]
    cond, null, gan_null = (torch.randn(bs, L, C, generator=g) for _ in range(3))
    real = torch.randn(bs, 4, h, w, generator=g)
    up = torch.nn.Upsample(scale_factor=8, mode="nearest")

    args = types.SimpleNamespace(**recipes)
    args.total_step, args.K, args.resolution, args.train_batch_size = N, K, 8 * h, bs
    args.pretrain_model_name = "sd_1_5"   # no attribute concentration in this fixture (its pieces have their own)
    args.norm_grad = False

    class TrainableSDPipeline:      # the names the loop body tests its pipeline against
        pass

    class TrainableSDXLPipeline:
        pass
    Wp = nn.Parameter(W0.clone())
    pipe = TrainableSDPipeline()
    pipe._execution_device = torch.device("cpu")
    pipe.unet = types.SimpleNamespace(parameters=lambda: [Wp])
    pipe.text_encoder = None
    unet_calls = []

    def unet(x, t, encoder_hidden_states=None, cross_attention_kwargs=None, return_dict=False):
        unet_calls.append(int(t))
        return (samp.stub_unet(Wp, x, int(t), encoder_hidden_states),)
    pipe.unet = unet
    pipe.scheduler = samp.StubScheduler(noises)
    pipe.encode_prompt = lambda prompt, device, n, cfg, neg, prompt_embeds=None, negative_prompt_embeds=None, lora_scale=None: \
        (cond, negative_prompt_embeds)
    pipe.prepare_latents = lambda b, c, hh, ww, dtype, device, generator, latents: lat0
    pipe.prepare_extra_step_kwargs = lambda generator, eta: {}
    pipe.vae = types.SimpleNamespace(dtype=torch.float32, config=types.SimpleNamespace(scaling_factor=0.18215),
                                     decode=lambda z, return_dict=False: (up(torch.einsum("oc,bchw->bohw", V, z)),))
    pipe.forward = lambda **kw: forward(pipe, **kw)

    D = types.SimpleNamespace()
    D.unet = gan.StubUNet(mix0)
    D.mlp = nn.Sequential(nn.Linear(4, 1))
    with torch.no_grad():
        D.mlp[0].weight.copy_(head_w0)
        D.mlp[0].bias.copy_(head_b0)
    D.cls_loss_fn = nn.BCEWithLogitsLoss()
    D.D_args = types.SimpleNamespace(condition_discriminator=False, gan_unet_lastlayer_cls=False)
    D.ori_scheduler = gan.StubScheduler(lambda n: samp.O.DDPM().set_timesteps(n))
    D.weight_dtype = torch.float32
    D.D_parameters = [D.unet.mix] + list(D.mlp.parameters())
    D.set_D_sd_pipeline_lora = lambda requires_grad=True: set_lora(D, requires_grad=requires_grad)
    D.get_D_gt_noise = lambda device, **kw: get_gt(D, device, **kw)
    D.D_sd_pipeline_forward = lambda lat, side="G", **kw: d_forward(D, lat, side=side, **kw)

    order = []

    class Opt(torch.optim.AdamW):
        def __init__(self, name, *a, **k):
            super().__init__(*a, **k)
            self.tag = name

        def step(self, *a, **k):
            order.append(f"{self.tag}.step")
            return super().step(*a, **k)
    opt = Opt("G", [Wp], lr=args.learning_rate, betas=(args.adam_beta1, args.adam_beta2), weight_decay=args.adam_weight_decay,
              eps=args.adam_epsilon)
    opt_D = Opt("D", D.D_parameters, lr=args.learning_rate_D, betas=(args.adam_beta1_D, args.adam_beta2_D),
                weight_decay=args.adam_weight_decay, eps=args.adam_epsilon)

    def clip(params, max_norm):
        params = list(params)
        order.append(f"clip({'G' if params[0] is Wp else 'D'}, {max_norm})")
        return torch.nn.utils.clip_grad_norm_(params, max_norm)

    def backward(loss):
        order.append("backward")
        loss.backward()
    acc = types.SimpleNamespace(accumulate=lambda m: contextlib.nullcontext(), backward=backward, sync_gradients=True,
                                clip_grad_norm_=clip, gather=lambda x: x, device=torch.device("cpu"))
    crops = []

    def caption_model(image_crop, text, step=None, text_encoder=None, batch=None):
        crops.append(tuple(image_crop.shape))
        return {"total": caption_reward(image_crop.float()), "Blip": caption_reward(image_crop.float())}
    self = types.SimpleNamespace(accelerator=acc, pipeline=pipe, caption_model=caption_model, weight_dtype=torch.float32, D=D,
                                 optimizer=opt, D_optimizer=opt_D, G_parameters=[Wp], D_parameters=D.D_parameters,
                                 lr_scheduler=types.SimpleNamespace(step=lambda: order.append("lr.step"),
                                                                    get_last_lr=lambda: [args.learning_rate]))
    random.seed(5)
    ns = dict(self=self, args=args, random=random, torch=torch, batch={"text": ["p0", "p1"], "latents": real},
              null_embed=null, gan_null_embed=gan_null, gan_pooled_null_embed=None, step_count=0, train_loss=0.0,
              TrainableSDPipeline=TrainableSDPipeline, TrainableSDXLPipeline=TrainableSDXLPipeline)
    prev = torch.is_grad_enabled()
    exec(body, ns)
    torch.set_grad_enabled(prev)
    out = dict(W0=W0, V=V, mix0=mix0, head_w0=head_w0, head_b0=head_b0, latents=lat0, noises=torch.stack(noises), cond=cond, null=null,
               gan_null=gan_null, real=real, n_steps=np.int64(N), K=np.int64(K), resolution=np.int64(8 * h),
               training_steps=np.array(ns["training_steps"]), crop=np.array([ns["random_offset_x"], ns["random_offset_y"], ns["size"]]),
               gW=Wp.grad.clone(), gmix=D.unet.mix.grad.clone(), ghead_w=D.mlp[0].weight.grad.clone(), ghead_b=D.mlp[0].bias.grad.clone(),
               W1=Wp.detach().clone(), mix1=D.unet.mix.detach().clone(), head_w1=D.mlp[0].weight.detach().clone(),
               head_b1=D.mlp[0].bias.detach().clone(), order=np.array(order), unet_t=np.array(unet_calls),
               d_unet_t=np.array([c["t"] for c in D.unet.calls]), crop_shape=np.array(crops[0]),
               **{f"log:{k}": np.float64(v) for k, v in ns["logs"].items() if isinstance(v, (int, float))})
    np.savez_compressed(os.path.join(HERE, "step_body.npz"), **{k: (v.numpy() if torch.is_tensor(v) else v) for k, v in out.items()})
    print("order:", order)
    print("training_steps", ns["training_steps"], "crop", out["crop"], "logs", {k: round(float(v), 6) for k, v in ns["logs"].items() if isinstance(v, (int, float))})
    print("|dW|", float((Wp.detach() - W0).abs().max()), "|dmix|", float((D.unet.mix.detach() - mix0).abs().max()),
          "|dhead|", float((D.mlp[0].weight.detach() - head_w0).abs().max()))


if __name__ == "__main__":
    main()
