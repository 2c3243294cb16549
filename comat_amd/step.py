"""One CoMat optimisation step on MI355X — the counterpart of the step body `training_script.py:556-694`:

    sample K trained steps -> K-of-N differentiable denoise -> VAE decode -> random 510^2 crop -> BLIP reward
    [-> + w_g * G_loss] [-> + 1e-3 * token_loss + 5e-5 * pixel_loss] -> backward -> all-reduce(mean) of the flat LoRA
    gradient -> clip(0.1) + AdamW (one fused pass)  ||  D step: D_loss on [fake.detach(); real] -> backward ->
    all-reduce -> clip(1.0) + AdamW.

MI355X-first scheduling: the D step runs on its own HIP stream under the G backward chain; the G-gradient all-reduce
(RCCL over xGMI) is launched asynchronously as soon as the G backward is queued and overlaps the tail of the D step;
the per-step barrier of the reference (training_script.py:716) is dropped.  Optimizer state lives in flat fp32 buffers next to the flat parameter and
gradient buffers, so clip + AdamW is one HBM pass per buffer.
"""
from __future__ import annotations

import contextlib
import os
import random
import sys
from dataclasses import dataclass

import torch

from . import ops
from .blip import Blip
from .dist import GradReducer
from .gan import D_sd
from .losses import mask_loss
from .pipeline import TrainableSDPipeline
from .unet import LoRABank


@dataclass
class StepConfig:
    """Path-relevant flags of training_utils/arguments.py with the values of scripts/sd15.sh."""
    resolution: int = 512
    total_step: int = 50
    K: int = 5
    cfg_scale: float = 7.5
    gan_loss: bool = True
    gan_loss_weight: float = 1.0
    attrcon: bool = False
    attrcon_train_steps: int = 2
    train_layer_ls: tuple = ("mid_8", "up_16", "up_32", "up_64")
    attn_reses: tuple = (64, 32, 16, 8)
    mask_token_loss_weight: float = 1e-3
    mask_pixel_loss_weight: float = 5e-5
    lr: float = 5e-5
    lr_D: float = 2e-5          # --learning_rate_D 2e-5 (scripts/sd15.sh)
    adam_beta1: float = 0.9
    adam_beta2: float = 0.999
    adam_beta1_D: float = 0.0
    adam_beta2_D: float = 0.999
    adam_weight_decay: float = 1e-2
    adam_epsilon: float = 1e-8
    max_grad_norm: float = 0.1
    max_grad_norm_D: float = 1.0
    label_smoothing: float = 0.1

    @classmethod
    def sdxl(cls, **kw):
        """the values of scripts/sdxl.sh where they differ from scripts/sd15.sh (--learning_rate 2e-5 --learning_rate_D 5e-5
        --gan_loss_weight 0.5) and the SDXL layer list of training_script.py:312 (at 512 x 512)"""
        base = dict(lr=2e-5, lr_D=5e-5, gan_loss_weight=0.5, train_layer_ls=("mid_16", "up_16", "up_32"), attn_reses=(32, 16))
        base.update(kw)
        return cls(**base)


def _dbg(tag):
    """COMAT_DEBUG_SYNC=1: synchronise and print a phase marker (locates asynchronous device faults)."""
    if os.environ.get("COMAT_DEBUG_SYNC") in ("1", "phase"):
        torch.cuda.synchronize()
        print(f"[comat] phase ok: {tag}", file=sys.stderr, flush=True)


class FlatAdamW:
    """clip_grad_norm_ + AdamW over flat fp32 buffers (one or more segments sharing the global norm).

    The step count of the bias correction lives in device memory (`counters` int32 [2] = applied, skipped) and advances
    only when an update is applied: a non-finite gradient norm skips the update (the inf/NaN check of the reference's
    mixed-precision optimizer step, training_script.py:661-664) WITHOUT moving the bias correction ahead of the
    moments, the skip is visible to the caller (`counters[1]`, `gnorm_sq`), and a captured hipGraph of the whole step
    replays with the right count.  Deliberate deviation: plain torch AdamW would apply a NaN update."""

    def __init__(self, segments, lr, betas, eps, weight_decay, max_norm):
        self.segments = segments  # list of (param_flat, grad_flat)
        self.m = [torch.zeros_like(p) for p, _ in segments]
        self.v = [torch.zeros_like(p) for p, _ in segments]
        self.lr, self.betas, self.eps, self.wd, self.max_norm = lr, betas, eps, weight_decay, max_norm
        dev = segments[0][0].device
        self.gnorm_sq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.counters = torch.zeros(2, dtype=torch.int32, device=dev)

    @property
    def t(self):
        """number of applied updates (host read: synchronises; for logs and tests)"""
        return int(self.counters[0])

    def step(self, grad_scale=1.0):
        """grad_scale: 1 / world when the gradient buffers hold the SUM over data-parallel ranks (dist.GradReducer)"""
        k = ops.kernels()
        self.gnorm_sq.zero_()
        for _, g in self.segments:
            k.sumsq(g, g.numel(), self.gnorm_sq)
        for (p, g), m, v in zip(self.segments, self.m, self.v):
            k.adamw(p, g, m, v, p.numel(), self.lr, self.betas[0], self.betas[1], self.eps, self.wd, 0,
                    self.gnorm_sq, self.max_norm, step_dev=self.counters, grad_scale=grad_scale)
        k.adamw_tick(self.counters, self.gnorm_sq)


def sample_training_steps(total_step, K, rng: random.Random):
    """training_script.py:563-566"""
    interval = total_step // K
    max_start = total_step - interval * (K - 1) - 1
    start = rng.randint(0, max_start)
    return list(range(start, total_step, interval))


def sample_crop(resolution, rng: random.Random):
    """training_script.py:606-609: offsets in [0, resolution // 224], crop size resolution - offset_range."""
    offset_range = resolution // 224
    ox, oy = rng.randint(0, offset_range), rng.randint(0, offset_range)
    size = resolution - offset_range
    return (ox, oy, size, size)  # the reference slices dim 2 with x and dim 3 with y: (row0, col0, h, w)


@contextlib.contextmanager
def _own_streams_by_design():
    """The generator-side discriminator loss and the D step run on their own streams by design, so the discriminator head's
    AccumulateGrad nodes see gradients from more than one stream - which torch reports as a (once per process) warning about
    an unintended mismatch.  Switched off for the duration of THIS package's forward + backward only, and put back: user code
    around the step keeps torch's check (round-3 advice: it used to be switched off process-wide at import)."""
    get = getattr(torch._C, "_warn_on_accumulate_grad_stream_mismatch", None)
    put = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
    if get is None or put is None:  # older torch: no such check
        yield
        return
    before = get()
    put(False)
    try:
        yield
    finally:
        put(before)


class CoMatTrainer:
    def __init__(self, pipeline: TrainableSDPipeline, bank: LoRABank, blip: Blip, disc: D_sd | None,
                 cfg: StepConfig, seed=0):
        self.pipe, self.bank, self.blip, self.D, self.cfg = pipeline, bank, blip, disc, cfg
        self.opt = FlatAdamW([(bank.flat, bank.flat_grad)], cfg.lr, (cfg.adam_beta1, cfg.adam_beta2),
                             cfg.adam_epsilon, cfg.adam_weight_decay, cfg.max_grad_norm)
        self.opt_D = None
        if disc is not None:
            self.opt_D = FlatAdamW([(disc.bank.flat, disc.bank.flat_grad), (disc.head, disc.head_grad)], cfg.lr_D,
                                   (cfg.adam_beta1_D, cfg.adam_beta2_D), cfg.adam_epsilon, cfg.adam_weight_decay,
                                   cfg.max_grad_norm_D)
        self.rng = random.Random(seed)
        self.reducer = GradReducer()
        self.device = torch.device(pipeline.device)
        self._d_stream = None
        self._g_stream = None  # generator-side discriminator loss, next to VAE + BLIP (head_losses)
        self._d_pending = False
        self._d_keep = None
        self.serial_d = False  # GraphedStep: D step in stream order on the main stream
        self.flat_d = False    # GraphedStep: forked D stream, but no second-level fork for its weight gradients
        # hooks of segments.SegmentedStep: replay the head (VAE + BLIP + generator-side D loss) / the D step from graphs
        self.head_runner = None
        self.d_runner = None
        self._last_image_hw = None
        self.grad_scale = 1.0

    def drop_forked_streams(self):
        """After a FAILED graph capture: forget every stream this trainer forks from the capturing stream (the D step's,
        the generator-side D loss's).  The runtime may leave them in capture mode or invalidated; the next eager step
        makes fresh ones.  Per-stream workspaces keyed on the dead streams are dropped with them."""
        dead = [st.cuda_stream for st in (self._d_stream, self._g_stream) if st is not None]
        self._d_stream = self._g_stream = None
        self._d_pending, self._d_keep = False, None
        k = ops.kernels()
        ws = getattr(k, "_ws", None)
        if ws and dead:
            for key in [key for key in ws if isinstance(key, tuple) and any(h in key for h in dead)]:
                ws.pop(key, None)

    def head_losses(self, lat, batch, crop, bs, h, w):
        """final latents (channels-last tokens, fp32) -> VAE decode -> crop + BLIP caption reward [-> generator-side
        discriminator loss]: TrainableSDPipeline.py:219-223, training_script.py:606-623."""
        cfg = self.cfg
        # The generator-side discriminator loss needs the final latents and nothing of the decode / caption chain: on a
        # GPU it runs on its own stream next to VAE + BLIP (both chains are ~1 k latency-bound launches that leave most
        # of the chip idle), forward and - autograd keeps a node on the stream of its forward - backward.  Same kernels,
        # same bits (the two latent gradients are added, a + b == b + a); COMAT_G_STREAM=0 restores one stream.
        fork = (cfg.gan_loss and self.device.type == "cuda" and ops.side_streams_enabled() and not self.serial_d
                and os.environ.get("COMAT_G_STREAM", "1") != "0")
        G_loss = None
        if fork:
            main = torch.cuda.current_stream(self.device)
            if self._g_stream is None:
                self._g_stream = torch.cuda.Stream(device=self.device)
            self._g_stream.wait_stream(main)
            with torch.cuda.stream(self._g_stream):
                G_loss = self.D.D_sd_pipeline_forward(lat, "G", negative_prompt_embeds=batch["gan_null_embeds"],
                                                      num_inference_steps=cfg.total_step, h=h, w=w)
        img, H, W = self.pipe.decode_tokens(lat, bs, h, w, return_latents=True)
        _dbg("vae")
        self._last_image_hw = (H, W)
        reward, logp = self.blip.score(img, bs, H, W, batch["blip_input_ids"], batch["blip_attention_mask"], crop=crop,
                                       label_smoothing=cfg.label_smoothing)
        _dbg("blip")
        o = dict(reward=reward, logp=logp, image=(img, H, W))
        if fork:
            main.wait_stream(self._g_stream)
            o["G_loss"] = G_loss
        elif cfg.gan_loss:
            o["G_loss"] = self.D.D_sd_pipeline_forward(lat, "G", negative_prompt_embeds=batch["gan_null_embeds"],
                                                       num_inference_steps=cfg.total_step, h=h, w=w)
            _dbg("G loss")
        return o

    def compute_losses(self, batch, training_steps=None, crop=None, attrcon_steps=None):
        """Forward graph of the step up to the scalar loss.  batch keys: prompt_embeds, negative_prompt_embeds
        (bs,L,C); blip_input_ids, blip_attention_mask (bs,T); optional latents (bs,4,h,w), noises [N x (bs,4,h,w)],
        gan_null_embeds (bs,L,C_D), real_latents (bs,4,h,w), masks (list of [n_obj,H,W] bool arrays), attributes;
        SDXL pipelines also take pooled_prompt_embeds / negative_pooled_prompt_embeds (bs,1280) [+ add_time_ids]."""
        cfg = self.cfg
        res = cfg.resolution
        if training_steps is None:
            training_steps = sample_training_steps(cfg.total_step, cfg.K, self.rng)
        kw = {}
        if cfg.attrcon:
            if attrcon_steps is None:  # random.choices samples WITH replacement (training_script.py:590)
                attrcon_steps = self.rng.choices(training_steps, k=min(cfg.attrcon_train_steps, len(training_steps)))
            kw = dict(attrcon_train_steps=attrcon_steps, train_layer_ls=cfg.train_layer_ls, attn_reses=cfg.attn_reses)
        if "pooled_prompt_embeds" in batch:  # SDXL conditioning (TrainableSDPipeline.py:772-784)
            kw.update(pooled_prompt_embeds=batch["pooled_prompt_embeds"],
                      negative_pooled_prompt_embeds=batch["negative_pooled_prompt_embeds"],
                      add_time_ids=batch.get("add_time_ids"))
        lat = self.pipe.forward(
            batch["prompt_embeds"], batch["negative_prompt_embeds"], height=res, width=res,
            training_timesteps=training_steps, num_inference_steps=cfg.total_step, guidance_scale=cfg.cfg_scale,
            latents=batch.get("latents"), noises=batch.get("noises"), return_latents=True, output_type="latent_tokens",
            **kw)
        _dbg("sampler")
        bs = batch["prompt_embeds"].shape[0]
        if crop is None:
            crop = sample_crop(res, self.rng)
        h, w = res // 8, res // 8
        head = self.head_runner(lat, batch, crop, bs, h, w) if self.head_runner is not None else \
            self.head_losses(lat, batch, crop, bs, h, w)
        reward = head["reward"]
        out = dict(Blip=reward.detach(), token_logp=head["logp"], training_steps=training_steps, crop=crop)
        loss = -reward
        if cfg.gan_loss:
            loss = loss + cfg.gan_loss_weight * head["G_loss"]
            out["G_loss"] = head["G_loss"].detach()
        img, H, W = head["image"]
        if cfg.attrcon:
            tl, pl = mask_loss(self.pipe.attn_dict, batch["masks"], batch["attributes"], cfg.train_layer_ls, bs,
                               self.device)
            loss = loss + cfg.mask_token_loss_weight * tl + cfg.mask_pixel_loss_weight * pl
            out["token_loss"], out["pixel_loss"] = tl.detach(), pl.detach()
            self.pipe.attn_dict = {}
            _dbg("mask loss")
        out["loss"] = loss
        out["training_latents"] = lat
        out["image"] = (img, H, W)
        return out

    def _d_step(self, out, batch):
        if self.d_runner is not None:
            return self.d_runner(out, batch)
        return self._d_step_eager(out, batch)

    def _d_forward(self, out, batch):
        """forward half of the D step: zero the discriminator's gradients, D_loss on [fake.detach(); real] with its autograd
        graph (training_script.py:683-688)"""
        cfg = self.cfg
        h = w = cfg.resolution // 8
        self.D.zero_grad()
        real = ops.nchw_to_tokens(batch["real_latents"].to(self.device, torch.float32))
        return self.D.D_sd_pipeline_forward(out["training_latents"].detach(), "D",
                                            negative_prompt_embeds=batch["gan_null_embeds"],
                                            num_inference_steps=cfg.total_step, h=h, w=w, real_latents=real)

    def _d_step_eager(self, out, batch):
        """D forward + backward on [fake.detach(); real] (training_script.py:683-690)."""
        D_loss = self._d_forward(out, batch)
        D_loss.backward()
        return D_loss.detach()

    def _forward_backward(self, batch, fixed):
        with _own_streams_by_design():
            return self._forward_backward_impl(batch, fixed)

    def _forward_backward_impl(self, batch, fixed):
        """G forward + backward and the D forward + backward (everything of the step that precedes the exchange and
        the optimizer updates).  Returns a dict of device scalars (no host sync).

        The D step needs only the detached final latents, and it touches only the discriminator's gradient buffers:
        it is issued on its own HIP stream right after the G forward, so its ~1.2 k small kernels run concurrently
        with the G backward chain instead of after it (both are latency-bound at bs=1, neither fills the chip).  The
        results are bit-identical to the serial order (no atomics anywhere); COMAT_D_STREAM=0 restores it."""
        cfg = self.cfg
        ops.reset_side_stream_state()
        if self._d_pending and self.device.type == "cuda":
            # a previous step raised between the D fork and the join in _apply_updates: its D kernels may still read
            # buffers this step is about to reuse - join before anything else is queued
            torch.cuda.current_stream(self.device).wait_stream(self._d_stream)
            self._d_pending, self._d_keep = False, None
        self.bank.set_requires_grad(True)
        self.bank.zero_grad()
        out = self.compute_losses(batch, **fixed)
        logs = {k: v for k, v in out.items() if k in ("Blip", "G_loss", "token_loss", "pixel_loss")}
        logs["step_loss"] = out["loss"].detach()
        self._last = (out["training_steps"], out["crop"])
        concurrent = (cfg.gan_loss and self.device.type == "cuda" and ops.side_streams_enabled() and not self.serial_d
                      and os.environ.get("COMAT_D_STREAM", "1") != "0")
        if concurrent:
            main = torch.cuda.current_stream(self.device)
            if self._d_stream is None:
                self._d_stream = torch.cuda.Stream(device=self.device)
            self._d_stream.wait_stream(main)  # the G forward (latents, the discriminator's compute copies) is queued
            # the D stream reads the final latents: they stay referenced until that stream has been joined
            # (_apply_updates), so the allocator cannot hand their memory to main-stream work in the meantime
            self._d_keep = out["training_latents"]
            self._d_pending = True  # from here on the D stream holds work that must be joined, whatever happens below
            with torch.cuda.stream(self._d_stream):
                if self.flat_d:
                    with ops.no_side_streams():
                        logs["D_loss"] = self._d_step(out, batch)
                else:
                    logs["D_loss"] = self._d_step(out, batch)
        out["loss"].backward()  # LoRA weight gradients run on the side stream; joined at end of backward
        _dbg("G backward")
        self._d_pending = concurrent  # joined in _apply_updates, after the G all-reduce has been launched
        if not concurrent and cfg.gan_loss:
            logs["D_loss"] = self._d_step(out, batch)
        return logs

    def _forward_backward_joined(self, batch, fixed):
        """_forward_backward with every stream it forked joined again: the capturable part of a data-parallel step (the
        gradient exchange and the optimizer follow outside the graph, see GraphedStep)."""
        logs = self._forward_backward(batch, fixed)
        if self.device.type == "cuda":
            ops.join_side_streams()
            if self._d_pending:
                torch.cuda.current_stream(self.device).wait_stream(self._d_stream)
                self._d_pending = False
                self._d_keep = None
        return logs

    def _apply_updates(self):
        """all-reduce(mean) of the flat gradient buffers (RCCL, async) + clip + AdamW for G and D.  The G all-reduce is
        launched as soon as the G backward is queued, i.e. before the concurrently running D step is joined: on
        several GPUs it overlaps the tail of the D step; the D buffers follow once that stream has been joined."""
        if self.device.type == "cuda":
            ops.join_side_streams()  # idempotent; does not rely on the end-of-backward callback alone
        self.reducer.start(self.bank.flat_grad)
        if self._d_pending:
            torch.cuda.current_stream(self.device).wait_stream(self._d_stream)
            self._d_pending = False
            self._d_keep = None
        if self.cfg.gan_loss:
            self.reducer.start(self.D.bank.flat_grad, self.D.head_grad)
        scale = self.reducer.finish()  # 1 / world: the mean is taken inside the clip + AdamW pass
        # NOTE for readers of the buffers after this point: with more than one rank `flat_grad` / `head_grad` hold the SUM
        # over ranks and `opt.gnorm_sq` its squared norm; only comat_adamw applies `scale` (to the gradient and to the norm
        # it clips by).  The logs carry `grad_scale`: |mean gradient|^2 = grad_norm_sq * grad_scale^2.
        self.grad_scale = scale
        self.opt.step(scale)
        self.bank.mark_updated()
        if self.cfg.gan_loss:
            self.opt_D.step(scale)
            self.D.bank.mark_updated()
        ops.fp8_end_of_step()  # fp8 forward with delayed scaling: this step's abs-maxima become the next step's scales

    def fp8_calibrate(self, batch):
        """scales of the first step under delayed fp8 scaling (TrainableSDPipeline.fp8_calibrate) from this batch's prompt"""
        cfg = self.cfg
        kw = {}
        if "pooled_prompt_embeds" in batch:
            kw = dict(pooled_prompt_embeds=batch["pooled_prompt_embeds"],
                      negative_pooled_prompt_embeds=batch["negative_pooled_prompt_embeds"], add_time_ids=batch.get("add_time_ids"))
        return self.pipe.fp8_calibrate(batch["prompt_embeds"], batch["negative_prompt_embeds"], cfg.resolution, cfg.resolution,
                                       cfg.total_step, guidance_scale=cfg.cfg_scale, latents=batch.get("latents"),
                                       noises=batch.get("noises"), **kw)

    def train_step(self, batch, **fixed):
        """Full step: G forward/backward, D forward/backward, gradient exchange, G and D updates.  Returns a dict of
        detached device scalars (no host sync here) plus `training_steps` / `crop`.  The same schedule serves one GPU
        and data-parallel runs (the exchange is a no-op in a single-process run)."""
        logs = self._forward_backward(batch, fixed)
        self._apply_updates()
        logs["grad_norm_sq"] = self.opt.gnorm_sq  # non-finite => the generator update of this step was skipped
        logs["grad_scale"] = self.grad_scale      # 1 / world: grad_norm_sq is the norm of the SUM over ranks
        logs["training_steps"], logs["crop"] = self._last
        return logs


class GraphedStep:
    """The whole optimisation step as ONE hipGraph: G forward + backward, D forward + backward (on its own stream),
    LoRA weight gradients (side streams), gradient norms, clip + AdamW for G and D — ~17 k kernel launches that cost
    the host ~10 us each when issued one by one (the eager step is host-bound: bench `host_enqueue_ms_per_step` ~= the
    step time) replayed by the GPU's own command processor.

    What makes the step replayable:
      * every host value that changes between steps is a graph INPUT at a fixed device address: the batch tensors, the
        per-step noises, the crop (the crop + resize operator of the BLIP preprocessing is a pair of tap tables:
        `ResampleTables.static_copy / load`), and the AdamW step count (device counter, `FlatAdamW.counters`);
      * what does not change for a given list of trained denoise steps is baked in: timesteps (time-embedding
        projections, DDPM coefficients), the launch topology, every workspace.  One graph per distinct
        `training_steps` tuple (C2: N = K, a single tuple), captured lazily at its first use after one eager step;
      * no host synchronisation and no host-side data dependence inside the step (asserted by
        tests/test_step.py::test_step_is_enqueue_only).
    Streams inside the capture: the LoRA weight gradients of the G backward fork onto a side stream, the D step forks onto
    its own stream (its weight gradients stay on that stream: a fork from a forked stream crashes hipStreamEndCapture on
    ROCm 7.2 - located stage by stage in profiles/r02_c_stepgraph_stages.txt); both rejoin before the optimizer.
    Data-parallel runs (and COMAT_GRAPH_SPLIT=1): the graph ends where the streams have rejoined after the backward passes;
    the RCCL all-reduces and the two clip + AdamW updates (a dozen launches) follow eagerly, exactly as in the eager step -
    no collective is ever captured.
    Not captured (the eager path runs instead): attribute-concentration steps (their masks are resized on the host).
    Results are bit-identical to eager steps (`tests/test_step.py::test_graphed_step_matches_eager`)."""

    BATCH_KEYS = ("prompt_embeds", "negative_prompt_embeds", "gan_null_embeds", "latents", "real_latents",
                  "blip_input_ids", "blip_attention_mask", "pooled_prompt_embeds", "negative_pooled_prompt_embeds")

    def __init__(self, trainer: CoMatTrainer):
        self.tr = trainer
        self.graphs = {}
        self.static = None      # fixed-address copies of the batch
        self.static_key = None  # shapes / dtypes they were built for
        self.pool = None
        self.failed = None      # message of a failed capture: from then on every call is an eager step

    def supported(self, batch):
        return (self.tr.device.type == "cuda" and not self.tr.cfg.attrcon
                and batch.get("noises") is not None and batch.get("latents") is not None)

    @staticmethod
    def split():
        """graph = forward + backward only; exchange + optimizer eager (always so with more than one rank)"""
        from .dist import world_size
        return world_size() > 1 or os.environ.get("COMAT_GRAPH_SPLIT") == "1"

    def _stage(self, batch):
        """copy the batch into the fixed-address buffers (allocating them at the first call / on a shape change)"""
        dev = self.tr.device
        key = tuple((k, tuple(batch[k].shape), batch[k].dtype) for k in self.BATCH_KEYS if k in batch) + \
            (("noises", len(batch["noises"]), tuple(batch["noises"][0].shape)),)
        if self.static is None or key != self.static_key:
            self.static = {k: torch.empty_like(batch[k], device=dev) for k in self.BATCH_KEYS if k in batch}
            self.static["noises"] = [torch.empty_like(n, device=dev) for n in batch["noises"]]
            self.static_key = key
            self.graphs = {}
        for k in self.BATCH_KEYS:
            if k in batch:
                self.static[k].copy_(batch[k], non_blocking=True)
        for dst, src in zip(self.static["noises"], batch["noises"]):
            dst.copy_(src, non_blocking=True)
        for k in batch:  # host-side metadata (masks, token lists, time ids) passes through untouched
            if k not in self.BATCH_KEYS and k != "noises":
                if torch.is_tensor(batch[k]):
                    raise KeyError(f"batch tensor '{k}' has no fixed-address staging buffer (GraphedStep.BATCH_KEYS): a "
                                   "captured graph would keep reading the address it saw at capture time")
                self.static[k] = batch[k]
        return self.static

    def __call__(self, batch, training_steps=None, crop=None):
        tr = self.tr
        cfg = tr.cfg
        if self.failed is not None or not self.supported(batch):
            return tr.train_step(batch, **{k: v for k, v in (("training_steps", training_steps), ("crop", crop))
                                            if v is not None})
        if training_steps is None:
            training_steps = sample_training_steps(cfg.total_step, cfg.K, tr.rng)
        if crop is None:
            crop = sample_crop(cfg.resolution, tr.rng)
        sb = self._stage(batch)
        res = cfg.resolution
        split = self.split()
        # host-side values the capture bakes in are part of the key (SDXL: add_time_ids feed the added time embedding)
        meta = tuple(batch["add_time_ids"]) if batch.get("add_time_ids") is not None else None
        key = (tuple(training_steps), split, meta, os.environ.get("COMAT_GRAPH_D", "fork"))
        ent = self.graphs.get(key)
        # D step inside the capture: forked onto its own stream (it overlaps the G backward chain: 187 -> 168 ms per C2 step
        # on MI355X) with its weight gradients kept on that stream - a fork from a forked stream (nested) crashes
        # hipStreamEndCapture on ROCm 7.2.  COMAT_GRAPH_D=serial runs it in stream order on the main stream instead.
        saved = (tr.serial_d, tr.flat_d)
        if os.environ.get("COMAT_GRAPH_D", "fork") == "serial":
            tr.serial_d, tr.flat_d = True, False
        else:
            tr.serial_d, tr.flat_d = False, True
        try:
            if ent is None:
                # one eager step with these inputs first: fills every host-side memo (time embeddings, targets, crop
                # tables, workspaces of the default stream) and is a real optimisation step of its own.  The fixed-address
                # crop tables serve the eager step too.
                tr.blip.install_static_tables(res, res, crop)
                logs = tr.train_step(sb, training_steps=list(training_steps), crop=crop)
                tr.bank.mark_updated()
                if tr.D is not None:
                    tr.D.bank.mark_updated()
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                # more than one rank: RCCL's watchdog thread polls events while we capture - only THIS thread's calls may
                # invalidate the capture ("thread_local"; the default "global" mode would abort it)
                mode = {"capture_error_mode": "thread_local"} if split else {}
                # captured on the package's capture stream, whose workspaces (and its side stream's) exist and are zeroed
                # already: nothing a later graph on the same stream relies on is initialised by a node of this one
                cap = ops.capture_stream(tr.device)
                if tr._d_stream is not None:
                    ops.prepare_capture_stream(tr.device, tr._d_stream)
                try:
                    with ops.graph_capture(g, pool=self.pool, stream=cap, **mode):
                        if split:
                            out = tr._forward_backward_joined(sb, dict(training_steps=list(training_steps), crop=crop))
                        else:
                            out = tr.train_step(sb, training_steps=list(training_steps), crop=crop)
                except Exception as e:  # noqa: BLE001 - see `failed`
                    # A call of this object is ONE optimisation step with ONE gradient exchange, whatever happens: the eager
                    # step above was it (its all-reduces are matched on every rank), so a failed capture must not surface
                    # as an exception a caller would answer by stepping again.  From here on every call steps eagerly.
                    self.failed = f"{type(e).__name__}: {e}"
                    # (the fixed-address crop tables stay installed: segment graphs captured earlier read them, and
                    # Blip.tables() loads them with whatever crop an eager call asks for)
                    ops.reset_capture_stream(tr.device)
                    tr.drop_forked_streams()  # _d_stream, _g_stream: forked inside the capture, possibly left capturing
                    ops.drop_side_stream_state()  # weight gradients queued by the aborted capture
                    try:
                        torch.cuda.synchronize()
                    except Exception:  # noqa: BLE001 - the pending error of the failed capture
                        pass
                    return logs
                if self.pool is None:
                    self.pool = g.pool()
                self.graphs[key] = (g, out)
                # the capture did not execute anything: the eager step above is this call's step
                return logs
            g, out = ent
            tr.blip.tables(res, res, crop)  # loads this crop's operator into the fixed-address tables
            g.replay()
            out = dict(out)
            if split:
                tr._apply_updates()
                out["grad_norm_sq"] = tr.opt.gnorm_sq
            out["training_steps"], out["crop"] = list(training_steps), crop
            return out
        finally:
            tr.serial_d, tr.flat_d = saved
