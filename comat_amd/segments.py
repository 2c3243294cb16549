"""Step segments: the heavy, fixed-topology pieces of one CoMat optimisation step as replayable hipGraphs.

A whole-step graph (step.GraphedStep) needs a static launch topology: it serves config C2 (every denoise step trained, no
attribute concentration) on one GPU and nothing else.  The configurations the reference actually trains
(`scripts/sd15.sh`, `scripts/sdxl.sh`: 50 denoise steps of which 5 sampled ones carry gradient,
`training_script.py:563-566`; attribute concentration on 2 steps drawn from those, `:589-590`) change WHICH steps are
trained and WHERE the attention maps are captured from one optimisation step to the next, and launched eagerly they are
host-bound (~10 us of Python per launch x 17 k launches).  What does not change is the topology of the pieces:

  U   one trained UNet call (`unet(latent_model_input, t, encoder_hidden_states=...)` with gradient,
      TrainableSDPipeline.py:138-150; with or without captured cross-attention maps,
      AttrConcenTrainableSDPipeline.py:239-279) - forward graph + backward graph per (slot, variant); the timestep
      enters as a device tensor (its sinusoid), so ONE pair serves every timestep;
  H   the head: VAE decode -> crop + BLIP caption reward -> generator-side discriminator loss
      (TrainableSDPipeline.py:219-223, training_script.py:606-623) - forward + backward graph;
  D   the discriminator step (forward + backward on [fake.detach(); real], training_script.py:683-690) - one graph,
      captured on and replayed on the discriminator's own stream (it overlaps the generator's backward);
  the N-K no-grad UNet forwards keep their per-timestep graphs (unet.GraphedUNetForward).

Each U / H piece is a `GraphedSegment`: a `torch.autograd.Function` whose forward copies its inputs into fixed-address
buffers and replays the forward graph, and whose backward copies the incoming gradients and replays the backward graph
(LoRA weight gradients accumulate into the flat gradient buffer as a side effect of that replay, exactly as they do in
the eager backward).  The sampler loop, the loss assembly, the gradient exchange and the optimizer stay the eager code of
step.CoMatTrainer / pipeline.TrainableSDPipeline - a few hundred launches instead of 17 k - so ONE code path serves
every configuration and any number of ranks: no collective is ever captured, and a data-parallel run replays exactly
what a single GPU replays.

First use of a segment runs it eagerly (a real part of that step: it also fills every host-side memo) and captures it
right afterwards; from the second use on it is replayed.  Replays are bit-identical to the eager launches
(tests/test_segments.py).
"""
from __future__ import annotations

import os

import torch
from torch.autograd import Function

from . import ops


def _copy_into(dst, src):
    """dst <- src (fixed-address buffer); float tensors through the library's copy kernel, the rest through torch"""
    if dst.data_ptr() == src.data_ptr():
        return
    with torch.no_grad():
        if dst.dtype in (torch.float32, torch.bfloat16) and src.dtype in (torch.float32, torch.bfloat16) \
                and src.is_contiguous() and src.device == dst.device:
            ops.kernels().unary(ops.UN_COPY, src, dst, dst.numel())
        else:
            dst.copy_(src, non_blocking=True)


def _capture_kwargs():
    # more than one rank: RCCL's watchdog thread polls events while we capture - only THIS thread's calls may invalidate
    # the capture ("thread_local"; the default "global" mode would abort it)
    import torch.distributed as dist
    return {"capture_error_mode": "thread_local"} if dist.is_available() and dist.is_initialized() else {}


class GraphedSegment:
    """fn(*tensors) -> tuple of tensors, captured as a forward hipGraph and (if any output requires grad) a backward
    hipGraph.  `stream`: the stream the graphs are captured on (it selects the kernels' workspaces: graphs that may be
    replayed CONCURRENTLY must be captured on different streams) - default: the package's capture stream.
    `pool`: memory pool shared with other segments that are never live at the same time (the two variants of one slot)."""

    def __init__(self, fn, name, stream=None, pool=None):
        self.fn, self.name, self.stream, self.pool = fn, name, stream, pool
        self.fg = self.bg = None
        self.side_out = None
        self.replays = 0
        self.timing = None  # [] -> (kind, start event, end event) per replay (SegmentedStep(timing=True))

    @property
    def captured(self):
        return self.fg is not None

    def capture(self, inputs, bwd_side=None):
        """bwd_side = (fn, stream): fn() is captured INSIDE the backward graph on `stream`, forked at its beginning and
        joined at its end (work that is independent of this backward and should overlap it: the discriminator step);
        its result is kept in `self.side_out`.  `stream` needs its own workspaces (ops.prepare_capture_stream)."""
        dev = inputs[0].device
        st = self.stream or ops.capture_stream(dev)
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()
        # fixed-address inputs live OUTSIDE the graph pool (they survive pool reuse by a sibling segment)
        self.si = [torch.empty_like(x).requires_grad_(x.requires_grad) for x in inputs]
        for s, x in zip(self.si, inputs):
            _copy_into(s, x.detach())
        kw = _capture_kwargs()
        self.fg = torch.cuda.CUDAGraph()
        with ops.graph_capture(self.fg, pool=self.pool, stream=st, **kw):
            outs = self.fn(*self.si)
        outs = tuple(outs) if isinstance(outs, (tuple, list)) else (outs,)
        self.out_req = [bool(o.requires_grad) for o in outs]
        self.sgo = [torch.zeros_like(o) if r else None for o, r in zip(outs, self.out_req)]
        self.sgi = [None] * len(self.si)
        if any(self.out_req):
            self.bg = torch.cuda.CUDAGraph()
            with ops.graph_capture(self.bg, pool=self.pool, stream=st, **kw):
                if bwd_side is not None:
                    side_fn, side_st = bwd_side
                    side_st.wait_stream(torch.cuda.current_stream(dev))
                    with torch.cuda.stream(side_st):
                        self.side_out = side_fn()
                torch.autograd.backward([o for o, r in zip(outs, self.out_req) if r],
                                        [g for g in self.sgo if g is not None])
                if bwd_side is not None:
                    torch.cuda.current_stream(dev).wait_stream(side_st)
            self.sgi = [x.grad for x in self.si]
            for x in self.si:
                x.grad = None
        # keep the outputs' storage, drop the recorded autograd graph: the saved activations stay where the graphs
        # expect them, but the pool may hand their addresses to a sibling capture (never live at the same time)
        self.outs = [o.detach() for o in outs]
        del outs
        # the LoRA factors are not inputs of the replaying autograd node (their gradients are a side effect of the backward
        # graph), so an input-less dependence is needed for its outputs to require grad when no INPUT does (the first
        # trained denoise step; every SDXL step: the UNet input is detached there) - the usual dummy-leaf anchor
        self.anchor = torch.zeros((), device=dev, requires_grad=True)

    def _replay(self, g, kind):
        if self.timing is None:
            g.replay()
            return
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        e.record()
        self.timing.append((kind, s, e))

    def __call__(self, *inputs):
        return _Replay.apply(self, self.anchor, *inputs)


class _Replay(Function):
    @staticmethod
    def forward(ctx, seg, anchor, *inputs):
        for s, x in zip(seg.si, inputs):
            _copy_into(s, x)
        seg._replay(seg.fg, "fwd")
        seg.replays += 1
        ctx.seg = seg
        outs = tuple(o.detach() for o in seg.outs)
        ctx.mark_non_differentiable(*[o for o, r in zip(outs, seg.out_req) if not r])
        ctx.set_materialize_grads(False)
        return outs

    @staticmethod
    def backward(ctx, *gos):
        seg = ctx.seg
        for sg, g in zip(seg.sgo, gos):
            if sg is None:
                continue
            if g is None:
                sg.zero_()
            else:
                _copy_into(sg, g if g.is_contiguous() else g.contiguous())
        seg._replay(seg.bg, "bwd")
        return (None, None) + tuple(None if gi is None else gi.detach() for gi in seg.sgi)


class SegmentedStep:
    """`trainer.train_step` with the UNet calls, the head and the discriminator step replayed from segment graphs.

        stepper = SegmentedStep(trainer);  logs = stepper(batch, training_steps=..., crop=..., attrcon_steps=...)

    Same call convention and same results (bit for bit) as CoMatTrainer.train_step; any configuration, any number of
    ranks.  Without a GPU it simply runs the eager step."""

    FLOAT_KEYS = ("prompt_embeds", "negative_prompt_embeds", "gan_null_embeds", "latents", "real_latents",
                  "pooled_prompt_embeds", "negative_pooled_prompt_embeds")
    INT_KEYS = ("blip_input_ids", "blip_attention_mask")

    def __init__(self, trainer, use_head=True, use_d="head", dry=False):
        """use_d: "head" = the discriminator step is captured inside the head's backward graph (on its own stream, it
        overlaps the generator's backward), "own" = its own graph on the discriminator's stream, False = eager.
        dry: run every segment's function eagerly through the same hooks, never capture (checks the host logic of the
        hooks where there is no GPU: tests/test_segments.py)"""
        self.tr = trainer
        self.dry = dry
        self.unet_segs, self.slot_pools = {}, {}
        self.head_seg = self.d_seg = None
        self.use_head, self.use_d = use_head, (use_d if use_head or use_d != "head" else "own")
        self.static = {}
        self.enabled = trainer.device.type == "cuda"
        self.failed = None  # message of a failed capture: from then on every call (and the rest of that step) runs eagerly

    def _capture(self, seg, inputs, **kw):
        """seg.capture(...) -> True; a capture that raises is turned into a STATE, as step.GraphedStep does: the eager run
        that preceded it was this step's real work, so the step goes on eagerly (its gradient exchange stays matched on
        every rank) and the segment is never registered - a half-built one (fg = None, no `si`) would be taken for
        replayable by the next call.  The capture stream, the streams forked from it and the weight gradients the aborted
        pass queued are dropped."""
        try:
            seg.capture(inputs, **kw)
            return True
        except Exception as e:  # noqa: BLE001 - see `failed`
            self.failed = f"{seg.name}: {type(e).__name__}: {e}"
            import sys
            # said once, loudly: from here on every step of this trainer launches eagerly (~10 us of host time per kernel)
            print(f"[comat_amd] capture of segment '{seg.name}' failed ({type(e).__name__}: {e}); this SegmentedStep runs "
                  "eagerly from now on", file=sys.stderr, flush=True)
            dev = self.tr.device
            ops.reset_capture_stream(dev)
            self.tr.drop_forked_streams()
            ops.drop_side_stream_state()
            try:
                torch.cuda.synchronize()
            except Exception:  # noqa: BLE001 - the pending error of the failed capture
                pass
            return False

    # ---- batch staging: every tensor of the batch at a fixed device address ------------------------------------------
    def _stage(self, batch):
        dev = self.tr.device
        out = dict(batch)
        for k in self.FLOAT_KEYS + self.INT_KEYS:
            if k not in batch:
                continue
            src = batch[k]
            dst = self.static.get(k)
            if dst is None or dst.shape != src.shape or dst.dtype != src.dtype:
                dst = self.static[k] = torch.empty_like(src, device=dev)
            dst.copy_(src, non_blocking=True)
            out[k] = dst
        if batch.get("noises") is not None:
            ns = self.static.get("noises")
            if ns is None or len(ns) != len(batch["noises"]) or ns[0].shape != batch["noises"][0].shape:
                ns = self.static["noises"] = [torch.empty_like(n, device=dev) for n in batch["noises"]]
            for d, s in zip(ns, batch["noises"]):
                d.copy_(s, non_blocking=True)
            out["noises"] = ns
        return out

    # ---- U: trained UNet call -----------------------------------------------------------------------------------------
    def _run_unet(self, slot, xin, B, H, W, t, ctx, L, cap, added, wanted=None):
        """the pipeline's trained-call hook (pipeline.TrainableSDPipeline.trained_runner).  `wanted`: the (place,
        resolution) pairs the attribute-concentration loss will read (train_layer_ls): only those maps leave the
        segment - an unused map output would cost a zero-filled gradient buffer in every backward replay."""
        unet = self.tr.pipe.unet
        cap = tuple(cap)
        wkey = None if wanted is None else tuple(sorted(wanted))
        key = (slot, cap, wkey, B, H, W, L, bool(xin.requires_grad))
        te = unet.time_sinusoid(t, B)

        def fn(x, te_, ctx_, *rest):
            eps, maps = unet(x, B, H, W, te_, ctx_, L, capture_places=cap, added=rest[0] if rest else None)
            keep = {place: [p for p in maps[place]
                            if wanted is None or (place, int(round(p.shape[2] ** 0.5))) in wanted] for place in cap}
            self._map_counts[key] = [len(keep[place]) for place in cap]
            return (eps,) + tuple(p for place in cap for p in keep[place])

        self._map_counts = getattr(self, "_map_counts", {})
        inputs = (xin, te, ctx) + ((added,) if added is not None else ())
        eager = self.dry or self.failed is not None
        seg = None if eager else self.unet_segs.get(key)
        if eager:
            outs = fn(*inputs)
        elif seg is None:
            pool = self.slot_pools.get(slot)
            if pool is None:
                pool = self.slot_pools[slot] = torch.cuda.graph_pool_handle()
            seg = GraphedSegment(fn, f"unet slot {slot} capture={cap}", pool=pool)
            outs = fn(*inputs)          # eager: this call's real work (and the memo / workspace warm-up of the capture)
            if self._capture(seg, inputs):
                self.unet_segs[key] = seg
        else:
            outs = seg(*inputs)
        eps, flat = outs[0], list(outs[1:])
        maps, i = {}, 0
        for place, n in zip(cap, self._map_counts[key]):
            maps[place] = flat[i:i + n]
            i += n
        return eps, maps

    # ---- H: VAE decode + BLIP reward + generator-side discriminator loss ------------------------------------------------
    def _run_head(self, lat, batch, crop, bs, h, w):
        tr = self.tr
        cfg = tr.cfg
        res = cfg.resolution
        tr.blip.tables(res, res, crop)  # loads this crop's operator into the fixed-address tables the graph reads

        def fn(lat_, ids, mask, *null):
            b = dict(batch, blip_input_ids=ids, blip_attention_mask=mask)
            if null:
                b["gan_null_embeds"] = null[0]
            o = tr.head_losses(lat_, b, crop, bs, h, w)
            return (o["reward"], o["logp"]) + ((o["G_loss"],) if cfg.gan_loss else ()) + (o["image"][0],)

        inputs = (lat, batch["blip_input_ids"], batch["blip_attention_mask"]) + \
            ((batch["gan_null_embeds"],) if cfg.gan_loss else ())
        d_in_head = self.use_d == "head" and cfg.gan_loss and tr.D is not None
        if d_in_head:
            inputs = inputs + (batch["real_latents"],)  # staged with the head's inputs, read by the D branch
        if self.dry or self.failed is not None:
            outs = fn(*inputs)
            self._img_hw = tr._last_image_hw
        elif self.head_seg is None:
            tr.blip.install_static_tables(res, res, crop)
            seg = GraphedSegment(fn, "head")
            outs = fn(*inputs)
            self._img_hw = tr._last_image_hw
            side = None
            if d_in_head:
                # The discriminator step reads the (detached) final latents and nothing of the generator's backward: it is
                # captured INSIDE the head's backward graph on the discriminator's stream and overlaps it (two graphs
                # launched on two streams do not overlap on this runtime: measured 176 vs 153 ms per C2 step).  A fork
                # from a forked stream crashes hipStreamEndCapture on ROCm 7.2, so its weight gradients stay on that stream.
                # (Round 5 also split the step - forward half inside the head's FORWARD graph, backward half here: head forward +
                # backward 36.9 -> 33.7 ms per C3 step, ~1 ms of the step - but hipStreamEndCapture of a forward graph with two
                # forked streams segfaults when the capturing process already holds the C2 graphs, as the default bench line's
                # does (profiles/r05_y_bench_default_dsplit_crash.txt).  Branch exp/d-split.)
                if tr._d_stream is None:
                    tr._d_stream = torch.cuda.Stream(device=tr.device)
                ops.prepare_capture_stream(tr.device, tr._d_stream)

                def d_side():
                    with ops.no_side_streams():
                        return tr._d_step_eager(dict(training_latents=seg.si[0].detach()),
                                                dict(batch, real_latents=seg.si[-1], gan_null_embeds=seg.si[3]))
                side = (d_side, tr._d_stream)
                # the capture must find every host-side memo of the discriminator's D-side call filled (its time embedding,
                # targets): run that step once eagerly now.  It leaves nothing behind - the step's real D step, which
                # follows in this same optimisation step, starts by zeroing the discriminator's gradients.
                cur = torch.cuda.current_stream(tr.device)
                tr._d_stream.wait_stream(cur)
                with torch.cuda.stream(tr._d_stream), ops.no_side_streams():
                    tr._d_step_eager(dict(training_latents=lat.detach()), batch)
                cur.wait_stream(tr._d_stream)
            if self._capture(seg, inputs, bwd_side=side):
                self.head_seg = seg
        else:
            outs = self.head_seg(*inputs)
        o = dict(reward=outs[0], logp=outs[1], image=(outs[-1],) + tuple(self._img_hw))
        if cfg.gan_loss:
            o["G_loss"] = outs[2]
        return o

    # ---- D: discriminator step ---------------------------------------------------------------------------------------------
    def _run_d(self, out, batch):
        tr = self.tr

        def fn(lat_, real, null):
            return tr._d_step_eager(dict(training_latents=lat_), dict(batch, real_latents=real, gan_null_embeds=null))

        dev = tr.device
        real = batch["real_latents"]
        inputs = (out["training_latents"].detach(), real, batch["gan_null_embeds"])
        if self.dry or self.failed is not None:
            return fn(*inputs)
        if self.use_d == "head":
            # replayed inside the head's backward graph; until that graph exists (first step) the step runs eagerly here
            if self.head_seg is not None and self.head_seg.replays > 0 and self.head_seg.side_out is not None:
                return self.head_seg.side_out  # fixed-address scalar, written when the backward graph replays
            return fn(*inputs)
        if self.d_seg is None:
            st = torch.cuda.current_stream(dev)  # the discriminator's stream (step.CoMatTrainer forks it) or the main one
            if st.cuda_stream != torch.cuda.default_stream(dev).cuda_stream:
                ops.prepare_capture_stream(dev, st)
                seg = GraphedSegment(fn, "discriminator step", stream=st)
            else:
                seg = GraphedSegment(fn, "discriminator step")
            loss = fn(*inputs)
            if self._capture(seg, inputs):
                self.d_seg = seg
            return loss
        return self.d_seg(*inputs)[0]

    # ---- the step --------------------------------------------------------------------------------------------------------------
    def __call__(self, batch, training_steps=None, crop=None, attrcon_steps=None):
        tr = self.tr
        fixed = {k: v for k, v in (("training_steps", training_steps), ("crop", crop), ("attrcon_steps", attrcon_steps))
                 if v is not None}
        if (not self.enabled and not self.dry) or self.failed is not None:
            return tr.train_step(batch, **fixed)
        sb = self._stage(batch)
        # the derived LoRA copies are refreshed HERE, eagerly: inside a capture the refresh would be baked into that one
        # graph (and marked done without having run)
        tr.bank.ensure_compute_copy()
        if tr.D is not None:
            tr.D.bank.ensure_compute_copy()
        tr.pipe.trained_runner = self._run_unet
        tr.head_runner = self._run_head if self.use_head else None
        tr.d_runner = self._run_d if (self.use_d and tr.D is not None) else None
        try:
            return tr.train_step(sb, **fixed)
        finally:
            tr.pipe.trained_runner = None
            tr.head_runner = tr.d_runner = None

    def _all(self):
        return [("unet", s) for s in self.unet_segs.values()] + \
            [(n, s) for n, s in (("head", self.head_seg), ("disc", self.d_seg)) if s is not None]

    def set_timing(self, flag=True):
        """HIP events around every segment replay from now on (they cost a few microseconds each: diagnostics only)"""
        for _, s in self._all():
            s.timing = [] if flag else None

    def timing_summary(self, steps):
        """-> {"unet fwd": ms per step, ...} from the replays since set_timing(); synchronises"""
        torch.cuda.synchronize()
        acc = {}
        for name, s in self._all():
            for kind, a, b in s.timing or ():
                d = acc.setdefault(f"{name} {kind}", [0.0, 0])
                d[0] += a.elapsed_time(b)
                d[1] += 1
        return {k: {"ms_per_step": round(v[0] / steps, 2), "replays_per_step": round(v[1] / steps, 2)} for k, v in sorted(acc.items())}

    def stats(self):
        segs = [s for _, s in self._all()]
        return {"segments": len(segs), "replays": sum(s.replays for s in segs)}
