"""bench.py — CoMat train-step throughput on MI355X (BASELINE.json metric: train-step images/sec, SD1.5 512^2, bs=1/GPU).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Workload: SD1.5 UNet (859.5 M params, LoRA r=128 on all 32 attentions), 512x512
(latent 64x64), 1 prompt per GPU, N=5 denoise steps all 5 with gradient, CFG 7.5, VAE decode, 510^2 crop ->
BLIP-large caption CE (concept matching) + GAN fidelity loss (second SD1.5 UNet discriminator, G side and D side),
backward, all-reduce(mean) of the flat LoRA gradients, clip + AdamW for G and D.  bf16 storage / fp32 accumulate.
Synthetic, seeded weights and inputs (no checkpoints or tokenizer offline).  One "step" = one such optimisation step;
value = total images/s over all ranks (weak scaling: one prompt per GPU).

Extra objects on the JSON line:
  roofline     — the dominant kernel family (MFMA implicit-GEMM conv / GEMM), algorithmic FLOPs of its launches in one
                 step divided by their HIP-event time (one extra instrumented step, events on the launch stream);
                 peak = 2.5 PFLOP/s dense bf16 (MI355X_MICROARCH.md).  `step_frac` = whole-step algorithmic FLOPs
                 (27.7 TFLOP) / step time / peak.
  cpu_baseline — the CPU oracle (oracle/, kind "port") timed on this box's host cores on a bounded sample and
                 extrapolated by algorithmic FLOPs; rank 0, N=1 only.
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # before the HIP runtime initialises (see comat_amd/__init__.py)

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # dense MFMA bf16, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_FP8_TFLOPS = 5000.0  # dense MFMA fp8 (e4m3), same guide: the peak an fp8 kernel family is priced against
# algorithmic TFLOP per step per prompt: UNet fwd 0.839 (incl. LoRA), VAE 2.515, BLIP 0.408
F_UNET, F_VAE, F_BLIP = 0.839, 2.515, 0.408


F_UNET_SDXL = 1.678  # SDXL UNet @64x64 latent incl. LoRA r=128, per sample forward


def step_tflop(total_step, K, gan, sdxl=False, res=512, bs=1):
    """Algorithmic TFLOP of one step (SURVEY.md 8d counting convention).  The discriminator is the SD1.5 UNet in every
    configuration (scripts of the reference).  The per-call figures are quoted at 512^2 (64^2 latents); at 1024^2 the
    UNets and the VAE see 4x the positions (self-attention grows 16x: undercounted here), BLIP still sees 384^2."""
    px = (res / 512.0) ** 2
    fu = (F_UNET_SDXL if sdxl else F_UNET) * px
    nograd = (total_step - K) * 2 * fu
    train = K * 2 * fu * 2
    f = nograd + train + F_VAE * px * 2 + F_BLIP * 2
    if gan:
        f += F_UNET * px * 2 + 2 * F_UNET * px * 2
    return f * bs  # every term is per prompt


MFMA_CALLS = ("gemm", "gemm_segments", "conv2d", "flash_attn_fwd", "flash_attn_bwd", "gemm_tt_grouped")
NOT_LAUNCHES = ("tt_group_ok", "gemm_workspace_bytes", "prepare_stream")  # helpers of the backend that enqueue nothing


class CallRecorder:
    """Wraps the kernel backend for ONE eager, single-stream step after the timed region.  Two measurements per MFMA
    problem of the step (GEMM / K-segmented GEMM / conv / fused attention):
      in step  - every launch is bracketed by HIP events on the launch stream (torch's current stream, the one the C ABI
                 enqueues on).  With one stream the GPU is the slower side (a serialised step is ~250 ms of kernels
                 against ~180 ms of host enqueue), so an event pair spans the kernel plus at most a launch gap; operands
                 are as warm or cold as they are in a real step (frozen weights come from HBM).  `roofline.achieved`
                 uses THIS time; it agrees with the rocprofv3 durations of the same kernels (profiles/).
      replayed - each distinct problem (entry point, shape, strides, epilogue) again on its own operands, back to back
                 from a hipGraph of `n` launches: kernel time with warm caches and no gaps, the kernel's own ceiling."""

    def __init__(self, inner):
        self.inner = inner
        self.calls = {}
        self.other = {}
        self.events = []
        self.nulls = []
        self.null_us = 0.0

    def __getattr__(self, name):
        fn = getattr(self.inner, name)
        if not callable(fn):
            return fn
        if name in NOT_LAUNCHES:
            return fn
        if name not in MFMA_CALLS:
            def counted(*a, **kw):
                self.other[name] = self.other.get(name, 0) + 1
                return fn(*a, **kw)
            return counted

        def wrapped(*a, **kw):
            sig = self._sig(name, a, kw)
            rec = self.calls.get(sig)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = fn(*a, **kw)
            e.record()
            self.events.append((sig, s, e))
            if len(self.events) % 16 == 0:  # an EMPTY bracket under the same queue conditions: what the events themselves cost
                n0, n1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n0.record()
                n1.record()
                self.nulls.append((n0, n1))
            if rec is None:
                from comat_amd import _hip
                kid = _hip.last_gemm_kernel() if name in ("gemm", "gemm_segments", "conv2d", "gemm_tt_grouped") else -1
                self.calls[sig] = [name, a, kw, self._flops(name, a, kw), 1, kid, self._bytes(name, a, kw)]
            else:
                rec[4] += 1
            return r
        return wrapped

    @staticmethod
    def _epi(kw):
        return (f"bias={int(kw.get('bias') is not None)}{int(kw.get('bias2') is not None)} R={int(kw.get('R') is not None)} "
                f"act={kw.get('act', 0)}")

    @classmethod
    def _sig(cls, name, a, kw):
        if name == "gemm_tt_grouped":  # a group of k-major weight-gradient problems: shapes with their multiplicities
            shapes = {}
            for p in a[0]:
                shapes[p[3:6]] = shapes.get(p[3:6], 0) + 1
            return f"tt_grouped n={len(a[0])} " + " ".join(f"{m}x{n}x{k}*{c}" for (m, n, k), c in sorted(shapes.items()))
        if name == "gemm":
            out = a[2] if a[2] is not None else kw["geglu"][0]  # fused GEGLU without stored pre-activations: C is absent
            odt = out.dtype if out is not None else "e4m3"   # ... and with the e4m3 output only (q8) there is no bf16 product either
            return (f"gemm M={a[3]} N={a[4]} K={a[5]} ld={a[6]},{a[7]},{a[8]} tA={int(kw.get('transA', False))} "
                    f"tB={int(kw.get('transB', False))} b={kw.get('batch', (1, 1))} in={a[0].dtype} out={odt} "
                    + cls._epi(kw) + (f" geglu={1 if kw['geglu'][1] else 2}" if kw.get("geglu") is not None else "")
                    + (" q8" if kw.get("q8") is not None else "") + (f" ktail={kw['ktail'][2]}" if kw.get("ktail") is not None else ""))
        if name == "gemm_segments":
            return (f"gemm_segments M={a[2]} N={a[3]} K={'+'.join(str(sg[2]) for sg in a[0])} b={kw.get('batch', 1)} "
                    f"in={a[0][0][0].dtype} out={a[1].dtype} " + cls._epi(kw))
        if name == "conv2d":
            return (f"conv B={a[3]} HWin={a[4]}x{a[5]} Cin={a[6]} HWout={a[7]}x{a[8]} Cout={a[9]} k={a[10]} s={a[12]} "
                    f"mode={kw.get('mode', 0)} ups={kw.get('ups', 1)} in={a[0].dtype} out={a[2].dtype} " + cls._epi(kw))
        if name == "flash_attn_fwd":
            return f"flash_fwd B={a[5]} H={a[6]} Nq={a[7]} Nk={a[8]} d={a[9]} {a[0].dtype}"
        return f"flash_bwd B={a[10]} H={a[11]} Nq={a[12]} Nk={a[13]} d={a[14]} {a[0].dtype}"

    @staticmethod
    def _flops(name, a, kw):
        if name == "gemm_tt_grouped":
            return sum(2.0 * p[3] * p[4] * p[5] for p in a[0])
        if name == "gemm":
            M, N, K = a[3], a[4], a[5]
            b = kw.get("batch", (1, 1))
            return 2.0 * M * N * (K + (kw["ktail"][2] if kw.get("ktail") is not None else 0)) * b[0] * b[1]
        if name == "gemm_segments":
            return 2.0 * a[2] * a[3] * sum(sg[2] for sg in a[0]) * kw.get("batch", 1)
        if name == "conv2d":
            B, Cin, Hout, Wout, Cout, KH, KW, stride = a[3], a[6], a[7], a[8], a[9], a[10], a[11], a[12]
            f = 2.0 * B * Hout * Wout * Cout * KH * KW * Cin
            return f / (stride * stride) if kw.get("mode", 0) == 1 else f
        if name == "flash_attn_fwd":  # (q, k, v, o, lse, B, H, Nq, Nk, d, ...): QK^T + PV
            return 4.0 * a[5] * a[6] * a[7] * a[8] * a[9]
        return 10.0 * a[10] * a[11] * a[12] * a[13] * a[14]  # flash_attn_bwd: 5 products

    @staticmethod
    def _bytes(name, a, kw):
        """algorithmic HBM bytes of one launch: every operand read once, the output written once"""
        sz = lambda t: t.element_size()
        if name == "gemm_tt_grouped":  # both operands once, the fp32 output read + written
            return sum((p[3] + p[4]) * p[5] * 2 + 2 * p[3] * p[4] * 4 for p in a[0])
        if name == "gemm":
            M, N, K = a[3], a[4], a[5]
            b = kw.get("batch", (1, 1))
            nb = b[0] * b[1]
            r = M * N * sz(kw["R"]) if kw.get("R") is not None else 0
            if kw.get("geglu") is not None:  # + the [M, N / 2] product; the pre-activations only when they are stored
                y, keep = kw["geglu"]
                if isinstance(y, str) or y is None:  # "bwd" form: (pre, "bwd") / e4m3 output only (q8): no bf16 product is written
                    y = None
                out = (M * (N // 2) * sz(y) if y is not None else 0) + (M * (N // 2) if kw.get("q8") is not None else 0)
                return (M * K + N * K) * sz(a[0]) + (M * N * 2 if keep else 0) + out
            kt = kw.get("ktail")  # bf16 k-tail of an fp8 product: its two operands once
            tail = (M + N) * kt[2] * 2 * nb if kt is not None else 0
            return nb * ((M * K + N * K) * sz(a[0]) + M * N * sz(a[2]) + r) + tail
        if name == "gemm_segments":
            M, N, nb = a[2], a[3], kw.get("batch", 1)
            r = M * N * sz(kw["R"]) if kw.get("R") is not None else 0
            return nb * (sum((M + N) * sg[2] for sg in a[0]) * sz(a[0][0][0]) + M * N * sz(a[1]) + r)
        if name == "conv2d":
            B, Hin, Win, Cin, Hout, Wout, Cout, KH, KW = a[3:12]
            r = B * Hout * Wout * Cout * sz(kw["R"]) if kw.get("R") is not None else 0
            return (B * Hin * Win * Cin + Cout * KH * KW * Cin) * sz(a[0]) + B * Hout * Wout * Cout * sz(a[2]) + r
        return 0

    def measure(self, n=10):
        """-> {family: [replayed seconds per step, flops per step, launches per step, bytes per step, in-step seconds]};
        family = the kernel that served the problem"""
        from comat_amd import _hip
        torch.cuda.synchronize()
        # an event pair costs time of its own (two timestamp commands the GPU executes in stream order): measured by the
        # empty brackets recorded in the same step and subtracted from every bracket (call 11: 25.2 us per raw bracket of
        # the pipelined kernel against 20.7 us in the rocprofv3 trace of the same launches)
        nulls = sorted(n0.elapsed_time(n1) for n0, n1 in self.nulls)
        self.null_us = nulls[len(nulls) // 2] * 1e3 if nulls else 0.0
        in_step, self.raw_in_step = {}, {}
        for sig, s0, e0 in self.events:
            raw = s0.elapsed_time(e0) * 1e-3
            self.raw_in_step[sig] = self.raw_in_step.get(sig, 0.0) + raw
            in_step[sig] = in_step.get(sig, 0.0) + max(raw - self.null_us * 1e-6, 0.0)
        self.events, self.nulls = [], []
        side = torch.cuda.Stream()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fam, rows = {}, []
        for sig, (name, a, kw, fl, count, kid, nbytes) in self.calls.items():
            fn = getattr(self.inner, name)
            with torch.cuda.stream(side):  # workspaces of this stream exist before the capture
                fn(*a, **kw)
                fn(*a, **kw)
            side.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                for _ in range(n):
                    fn(*a, **kw)
            g.replay()
            torch.cuda.synchronize()
            s.record()
            g.replay()
            e.record()
            torch.cuda.synchronize()
            t = s.elapsed_time(e) * 1e-3 / n
            del g
            family = _hip.GEMM_KERNEL_NAMES[kid] if kid >= 0 else ("flash_fwd_kernel" if name == "flash_attn_fwd"
                                                                   else "flash_bwd (dq + dkdv kernels)")
            d = fam.setdefault(family, [0.0, 0.0, 0, 0.0, 0.0])
            d[0] += t * count
            d[1] += fl * count
            d[2] += count
            d[3] += nbytes * count
            d[4] += in_step[sig]
            rows.append((in_step[sig] * 1e3, count, fl * count / in_step[sig] / 1e12, in_step[sig] / count * 1e6, t * 1e6,
                         family.split(" ")[0], sig))
        dump = os.environ.get("COMAT_BENCH_DUMP")
        if dump:
            with open(dump, "w") as f:
                f.write("# in-step ms/step  calls  in-step TF/s  in-step us/launch  replayed us/launch  kernel  problem\n")
                for ms, cnt, tf, us, us_r, kname, sig in sorted(rows, reverse=True):
                    f.write(f"{ms:9.3f} {cnt:5d} {tf:8.1f} {us:9.2f} {us_r:9.2f}  {kname:16s} {sig}\n")
                f.write(f"# other launches per step: {sum(self.other.values())} "
                        f"{dict(sorted(self.other.items(), key=lambda kv: -kv[1]))}\n")
        return fam


def build_world(device, dtype, rank, cfg_name, bs=1):
    from comat_amd import config, weights
    from comat_amd.blip import Blip
    from comat_amd.gan import D_sd
    from comat_amd.pipeline import TrainableSDPipeline
    from comat_amd.step import CoMatTrainer, StepConfig
    from comat_amd.unet import LoRABank, UNet, VAEDecoder

    ucfg, vcfg, bcfg = config.SD15_UNET, config.SD15_VAE, config.BLIP_LARGE
    tiny = cfg_name == "selftest"
    if tiny:
        ucfg, vcfg, bcfg = config.TINY_UNET, config.TINY_VAE, config.TINY_BLIP
    sdxl = cfg_name in ("c4", "c5")
    res = 1024 if cfg_name == "c5" else 512
    if sdxl:  # BASELINE config C4: SDXL generator at 512^2 (64^2 latents), SD1.5 discriminator, full CoMat losses;
        # C5: the same at 1024^2 (128^2 latents) with the generator UNet's forward on the fp8 MFMA, bf16 backward
        from comat_amd.pipeline import TrainableSDXLPipeline
        ucfg, vcfg = config.SDXL_UNET, config.SDXL_VAE
        if cfg_name == "c5":
            scfg = StepConfig.sdxl(resolution=1024, total_step=50, K=5, gan_loss=True, attrcon=True,
                                   train_layer_ls=("mid_32", "up_32", "up_64"), attn_reses=(64, 32))
        else:
            scfg = StepConfig.sdxl(resolution=512, total_step=50, K=5, gan_loss=True, attrcon=True)
    elif tiny:
        scfg = StepConfig(resolution=64, total_step=3, K=2, gan_loss=True, attrcon=False)
    elif cfg_name == "c2":
        scfg = StepConfig(resolution=512, total_step=5, K=5, gan_loss=True, attrcon=False)
    elif cfg_name == "c3":
        scfg = StepConfig(resolution=512, total_step=50, K=5, gan_loss=True,
                          attrcon=os.environ.get("COMAT_C3_ATTRCON", "1") != "0")
    else:
        raise ValueError(cfg_name)
    t0 = time.time()
    usd = weights.make_unet_weights(ucfg, seed=1234)
    lsd = weights.make_lora_weights(ucfg, seed=4321)
    bank = LoRABank(ucfg, lsd, dtype, device)
    if cfg_name == "c5":  # activation scales of the fp8 forward: "delayed" (round 6, default) or "jit" (rounds 2-5) for A/B runs
        from comat_amd import ops as _ops
        _ops.set_fp8_scaling(os.environ.get("COMAT_FP8_SCALING", "delayed"))
    unet = UNet(ucfg, usd, dtype, device, bank, fp8_forward=cfg_name == "c5")
    keep_for_cpu = usd if (rank == 0) else None
    vae = VAEDecoder(vcfg, weights.make_vae_weights(vcfg, seed=2345), dtype, device)
    blip = Blip(bcfg, weights.make_blip_weights(bcfg, seed=3456), dtype, device)
    dcfg = config.TINY_UNET if tiny else config.SD15_UNET  # the discriminator is the SD1.5 UNet in every configuration
    dsd = weights.make_unet_weights(dcfg, seed=1235)
    dbank = LoRABank(dcfg, weights.make_lora_weights(dcfg, seed=4322), dtype, device)
    g = torch.Generator().manual_seed(99)
    disc = D_sd(UNet(dcfg, dsd, dtype, device, dbank), dbank, torch.randn(4, generator=g) * 0.5,
                torch.randn(1, generator=g) * 0.1)
    del dsd
    pipe = TrainableSDXLPipeline(unet, vae) if sdxl else TrainableSDPipeline(unet, vae)
    trainer = CoMatTrainer(pipe, bank, blip, disc, scfg, seed=rank)
    if scfg.total_step > scfg.K and os.environ.get("COMAT_PRECAPTURE", "1") != "0" and torch.device(device).type == "cuda":
        trainer.pipe.prepare_graphs(bs, scfg.resolution, scfg.resolution, 77, scfg.total_step)
    # synthetic batch; per-rank seeds differ (each rank has its own prompt)
    g = torch.Generator().manual_seed(1000 + rank)
    L, T = (7, 9) if tiny else (77, 16)
    hl = scfg.resolution // 8
    if tiny:
        ids = torch.randint(1, bcfg.vocab_size, (1, T), generator=g)
    else:
        ids = torch.cat([torch.tensor([101, 1037, 5855, 1997]), torch.randint(1000, 30522, (11,), generator=g),
                         torch.tensor([102])]).reshape(1, T)
    batch = dict(
        prompt_embeds=torch.randn(bs, L, ucfg.cross_attention_dim, generator=g),
        negative_prompt_embeds=torch.randn(bs, L, ucfg.cross_attention_dim, generator=g),
        gan_null_embeds=torch.randn(1, L, dcfg.cross_attention_dim, generator=g).expand(bs, -1, -1).contiguous(),
        latents=torch.randn(bs, 4, hl, hl, generator=torch.Generator().manual_seed(42 + rank)),
        noises=[torch.randn(bs, 4, hl, hl, generator=torch.Generator().manual_seed(100 + i)).to(device)
                for i in range(scfg.total_step)],
        real_latents=torch.randn(bs, 4, hl, hl, generator=torch.Generator().manual_seed(7)) * (0.2 / 0.18215),
        blip_input_ids=ids.expand(bs, -1).contiguous(), blip_attention_mask=torch.ones_like(ids).expand(bs, -1).contiguous())
    if sdxl:
        batch.update(pooled_prompt_embeds=torch.randn(bs, ucfg.pooled_dim, generator=g),
                     negative_pooled_prompt_embeds=torch.randn(bs, ucfg.pooled_dim, generator=g),
                     add_time_ids=(res, res, 0, 0, res, res))
    if scfg.attrcon:
        import numpy as np
        m = np.zeros((2, res, res), dtype=bool)
        m[0, 60 * res // 512:250 * res // 512, 40 * res // 512:230 * res // 512] = True
        m[1, 280 * res // 512:480 * res // 512, 260 * res // 512:500 * res // 512] = True
        batch["masks"] = [m] * bs
        batch["attributes"] = [[[2, 3], [6, 7]]] * bs
    # inputs resident in HBM before anything is timed (the bench contract): the step's staging copies are then device-to-device
    # and asynchronous, instead of fourteen pageable host-to-device copies that each wait for the stream to drain
    batch = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
    fixed = dict(crop=(0, 0, 64, 64) if tiny else (1, 1, res - 2, res - 2))
    if cfg_name == "c5":  # delayed fp8 scaling: the first step's scales from one eager no-grad sampler pass (untimed set-up)
        trainer.fp8_calibrate(batch)
    if cfg_name == "c2":
        fixed["training_steps"] = [0, 1, 2, 3, 4]
    return trainer, batch, fixed, scfg, keep_for_cpu, time.time() - t0


def cpu_baseline(usd, scfg):
    """The CPU oracle (fp32 port of the reference's path: its own Python needs diffusers / torchvision / weights that
    are absent) on a BOUNDED sample of the same workload, per SURVEY.md 8d: one UNet call without grad and one with
    grad (LoRA + input gradients) at the workload's own size (CFG batch 2, 64x64 latents, 77 text tokens), one BLIP
    reward forward + backward at 510^2 -> 384^2, one VAE decode forward + backward on a 32x32 latent (a quarter of the
    pixels; x4).  Each component: 2 warm-ups (1 for the trained UNet call) + 3 timed repetitions, median, on min(host threads, 64) threads (see below).
    The step time is the medians combined by
    the step's call counts (the discriminator is the same UNet: G side batch 1 = half a trained call, D side batch 2 = one
    trained call)."""
    from comat_amd import config, weights
    from oracle import blip as OB
    from oracle import sd as O
    import dataclasses
    cores = os.cpu_count() or 1

    def timed(fn, reps=3, warm=2):
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(reps):
            t0 = time.time()
            fn()
            ts.append(time.time() - t0)
        return sorted(ts)[len(ts) // 2], ts

    ocfg = O.UNetConfig(**dataclasses.asdict(config.SD15_UNET))
    g = torch.Generator().manual_seed(0)
    x, ctx = torch.randn(2, 4, 64, 64, generator=g), torch.randn(2, 77, 768, generator=g)
    lora = {k: v.clone().requires_grad_(True) for k, v in weights.make_lora_weights(config.SD15_UNET, seed=4321).items()}
    t, reps_all = {}, {}

    def nograd():
        with torch.no_grad():
            O.unet_forward(usd, ocfg, x, 801, ctx, None, None)

    # Threads: at most 64.  On the 256-thread host of an MI355X box the same call takes 158.8 s with all 256 threads against
    # 4.9 s with 64 (measured in round 3, profiles/r03_z_bench_default.log: oversubscribed fork-join of many small ops) -
    # timing that again in every run would cost ten minutes for a number nobody would quote; COMAT_CPU_THREADS overrides.
    threads = int(os.environ.get("COMAT_CPU_THREADS", min(cores, 64)))
    torch.set_num_threads(threads)
    by_threads = {threads: timed(nograd)}
    t["unet_nograd"], reps_all["unet_nograd"] = by_threads[threads]

    def train():
        for p in lora.values():
            p.grad = None
        xg = x.clone().requires_grad_(True)
        O.unet_forward(usd, ocfg, xg, 801, ctx, lora, None).float().square().mean().backward()
    t["unet_train"], reps_all["unet_train"] = timed(train, warm=1)  # (round 3 ran it cold: 13.3 / 10.7 / 10.2 s)
    del lora
    vcfg = O.VAEConfig(**dataclasses.asdict(config.SD15_VAE))
    vsd = weights.make_vae_weights(config.SD15_VAE, seed=2345)
    z0 = torch.randn(1, 4, 32, 32, generator=g)

    def vae():
        z = z0.clone().requires_grad_(True)
        O.vae_decode(vsd, vcfg, z).square().mean().backward()
    t["vae_quarter_train"], reps_all["vae_quarter_train"] = timed(vae)
    del vsd
    bcfg = OB.BlipConfig(**dataclasses.asdict(config.BLIP_LARGE))
    bsd = weights.make_blip_weights(config.BLIP_LARGE, seed=3456)
    img0 = torch.rand(1, 3, 510, 510, generator=g)
    ids = torch.cat([torch.tensor([101, 1037, 5855, 1997]), torch.randint(1000, 30522, (11,), generator=g),
                     torch.tensor([102])]).reshape(1, 16)

    def blip():
        img = img0.clone().requires_grad_(True)
        reward, _ = OB.score(bsd, bcfg, img, ids, torch.ones_like(ids), label_smoothing=0.1)
        reward.backward()
    t["blip_train"], reps_all["blip_train"] = timed(blip)
    del bsd
    step_s = (scfg.K * t["unet_train"] + (scfg.total_step - scfg.K) * t["unet_nograd"] + 4 * t["vae_quarter_train"]
              + t["blip_train"] + (1.5 * t["unet_train"] if scfg.gan_loss else 0.0))
    return {"value": 1.0 / step_s, "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": f"oracle/ (CPU fp32) components, median of 3 repetitions after two warm-ups (one for unet_train), on {threads} of {cores} host "
                      "threads (all 256 threads of an MI355X host measured 32x slower in round 3): " + ", ".join(f"{k} {v:.1f} s" for k, v in t.items())
                      + f"; step = {scfg.K} x unet_train + {scfg.total_step - scfg.K} x unet_nograd + 4 x vae_quarter_train + "
                      f"blip_train + 1.5 x unet_train (discriminator G and D sides) = {step_s:.0f} s",
            "components_s": {k: round(v, 2) for k, v in t.items()},
            "repetitions_s": {k: [round(x_, 2) for x_ in v] for k, v in reps_all.items()}}


def load_pmc_summary():
    """HBM traffic / MFMA-busy figures of the dominant kernel from the newest committed counter pass (separate rocprofv3
    --pmc runs, converted by tools/pmc_to_json.py - counters cannot be collected inside a timed run).  The pass records the
    build id of the library it profiled (comat_build_id(): a hash of the kernel sources, plan table and build flags); the
    figures are quoted only when that is the library loaded NOW, otherwise traffic and mfma_busy_frac are null and the note
    says why."""
    import glob
    from comat_amd import _hip
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_kernels.json")))
    if not files:
        return None
    with open(files[-1]) as f:
        d = json.load(f)
    out = dict(d.get("bench_roofline") or {})
    have, want = d.get("build_id"), _hip.build_id()
    if have != want:
        out["traffic_bytes_per_launch"] = out["mfma_busy_frac"] = None
        out["note"] = (f"{os.path.basename(files[-1])} was measured on build {have or '(not recorded: before round 4)'}, the loaded "
                       f"library is build {want}: counter figures withheld (re-run tools/calls/r5_final.sh pmc)")
    else:
        out["note"] = f"{out.get('note', '')} [{os.path.basename(files[-1])}, build {have}]".strip()
    return out


def attn_map_probe():
    """North-star counter: the attention-map path on its own.  Softmax write-back of one captured SD1.5 `up_64`
    cross-attention map ([8 heads, 4096 pixels, 77 tokens], fp32 scores -> bf16 probabilities, the map the attribute-
    concentration loss reads) and the gather of 4 token columns over it, each timed with HIP events on the launch
    stream over 20 launches; algorithmic bytes / time against the 8 TB/s HBM peak."""
    from comat_amd import ops
    k = ops.kernels()
    dev = torch.device("cuda:0")
    H, NP, L, NT = 8, 4096, 77, 4
    S = torch.randn(H, NP, L, device=dev)
    P = torch.empty(H, NP, L, device=dev, dtype=torch.bfloat16)
    mask = (torch.rand(2, NP, device=dev) > 0.5).float()
    tok_idx = torch.tensor([2, 3, 6, 7], dtype=torch.int32, device=dev)
    tok_obj = torch.tensor([0, 0, 1, 1], dtype=torch.int32, device=dev)
    num, den = torch.zeros(H, NT, device=dev), torch.zeros(H, NT, device=dev)
    avg = torch.zeros(NT, NP, device=dev)

    def timed(fn, n=20):
        for _ in range(3):
            fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / n * 1e-3

    t_sm = timed(lambda: k.softmax_fwd(S, P, H * NP, L))
    t_ga = timed(lambda: k.attnmap_gather_fwd(P, mask, tok_idx, tok_obj, num, den, avg, H, NP, L, NT))
    b_sm = H * NP * L * (4 + 2)            # scores read (fp32) + probabilities written (bf16)
    b_ga = H * NP * NT * 2 + NT * NP * 4   # selected columns read + per-token head-mean map written
    m_ga = H * NP * L * 2 + NT * NP * 4    # what the kernel moves: it streams the whole [pixels, 77] map, coalesced
    return {"softmax_writeback": {"bytes": b_sm, "us": round(t_sm * 1e6, 2), "GB/s": round(b_sm / t_sm / 1e9, 1),
                                  "frac_of_8TBs": round(b_sm / t_sm / 8e12, 4)},
            "gather": {"bytes": b_ga, "us": round(t_ga * 1e6, 2), "GB/s": round(b_ga / t_ga / 1e9, 1),
                       "frac_of_8TBs": round(b_ga / t_ga / 8e12, 4),
                       "moved_bytes": m_ga, "moved_GB/s": round(m_ga / t_ga / 1e9, 1),
                       "note": "algorithmic bytes = the 4 selected token columns; the probabilities are stored "
                               "[pixels, 77] (the layout the softmax and its backward stream), so every 64-byte segment "
                               "of a row holds a wanted element and the gather streams the map once through LDS "
                               "(two launches: per-head partial sums, then the head mean)"}}


def secondary_c3(trainer, batch, rank, sync, steps=3):
    """The reference's own recipe (scripts/sd15.sh:4-14 = BASELINE config C3: 50 denoise steps, 5 sampled ones with
    gradient, concept matching + GAN + attribute concentration on 2 of them) timed in the same run on the SAME models:
    45 no-grad UNet forwards replayed from per-timestep graphs + the step segments.  A few seconds; N = 1 only."""
    import numpy as np
    from comat_amd.segments import SegmentedStep
    from comat_amd.step import CoMatTrainer, StepConfig
    scfg = StepConfig(resolution=512, total_step=50, K=5, gan_loss=True, attrcon=True)
    tr = CoMatTrainer(trainer.pipe, trainer.bank, trainer.blip, trainer.D, scfg, seed=rank)
    n_graphs = tr.pipe.prepare_graphs(1, 512, 512, 77, scfg.total_step)
    dev = trainer.device
    b = dict(batch)
    b["noises"] = [torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(100 + i)).to(dev) for i in range(50)]
    m = np.zeros((2, 512, 512), dtype=bool)
    m[0, 60:250, 40:230] = True
    m[1, 280:480, 260:500] = True
    b["masks"], b["attributes"] = [m], [[[2, 3], [6, 7]]]
    fixed = dict(crop=(1, 1, 510, 510))
    st = SegmentedStep(tr)
    for kw in precapture_plan(scfg, fixed):
        st(b, **kw)
    st(b, **fixed)
    sync()
    t0, host = time.time(), 0.0
    for _ in range(steps):
        h0 = time.perf_counter()
        st(b, **fixed)
        host += time.perf_counter() - h0
    sync()
    ms = (time.time() - t0) / steps * 1e3
    total = step_tflop(50, 5, True)
    # where the step goes: HIP events around every segment replay and around the no-grad UNet replays (2 more steps)
    st.set_timing(True)
    ng = tr.pipe.graphed
    ng.timing = []
    for _ in range(2):
        st(b, **fixed)
    phases = st.timing_summary(2)
    phases["unet no-grad"] = {"ms_per_step": round(sum(a.elapsed_time(c) for a, c in ng.timing) / 2, 2),
                              "replays_per_step": len(ng.timing) / 2}
    st.set_timing(False)
    ng.timing = None
    return {"workload": "C3: SD1.5 512x512 bs=1, N=50 denoise steps (K=5 sampled with grad), concept-matching + GAN + attribute "
                        "concentration (2 steps), clip+AdamW for G and D - scripts/sd15.sh of the reference",
            "ms_per_step": round(ms, 1), "images_per_sec": round(1e3 / ms, 3), "steps": steps,
            "host_enqueue_ms_per_step": round(host / steps * 1e3, 1),
            "launch_mode": f"{n_graphs} no-grad UNet graphs + step segments ({st.stats()['segments']} segment graphs)",
            "step_algorithmic_tflop": total, "step_frac": total / (ms * 1e-3) / PEAK_BF16_TFLOPS,
            "gpu_ms_per_step_by_piece": phases}


def secondary_c4_own_process(timeout_s=400, config="c4", steps=4):
    """(config = "c5", round 6: the SDXL 1024^2 fp8-forward step the same way, two timed steps.)
    C4 measured the way a training job would run it - in a process of its own (`bench.py --config c4`), started from the
    default line.  Inside the process that already holds the C2 and C3 worlds, their graphs and ~100 GB of allocations the same
    step measured 12 - 19 % slower on every box (1 035 - 1 081 ms against 911 - 938: the segment replays themselves 3 - 6 %
    slower, the eager glue between them twice as slow; profiles/r04_w_c4_in_process.txt) - a property of this benchmark
    process, not of the step.  COMAT_SECONDARY_C4=inproc keeps the old form."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--config", config, "--no-cpu-baseline", "--no-kernel-timing",
           "--steps", str(steps), "--warmup", "1"]
    t0 = time.time()
    env = dict(os.environ, COMAT_SECONDARY="0")
    env.pop("COMAT_BENCH_DUMP", None)  # the parent's per-problem dump is the parent's
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
    line = next((l for l in reversed(r.stdout.splitlines()) if l.startswith("{") and '"metric"' in l), None)
    if r.returncode != 0 or line is None:
        raise RuntimeError(f"bench.py --config {config} exited with {r.returncode}: {r.stderr[-300:]}")
    d = json.loads(line)
    c = d["config"]
    return {"workload": c["workload"], "ms_per_step": round(d["ms_per_step"], 1), "images_per_sec": round(d["value"], 3),
            "steps": d["steps"], "warmup": d["warmup"], "build_s": c.get("build_s"), "wall_s": round(time.time() - t0, 1),
            "host_enqueue_ms_per_step": c.get("host_enqueue_ms_per_step"), "launch_mode": c.get("launch_mode"),
            "dtype": d.get("dtype"),
            "how": f"`python bench.py --config {config} --no-cpu-baseline --no-kernel-timing --steps {steps} --warmup 1` in a process of "
                   "its own, started by the default line (the C2 / C3 worlds of the parent stay allocated and idle meanwhile)"}


def secondary_c2_bs4_own_process(bs=4, timeout_s=300):
    """The C2 workload at the reference's operating point, `--train_batch_size 4` per GPU (scripts/sd15.sh:4): every kernel of
    the step sees 4x the rows (CFG batch 8).  Answers whether the kernels' fraction of peak at bs 1 is a property of the kernels
    or of the bs-1 problem sizes.  `python bench.py --bs 4 --steps 2` in a process of its own (as C4): two
    timed steps, the per-kernel family block from its own recorder step."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--config", "c2", "--bs", str(bs), "--no-cpu-baseline", "--steps", "2",
           "--warmup", "1"]
    t0 = time.time()
    env = dict(os.environ, COMAT_SECONDARY="0", COMAT_PROBE_EAGER="0", COMAT_ATTN_MAP_PROBE="0")
    env.pop("COMAT_BENCH_DUMP", None)  # the parent's per-problem dump is the parent's
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
    line = next((l for l in reversed(r.stdout.splitlines()) if l.startswith("{") and '"metric"' in l), None)
    if r.returncode != 0 or line is None:
        raise RuntimeError(f"bench.py --bs {bs} exited with {r.returncode}: {r.stderr[-300:]}")
    d = json.loads(line)
    c, rf = d["config"], d.get("roofline") or {}
    fams = {k: {kk: v[kk] for kk in ("ms", "launches", "TFLOP/s", "frac_of_peak")} for k, v in (rf.get("families") or {}).items()
            if k.startswith("gemm2_kernel (") or k.startswith("flash")}
    return {"workload": c["workload"], "per_gpu_batch": bs, "ms_per_step": round(d["ms_per_step"], 1),
            "images_per_sec": round(d["value"], 3), "steps": d["steps"], "warmup": d["warmup"], "launch_mode": c.get("launch_mode"),
            "step_algorithmic_tflop": rf.get("step_algorithmic_tflop"), "step_frac": rf.get("step_frac"), "families": fams,
            "gpu_ms_per_step_by_piece": c.get("gpu_ms_per_step_by_piece"), "wall_s": round(time.time() - t0, 1),
            "how": f"`python bench.py --config c2 --bs {bs} --no-cpu-baseline --steps 2 --warmup 1` in a process of its own"}


def secondary_c4(device, dtype, rank, sync, steps=3):
    """BASELINE config C4 (SDXL generator 512^2, SD1.5 discriminator, the full loss set of scripts/sdxl.sh) in the default
    line, so that the driver's run carries an SDXL number: its own world (random-init SDXL UNet / VAE, ~75 s to build), the 45
    no-grad UNet graphs, the step segments; `steps` timed steps after the capturing ones.  N = 1 only."""
    from comat_amd.segments import SegmentedStep
    t0 = time.time()
    tr, b, fixed, scfg, _, t_build = build_world(device, dtype, rank, "c4")
    st = SegmentedStep(tr)
    for kw in precapture_plan(scfg, fixed):
        st(b, **kw)
    st(b, **fixed)
    st(b, **fixed)  # one more untimed step: every lazily created buffer of the replay path exists
    sync()
    gc.collect()
    gc.freeze()     # the SDXL world's objects join the immortal set (as main() does for the C2 world before its timed region)
    t1, host = time.time(), 0.0
    for _ in range(steps):
        h0 = time.perf_counter()
        st(b, **fixed)
        host += time.perf_counter() - h0
    sync()
    ms = (time.time() - t1) / steps * 1e3
    total = step_tflop(scfg.total_step, scfg.K, scfg.gan_loss, sdxl=True, res=scfg.resolution)
    # where the step goes on the GPU (as for C3): HIP events around every segment replay and every no-grad UNet replay
    st.set_timing(True)
    ng = tr.pipe.graphed
    ng.timing = []
    for _ in range(2):
        st(b, **fixed)
    phases = st.timing_summary(2)
    phases["unet no-grad"] = {"ms_per_step": round(sum(a.elapsed_time(c) for a, c in ng.timing) / 2, 2),
                              "replays_per_step": len(ng.timing) / 2}
    st.set_timing(False)
    ng.timing = None
    out = {"workload": "C4: SDXL generator 512x512 bs=1 (SD1.5 discriminator), N=50 denoise steps (K=5 sampled with grad), concept-"
                       "matching + GAN + attribute concentration, clip+AdamW for G and D - scripts/sdxl.sh of the reference",
           "ms_per_step": round(ms, 1), "images_per_sec": round(1e3 / ms, 3), "steps": steps, "build_s": round(t_build, 1),
           "set_up_s": round(t1 - t0, 1), "launch_mode": f"no-grad UNet graphs + step segments ({st.stats()['segments']} segment graphs)",
           "step_algorithmic_tflop": total, "step_frac": total / (ms * 1e-3) / PEAK_BF16_TFLOPS,
           "host_enqueue_ms_per_step": round(host / steps * 1e3, 1), "gpu_ms_per_step_by_piece": phases,
           "note": "measured inside the process that also holds the C2 and C3 worlds and their graphs; `bench.py --config c4` on its own "
                   "measured 938 ms where this line's predecessor measured ~1 090 - 1 160 (profiles/r04_h_c4_ab.txt)"}
    del st, tr, b
    gc.collect()
    torch.cuda.empty_cache()
    return out


def precapture_plan(scfg, fixed):
    """Keyword sets for a few REAL optimisation steps that make a SegmentedStep visit every (slot, variant) once before
    the timed region: with attribute concentration each trained-call slot has two variants (maps captured or not) and
    the reference draws the capturing steps at random (training_script.py:589-590)."""
    if not scfg.attrcon:
        return []
    from comat_amd.step import sample_training_steps
    import random
    ts = fixed.get("training_steps") or sample_training_steps(scfg.total_step, scfg.K, random.Random(0))
    plans = [[ts[0], ts[1]], [ts[2], ts[3]]] if len(ts) >= 4 else [[t] for t in ts]
    if len(ts) >= 5:
        plans.append([ts[4]])
    base = {k: v for k, v in fixed.items() if k != "training_steps"}
    return [dict(base, training_steps=list(ts), attrcon_steps=p) for p in plans]


T_PROCESS = time.time()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c4", "c5"])
    ap.add_argument("--bs", type=int, default=1, help="prompts per GPU and step (the headline metric is quoted at 1; the reference "
                                                      "trains SD1.5 at 4: scripts/sd15.sh:4)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--selftest", action="store_true",
                    help="control-flow check of this script WITHOUT a GPU (tests/test_bench_contract.py): tiny model, "
                         "ABI simulator, gloo; its numbers mean nothing and the output is marked as such")
    args = ap.parse_args()
    import faulthandler
    faulthandler.enable()  # a native crash (GPU runtime, extension) leaves the Python stack of every thread on stderr

    from comat_amd import dist, ops
    rank, world, device = dist.init()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if args.selftest:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from sim_backend import SimKernels
        ops.set_kernel_backend(SimKernels())
        device = torch.device("cpu")
        sync = lambda: None
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
        from comat_amd import _hip
        ops.set_kernel_backend(_hip.HipKernels())
        sync = torch.cuda.synchronize
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    if not args.selftest:
        torch.cuda.set_per_process_memory_fraction(0.92)  # an over-sized configuration must fail as a Python OOM
    trainer, batch, fixed, scfg, usd_cpu, t_build = build_world(device, dtype, rank,
                                                                "selftest" if args.selftest else args.config, bs=args.bs)

    last_logs = {}

    # Launch modes (COMAT_STEP_MODE = auto | segments | graph | eager; COMAT_STEP_GRAPH=1/0 still selects graph / eager):
    #   eager    - ~17 k launches per step at ~10 us of host time each: host-bound.
    #   graph    - the WHOLE step captured once into a hipGraph (comat_amd.step.GraphedStep): needs a static launch topology
    #              (C2: every denoise step trained, no attribute concentration) and, with more than one rank, ends before the
    #              gradient exchange (exchange + optimizer follow eagerly).
    #   segments - the trained UNet calls, the head and the discriminator step replayed from per-piece graphs behind the eager
    #              sampler loop / loss assembly / exchange / optimizer (comat_amd.segments.SegmentedStep): ANY configuration,
    #              ANY number of ranks, the same path on 1 and on 8 GPUs (no collective is ever captured).
    #   auto     - segments everywhere; with a static topology the whole-step graph is probed against it (3 steps each) and
    #              the faster one is timed.  Every decision is agreed across ranks, so collectives stay matched.
    stepper, graph_note = None, "eager launches"
    mode = os.environ.get("COMAT_STEP_MODE") or {"1": "graph", "0": "eager"}.get(os.environ.get("COMAT_STEP_GRAPH", ""), "auto")
    probe_ms = {}

    def agree(value, op):
        """the same decision on every rank (each step holds two all-reduces: ranks that took different branches would
        leave the collectives of the timed region unmatched)"""
        if world == 1:
            return value
        t = torch.tensor([float(value)], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=op)
        return float(t)

    def drop_graph_hooks():
        # (the fixed-address crop tables of the BLIP preprocessing stay installed: segment graphs that survive a failed
        # whole-step capture read them, and Blip.tables() loads them with whatever crop an eager call asks for)
        trainer.pipe.trained_runner = trainer.head_runner = trainer.d_runner = None
        ops.reset_capture_stream(device)  # a failed capture may leave its streams in capture mode
        ops.drop_side_stream_state()
        trainer._d_stream = None
        trainer._d_pending, trainer._d_keep = False, None
        try:
            sync()
        except Exception:  # noqa: BLE001 - the pending error of the failed capture
            pass

    fail_at = os.environ.get("COMAT_SELFTEST_FAIL", "")  # "<rank>:<label>:<call index>": fault injection (--selftest only)

    def try_stepper(build, calls, label):
        """build() -> stepper; `calls` = keyword sets of the REAL optimisation steps to run through it (its capturing
        steps).  Every rank runs len(calls) optimisation steps whatever happens: a stepper call that raises is re-run
        eagerly, and so are the calls after it.  That keeps the collectives matched: every capture of a SegmentedStep call
        precedes that step's gradient exchange (sampler / head / D captures, then `_apply_updates`), so when one fails the
        other ranks are waiting in exactly the two all-reduces the eager re-run issues; GraphedStep captures AFTER an eager
        step of the same call and reports a failed capture through `.failed` instead of raising (that call was a whole
        step already).  Only then is the outcome agreed (MIN over ranks): one failing rank sends every rank to eager
        launches - loudly."""
        ok, err, cand = 1, "", None
        for i, kw in enumerate([None] + list(calls)):
            if ok:
                try:
                    if args.selftest and fail_at == f"{rank}:{label.split()[0]}:{i}":
                        raise RuntimeError("injected failure (COMAT_SELFTEST_FAIL)")
                    if kw is None:
                        cand = build()
                    else:
                        cand(batch, **kw)
                    continue
                except Exception as e:  # noqa: BLE001 - stay measurable
                    ok, err = 0, f"{type(e).__name__}: {e}"
                    print(f"[bench] rank {rank}: {label} failed ({err}); falling back to eager launches", file=sys.stderr)
                    drop_graph_hooks()
            if kw is not None:
                trainer.train_step(batch, **{k: v for k, v in kw.items() if v is not None})
        sync()
        if ok and getattr(cand, "failed", None):  # GraphedStep keeps a failed capture to itself (its call was one whole step)
            ok, err = 0, cand.failed
            print(f"[bench] rank {rank}: {label} failed ({err}); falling back to eager launches", file=sys.stderr)
        ok = agree(ok, torch.distributed.ReduceOp.MIN) if world > 1 else ok
        if not ok:
            drop_graph_hooks()
        return (cand if ok else None), (err or "another rank failed")

    def probe(fn, n=3):
        fn()
        sync()
        t0 = time.time()
        for _ in range(n):
            fn()
        sync()
        t = (time.time() - t0) / n
        return agree(t, torch.distributed.ReduceOp.MAX) if world > 1 else t

    static_topology = not scfg.attrcon and "training_steps" in fixed
    seg_stepper = graph_stepper = None
    if mode in ("auto", "segments"):
        from comat_amd.segments import SegmentedStep

        seg_stepper, err = try_stepper(lambda: SegmentedStep(trainer, dry=args.selftest),  # selftest: the same hooks, run
                                       list(precapture_plan(scfg, fixed)) + [fixed],          # eagerly (no GPU)
                                       "segments capture")  # real optimisation steps that visit every (slot, variant) once
        if seg_stepper is not None:
            stepper, graph_note = seg_stepper, "segments: trained UNet calls, head and D step replayed from hipGraphs"
        else:
            graph_note = f"eager launches (segment capture failed: {err})"
        if args.selftest and seg_stepper is not None:
            graph_note = "SELFTEST: segment hooks run eagerly on the CPU simulator"
    # (more than one rank: GraphedStep captures forward + backward only, the exchange and the optimizer follow eagerly - the
    # same split as the segments, so no collective is captured in either form)
    if not args.selftest and static_topology and mode in ("graph", "auto"):
        from comat_amd.step import GraphedStep

        # first call = one eager step (with its all-reduces), then the capture (no collective inside); second = first replay
        graph_stepper, err = try_stepper(lambda: GraphedStep(trainer), [fixed, fixed], "graph capture")
        if graph_stepper is not None and mode == "graph":
            stepper, graph_note = graph_stepper, ("whole step replayed from one hipGraph" if not graph_stepper.split() else
                                                  "forward + backward replayed from one hipGraph, gradient exchange and optimizer eager")
    if not args.selftest and mode == "auto" and seg_stepper is not None and graph_stepper is not None:
        probe_ms["segments"] = probe(lambda: seg_stepper(batch, **fixed)) * 1e3
        probe_ms["graph"] = probe(lambda: graph_stepper(batch, **fixed)) * 1e3
        if probe_ms["graph"] < probe_ms["segments"]:
            stepper, graph_note = graph_stepper, ("whole step replayed from one hipGraph" if not graph_stepper.split() else
                                                  "forward + backward replayed from one hipGraph, gradient exchange and optimizer eager")
        graph_note += f" (probe: segments {probe_ms['segments']:.0f} ms, whole-step graph {probe_ms['graph']:.0f} ms)"
    eager_ms = None
    if not args.selftest and world == 1 and static_topology and os.environ.get("COMAT_PROBE_EAGER", "1") != "0":
        # the same step launched eagerly, for reference (what a run without any graph costs on this box)
        eager_ms = probe(lambda: trainer.train_step(batch, **fixed)) * 1e3

    def run_step():
        if stepper is not None:
            last_logs.update(stepper(batch, **fixed))
        else:
            last_logs.update(trainer.train_step(batch, **fixed))

    for _ in range(args.warmup):
        run_step()
    sync()
    gc.collect()
    gc.freeze()  # the weights / module objects built above are immortal: keep the cyclic GC from rescanning them
    dist.barrier()
    sync()
    t0 = time.time()
    host_s = 0.0
    for _ in range(args.steps):
        h0 = time.perf_counter()
        run_step()
        host_s += time.perf_counter() - h0  # time the host needs to ENQUEUE a step (no sync inside)
    sync()
    dist.barrier()
    sync()
    dt = time.time() - t0
    per_rank_ms, allreduce_ms = None, None
    if world > 1:
        # diagnostics for the scaling runs: every rank's own time for the K steps, and the cost of the step's two
        # all-reduces (flat G and D gradient buffers) measured on their own after the timed region
        per_rank_ms = [round(v / args.steps * 1e3, 2) for v in dist.all_gather_scalar(dt, device)]
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
        if not args.selftest:
            bufs = [trainer.bank.flat_grad] + ([trainer.D.bank.flat_grad] if scfg.gan_loss else [])
            for b in bufs:
                torch.distributed.all_reduce(b.clone())
            sync()
            t0a = time.time()
            for _ in range(5):
                for b in bufs:
                    torch.distributed.all_reduce(b)
            sync()
            allreduce_ms = round((time.time() - t0a) / 5 * 1e3, 3)
    ms_per_step = dt / args.steps * 1e3
    value = world * args.bs * args.steps / dt  # args.bs images (prompts) per rank per step

    pieces = None
    if seg_stepper is not None and world == 1 and not args.no_kernel_timing and not args.selftest:
        # where a step goes on the GPU: HIP events around every segment replay of two more steps (segments mode)
        seg_stepper.set_timing(True)
        for _ in range(2):
            seg_stepper(batch, **fixed)
        pieces = seg_stepper.timing_summary(2)
        seg_stepper.set_timing(False)
    roofline = None
    if not args.no_kernel_timing and not args.selftest:
        # one extra eager step with a recorder around the kernel backend.  EVERY rank runs it (the step contains the
        # gradient all-reduce: a rank-0-only step would deadlock the collective); only rank 0 records and replays.
        rec = None
        if rank == 0:
            rec = CallRecorder(ops.kernels())
            ops.set_kernel_backend(rec)
        ops.set_side_stream_enabled(False)  # one stream: event pairs must not straddle kernels of another stream
        trainer.serial_d, trainer.flat_d = True, False
        # Hold the GPU back while the host queues the step: an event pair then brackets a kernel the queue already holds,
        # not the few microseconds the host needs to marshal the NEXT launch while the GPU sits idle (call 9: 25.0 us per
        # launch with a starving queue against 20.6 us in the rocprofv3 trace of the same kernels).
        if rank == 0 and hasattr(torch.cuda, "_sleep"):
            sync()
            t0s = time.time()
            torch.cuda._sleep(20_000_000)
            sync()
            per_cycle = (time.time() - t0s) / 20_000_000
            torch.cuda._sleep(int(0.25 / max(per_cycle, 1e-11)))
        trainer.train_step(batch, **fixed)
        sync()
        ops.set_side_stream_enabled(True)
        if rank == 0:
            ops.set_kernel_backend(rec.inner)
            fam = rec.measure()
    if rank == 0 and not args.no_kernel_timing and not args.selftest:
        dom = max(fam, key=lambda k: fam[k][4])
        t_rep, f_dom, n_dom, b_dom, t_dom = fam[dom]
        total = step_tflop(scfg.total_step, scfg.K, scfg.gan_loss, sdxl=args.config in ("c4", "c5"), res=scfg.resolution,
                           bs=args.bs)
        pmc = load_pmc_summary() if args.bs == 1 else None  # the committed counter passes are of the bs-1 step
        n_other = sum(rec.other.values())
        peak_of = lambda family: PEAK_FP8_TFLOPS if "fp8" in family else PEAK_BF16_TFLOPS
        roofline = {
            "bound": "mfma", "kernel": dom,
            "achieved": f_dom / t_dom / 1e12, "peak": peak_of(dom), "unit": "TFLOP/s",
            "frac": f_dom / t_dom / 1e12 / peak_of(dom),
            "traffic": pmc.get("traffic_bytes_per_launch") if pmc else None,
            "traffic_note": pmc.get("note") if pmc else ("no committed PMC pass found (profiles/r0N_pmc_kernels.json)" if args.bs == 1
                                                         else "the committed counter passes are of the bs-1 step"),
            "mfma_busy_frac": pmc.get("mfma_busy_frac") if pmc else None,
            "algorithmic_bytes_per_launch": int(b_dom / n_dom),
            "launches_per_step": n_dom, "avg_launch_ms": t_dom / n_dom * 1e3,
            "algorithmic_tflop_per_step_in_kernel": f_dom / 1e12,
            "timing": "HIP events around every launch of this kernel in one eager single-stream step (operands as warm as in "
                      "a real step), minus the cost of an empty event pair measured in the same step",
            "event_pair_overhead_us": round(rec.null_us, 2),
            "avg_launch_ms_raw_brackets": sum(v for k_, v in rec.raw_in_step.items()
                                              if rec.calls[k_][5] == 1) / n_dom * 1e3 if "gemm2_kernel (" in dom else None,
            "achieved_replayed": f_dom / t_rep / 1e12,
            "avg_launch_ms_replayed": t_rep / n_dom * 1e3,
            "timing_replayed": "every distinct problem of the step again, back to back from a hipGraph of 10 launches "
                               "(warm caches, no gaps), weighted by its call count",
            "step_algorithmic_tflop": total, "step_frac": total / (ms_per_step * 1e-3) / PEAK_BF16_TFLOPS,
            "mfma_kernel_ms_per_step": round(sum(v[4] for v in fam.values()) * 1e3, 2),
            "other_launches_per_step": n_other,
            "families": {k: {"ms": round(v[4] * 1e3, 2), "ms_replayed": round(v[0] * 1e3, 2), "tflop": round(v[1] / 1e12, 3),
                             "launches": v[2], "TFLOP/s": round(v[1] / v[4] / 1e12, 1),
                             "frac_of_peak": round(v[1] / v[4] / 1e12 / peak_of(k), 4),
                             "TFLOP/s_replayed": round(v[1] / v[0] / 1e12, 1)}
                         for k, v in sorted(fam.items(), key=lambda kv: -kv[1][4])},
        }
    def stage(what):  # progress markers on stderr (rank 0): which part of the run a log ends in
        if rank == 0:
            print(f"[bench] {time.time() - T_PROCESS:6.1f} s: {what}", file=sys.stderr, flush=True)

    stage(f"timed region done: {ms_per_step:.1f} ms per step")
    attn_map = None
    if rank == 0 and not args.no_kernel_timing and not args.selftest and os.environ.get("COMAT_ATTN_MAP_PROBE", "1") != "0":
        attn_map = attn_map_probe()
    secondary = None
    if (rank == 0 and world == 1 and args.config == "c2" and args.bs == 1 and not args.selftest and not args.no_kernel_timing
            and os.environ.get("COMAT_SECONDARY", "1") != "0"):
        stage("secondary c3")
        try:
            secondary = {"c3": secondary_c3(trainer, batch, rank, sync)}
        except Exception as e:  # noqa: BLE001 - the headline number must not depend on the secondary one
            secondary = {"c3": {"error": f"{type(e).__name__}: {e}"}}
        if os.environ.get("COMAT_SECONDARY_C4", "1") != "0":
            if time.time() - T_PROCESS > 300:  # keep the default run within minutes whatever the box
                secondary["c4"] = {"skipped": "the run had used more than 300 s before the SDXL measurement"}
            else:
                stage("secondary c4")
                try:
                    if os.environ.get("COMAT_SECONDARY_C4", "1") == "inproc":
                        secondary["c4"] = secondary_c4(device, dtype, rank, sync)
                    else:
                        secondary["c4"] = secondary_c4_own_process()
                except Exception as e:  # noqa: BLE001
                    secondary["c4"] = {"error": f"{type(e).__name__}: {e}"}
        if os.environ.get("COMAT_SECONDARY_BS4", "1") != "0":
            if time.time() - T_PROCESS > 420:
                secondary["c2_bs4"] = {"skipped": "the run had used more than 420 s before the batch-4 measurement"}
            else:
                stage("secondary c2_bs4")
                try:
                    secondary["c2_bs4"] = secondary_c2_bs4_own_process()
                except Exception as e:  # noqa: BLE001
                    secondary["c2_bs4"] = {"error": f"{type(e).__name__}: {e}"}
        if os.environ.get("COMAT_SECONDARY_C5", "1") != "0":  # BASELINE.json configs[4]: SDXL 1024^2, fp8 forward (delayed scales)
            if time.time() - T_PROCESS > 480:
                secondary["c5"] = {"skipped": "the run had used more than 480 s before the SDXL 1024^2 measurement"}
            else:
                stage("secondary c5")
                try:
                    secondary["c5"] = secondary_c4_own_process(timeout_s=400, config="c5", steps=2)
                except Exception as e:  # noqa: BLE001
                    secondary["c5"] = {"error": f"{type(e).__name__}: {e}"}
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.config not in ("c4", "c5") and not args.selftest:
        stage("cpu baseline")
        cpu = cpu_baseline(usd_cpu, scfg)
    stage("done")
    if rank == 0 and os.environ.get("COMAT_BENCH_LOGS"):  # loss terms of the last timed step (sanity evidence)
        print({k: (float(v) if torch.is_tensor(v) else v) for k, v in last_logs.items()}, file=sys.stderr)
    if rank == 0:
        out = {
            "metric": f"CoMat train-step images/sec ({'SDXL' if args.config in ('c4', 'c5') else 'SD1.5'} "
                      f"{scfg.resolution}^2, bs={args.bs}/GPU)",
            "value": value, "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("bf16 (fp8 e4m3 forward products of the generator UNet)" if args.config == "c5" else "bf16")
            if dtype == torch.bfloat16 else "f32",
            "data": "SELFTEST (CPU simulator, tiny model): not a measurement" if args.selftest else "synthetic",
            "config": {"workload": f"{args.config.upper()}: "
                                   f"{'SDXL (SD1.5 discriminator' + (', fp8 UNet forward)' if args.config == 'c5' else ')') if args.config in ('c4', 'c5') else 'SD1.5'} "
                                   f"{scfg.resolution}x{scfg.resolution} bs={args.bs}/GPU, N={scfg.total_step} denoise steps "
                                   f"(K={scfg.K} with grad), CFG 7.5, LoRA r=128, concept-matching (BLIP-large) + GAN "
                                   f"fidelity (G+D step)" + (" + attribute concentration" if scfg.attrcon else "") +
                                   ", clip+AdamW for G and D",
                       "parallelism": f"dp{world}", "build_s": round(t_build, 1),
                       "host_enqueue_ms_per_step": round(host_s / args.steps * 1e3, 1), "launch_mode": graph_note,
                       "probe_ms_per_step": {k: round(v, 1) for k, v in probe_ms.items()} or None,
                       "eager_ms_per_step": None if eager_ms is None else round(eager_ms, 1),
                       "gpu_ms_per_step_by_piece": pieces,
                       "per_rank_ms_per_step": per_rank_ms, "allreduce_ms_per_step": allreduce_ms},
            "roofline": roofline, "cpu_baseline": cpu, "attn_map": attn_map, "secondary": secondary,
        }
        print(json.dumps(out))


if __name__ == "__main__":
    main()
