"""Autograd operators of the CoMat step, each a thin `torch.autograd.Function` over the C-ABI kernels.

Activations are channels-last token matrices: an image tensor (B, H, W, C) is the contiguous 2-D tensor
[B*H*W, C].  Frozen weights carry both orientations (`w` [N,K] for the forward GEMM, `wt` [K,N] for the
data-gradient) so that every large contraction runs through the k-contiguous MFMA path; only LoRA factors and
attention use the k-major operand paths.  Nothing here computes on the CPU: the kernel backend is the HIP library
(`comat_amd._hip.HipKernels`); `set_kernel_backend` exists only so that `tests/` can check the host logic on a
machine without a GPU by plugging in a simulator of the C ABI.
"""
from __future__ import annotations

import os

import torch
from torch.autograd import Function

from . import _hip
from ._hip import ACT_GELU, ACT_NONE, ACT_SILU, UN_AFFINE, UN_COPY, UN_GELU, UN_SILU  # noqa: F401

_K = None


def kernels():
    global _K
    if _K is None:
        _K = _hip.HipKernels()  # raises if libcomat_hip.so is missing: no fallback
    return _K


def set_kernel_backend(k):
    """Test seam (tests/ only): replace the kernel backend by an object with the HipKernels method set."""
    global _K
    _K = k


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# ---- side stream for work that is off the critical path of backward (LoRA weight gradients) -----------------------
_side = {}
_side_keep = []
_side_enabled = True
_side_suspended = 0  # >0: weight gradients stay on the issuing stream (see no_side_streams)


def set_side_stream_enabled(flag: bool):
    global _side_enabled
    _side_enabled = bool(flag)


def side_streams_enabled():
    return _side_enabled


def _side_stream(dev):
    """A second HIP stream (None on CPU / when disabled), one per stream that issues backward work: the K = B*H*W
    split-K GEMMs of the LoRA weight gradients have few tiles each and no consumer until the optimizer, so they
    overlap the main backward chain."""
    if dev.type != "cuda" or not _side_enabled or _side_suspended:
        return None
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    st = _side.get(key)
    if st is None:
        st = _side[key] = torch.cuda.Stream(device=dev)
    return st


# ---- the stream hipGraphs of this package are captured on ---------------------------------------------------------------
_capture = {}


def capture_stream(dev):
    """ONE capture stream per device for every hipGraph the package records for replay on the main stream (step graph,
    step segments, no-grad UNet forwards).  Kernels pick their workspaces by stream, so these graphs share one set of
    workspaces - legal because they are only ever replayed on one stream, one after the other - and that set is created
    and zeroed HERE, eagerly, together with the set of the stream's side stream: a workspace first touched inside a
    capture would be zeroed by a node of that one graph only (see _hip.HipKernels._no_capture)."""
    dev = torch.device(dev)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    st = _capture.get(dev)
    if st is None:
        st = _capture[dev] = torch.cuda.Stream(device=dev)
    prepare_capture_stream(dev, st)  # idempotent per kernel backend instance (tests install a fresh one per test)
    return st


class graph_capture:
    """`torch.cuda.graph(g, pool=..., stream=...)` with Python's cyclic garbage collector switched off for the duration of
    the capture.  torch collects garbage BEFORE the capture begins; a collection that the allocation counters trigger in the
    MIDDLE of it runs finalizers of whatever became unreachable - an old CUDAGraph, a stream, pool memory of a finished
    test or stepper - and those release HIP objects while a stream is capturing: the capture is invalidated at best (the
    replay then faults), the process aborts at worst (both seen on MI355X, round 3).  Captures are short; the collector
    is switched back on (to its previous state) at the end."""

    def __init__(self, graph, pool=None, stream=None, **kw):
        self._ctx = torch.cuda.graph(graph, pool=pool, stream=stream, **kw)

    def __enter__(self):
        import gc
        self._gc_was_on = gc.isenabled()
        r = self._ctx.__enter__()  # synchronises, collects garbage, empties the cache, begins the capture
        gc.disable()
        return r

    def __exit__(self, *exc):
        import gc
        try:
            return self._ctx.__exit__(*exc)
        finally:
            if self._gc_was_on:
                gc.enable()


def reset_capture_stream(dev):
    """after a FAILED capture: the capture stream (and streams forked from it) may be left in capture mode by the runtime -
    forget it, the next capture_stream() call makes a fresh one"""
    dev = torch.device(dev)
    if dev.type != "cuda":
        return  # (the CPU simulator has no streams)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    st = _capture.pop(dev, None)
    if st is not None:
        _side.pop((dev, st.cuda_stream), None)


def prepare_capture_stream(dev, st):
    """create the per-stream workspaces of `st` and of the side stream forked from it, outside any capture (idempotent)"""
    k = kernels()
    if not hasattr(k, "prepare_stream"):
        return
    global _side_suspended
    made = False
    with torch.cuda.stream(st):
        made |= bool(k.prepare_stream(dev))
        saved, _side_suspended = _side_suspended, 0
        try:
            side = _side_stream(torch.device(dev))
        finally:
            _side_suspended = saved
        if side is not None:
            with torch.cuda.stream(side):
                made |= bool(k.prepare_stream(dev))
    if made:
        torch.cuda.synchronize(dev)


class no_side_streams:
    """Context: LoRA weight gradients of backward passes started inside it run on their issuing stream.  Used for the
    D step when it is itself forked onto its own stream inside a hipGraph capture: a fork from a forked stream (nested)
    crashes hipStreamEndCapture on ROCm 7.2, a single level of forks captures fine."""

    def __enter__(self):
        global _side_suspended
        _side_suspended += 1

    def __exit__(self, *exc):
        global _side_suspended
        _side_suspended -= 1


_join_queued = False
_side_dirty = []  # side streams that received work since the last join


def join_side_streams():
    """Make the current stream wait for everything queued on the side streams since the last join.  Queued
    automatically as an autograd end-of-backward callback, so LoRA gradients are complete (in stream order) when
    `.backward()` returns.  Only streams that were actually forked are waited for: inside a hipGraph capture a wait on
    a stream that is not part of the capture would be an illegal cross-capture dependency."""
    global _join_queued
    _join_queued = False
    flush_weight_grads()
    for dev, st in _side_dirty:
        torch.cuda.current_stream(dev).wait_stream(st)
    _side_dirty.clear()
    _side_keep.clear()


def _queue_join():
    global _join_queued
    if not _join_queued:
        _join_queued = True
        torch.autograd.Variable._execution_engine.queue_callback(join_side_streams)


def reset_side_stream_state():
    """Start of a top-level forward/backward: join what a previous backward may have left behind (an exception in the
    middle of a backward skips its end-of-backward callback; the latch would stay set and no later backward would ever
    join the side streams again) and drop the latch."""
    global _join_queued
    if _side_dirty or any(q.items for q in _ttq.values()):
        join_side_streams()
    _join_queued = False
    _side_keep.clear()


def drop_side_stream_state():
    """After a FAILED graph capture: forget the weight gradients the aborted pass queued and the side streams it marked
    (their operands belong to the dead capture - they must not be launched), without joining anything."""
    global _join_queued
    _ttq.clear()
    _side_dirty.clear()
    _side_keep.clear()
    _join_queued = False


# ---- deferred, grouped LoRA weight gradients -----------------------------------------------------------------------
# dU = g^T h and dD = u^T x of every LoRA projection are k-major products over the token axis with a handful of output
# tiles each (~720 per SD1.5 step).  Nothing reads them before the optimizer, so a backward pass QUEUES them here and
# hands them to comat_gemm_tt_grouped in groups of <= TT_GROUP problems: one launch fills the chip where 48 small ones
# each paid their own split-K combine.  A group is flushed when it is full, when a new problem
# accumulates into an output the group already holds (the same factor at another denoise step: the two must stay in
# stream order), and at the end of the backward pass (join_side_streams).  Grouping is a pure function of the call
# sequence, so results stay bit-reproducible run to run, eager or replayed from a graph.
TT_GROUP = 48
_ttq = {}  # (device, issuing stream) -> _TTQueue
_tt_grouping = os.environ.get("COMAT_TT_GROUPED", "1") != "0"


def set_tt_grouping(flag: bool):
    """tests / A-B runs: False issues every weight gradient as its own comat_gemm launch (the round-2 path)"""
    global _tt_grouping
    flush_weight_grads()
    _tt_grouping = bool(flag)


class _TTQueue:
    def __init__(self, dev, issuing, side):
        self.dev, self.issuing, self.side = dev, issuing, side
        self.items, self.outs, self.keep, self.pre = [], set(), [], []

    def add(self, prob, keep, pre=None):
        # the byte range the problem accumulates into: [C, C + ((M - 1) ldc + N) * 4).  A problem whose output OVERLAPS one
        # the group already holds (the same factor at another denoise step, or any partially overlapping view) must not
        # share its launch: the two read-modify-write passes would race
        lo = prob[2].data_ptr()
        hi = lo + ((prob[3] - 1) * prob[8] + prob[4]) * 4
        if len(self.items) >= TT_GROUP or any(lo < h and l < hi for l, h in self.outs):
            self.flush()
        self.items.append(prob)
        self.outs.add((lo, hi))
        self.keep.append(keep)
        if pre is not None:
            self.pre.append(pre)

    def flush(self):
        if not self.items:
            return
        items, keep, pre = self.items, self.keep, self.pre
        self.items, self.outs, self.keep, self.pre = [], set(), [], []
        if self.side is None:
            with torch.cuda.stream(self.issuing):
                for fn in pre:
                    fn()
                kernels().gemm_tt_grouped(items)
        else:
            self.side.wait_stream(self.issuing)  # every operand queued so far has been produced on the issuing stream
            with torch.cuda.stream(self.side):
                for fn in pre:  # operands the problems read that nothing on the issuing stream needs (merged LoRA: h, u)
                    fn()
                kernels().gemm_tt_grouped(items)
            if not any(st is self.side for _, st in _side_dirty):
                _side_dirty.append((self.dev, self.side))
            _side_keep.append(keep)  # operands stay alive until join_side_streams()


class _HostQueue(_TTQueue):
    """CPU tensors (tests with the ABI simulator): same grouping logic, no streams"""

    def flush(self):
        if self.items:
            items, pre = self.items, self.pre
            self.items, self.outs, self.keep, self.pre = [], set(), [], []
            for fn in pre:
                fn()
            kernels().gemm_tt_grouped(items)


def _tt_enqueue(dev, problems, keep, pre=None):
    """queue weight-gradient problems [(A, B, C, M, N, K, lda, ldb, ldc)] of the backward pass running on the current
    stream; `keep` = tensors that own the operands; `pre` = a callable that produces operands only these problems read
    (launched right in front of their group, on the stream the group runs on)"""
    if dev.type != "cuda":
        q = _ttq.get((dev, 0))
        if q is None:
            q = _ttq[(dev, 0)] = _HostQueue(dev, None, None)
    else:
        cur = torch.cuda.current_stream(dev)
        key = (dev, cur.cuda_stream)
        q = _ttq.get(key)
        if q is None:
            q = _ttq[key] = _TTQueue(dev, cur, None)
        if not q.items:
            q.side = _side_stream(dev)  # decided per group: side streams may be suspended for a forked D step
    for j, pr in enumerate(problems):
        q.add(pr, keep, pre if j == 0 else None)
    _queue_join()


def flush_weight_grads():
    """launch every queued weight-gradient group (idempotent)"""
    for q in list(_ttq.values()):
        q.flush()


# ----------------------------------------------------------------------------------------------------------------
# fp8 forward (BASELINE.json configs[4]: "fp8 MFMA UNet forward with bf16 backward")
# ----------------------------------------------------------------------------------------------------------------
# Inside `with fp8_forward(True)` the FORWARD product of every frozen Linear / conv that was tagged `allow_fp8` (the
# generator UNet's block layers, comat_amd/unet.py) and whose contraction length per tap is a multiple of 64 runs on
# the fp8 (OCP e4m3) MFMA: the activation is quantised per tensor on the fly (abs-max scale), the frozen weight once.
# The LoRA branch, the attention products, norms and every BACKWARD product stay in the storage dtype and use the
# unquantised saved activations and weights: gradients are those of the bf16 network evaluated at the fp8 forward's
# activations (the usual straight-through treatment of the quantiser).
_fp8_on = False


class fp8_forward:
    def __init__(self, flag=True):
        self.flag = bool(flag)

    def __enter__(self):
        global _fp8_on
        self.prev, _fp8_on = _fp8_on, self.flag
        return self

    def __exit__(self, *exc):
        global _fp8_on
        _fp8_on = self.prev
        return False


def fp8_eligible(holder, k_inner):
    return bool(getattr(holder, "allow_fp8", False)) and k_inner % 64 == 0


def _use_fp8(holder, k_inner):
    return _fp8_on and fp8_eligible(holder, k_inner)


def fp8_weight(holder):
    """(e4m3 bytes, scale) of a frozen weight in its forward orientation, quantised once (frozen: never refreshed)"""
    w8 = getattr(holder, "_w8", None)
    if w8 is None:
        w8 = holder._w8 = kernels().fp8_quantize(holder.w.contiguous())
    return w8


def fp8_weight_group(lins):
    """(bytes [G, N, K], scales [G]) of projections whose frozen weights are co-allocated at a constant spacing (frozen_linear_group):
    each weight under ITS OWN scale, as fp8_weight would quantise it, but in one buffer - the group's fp8 products are then one
    batched launch (comat_gemm_params::s_scale_b).  None when the weights are not co-allocated."""
    g8 = getattr(lins[0], "_w8_group", None)
    if g8 is None:
        if len(lins) < 2 or _uniform_stride([lin.w for lin in lins]) is None:
            return None
        G, (N, Kd) = len(lins), lins[0].w.shape
        w8 = torch.empty((G, N, Kd), dtype=torch.uint8, device=lins[0].w.device)
        sc = torch.empty(G, dtype=torch.float32, device=w8.device)
        for i, lin in enumerate(lins):
            lin._w8 = kernels().fp8_quantize(lin.w.contiguous(), out=w8[i], scale=sc[i:i + 1])
        g8 = lins[0]._w8_group = (w8, sc)
    return g8


# ---- activation scales ------------------------------------------------------------------------------------------------
# "jit" (rounds 2-5): every activation that enters an fp8 product is quantised under its OWN abs-max - two launches per tensor
# (a reduction with a ticket, then the bytes), ~1 300 of them per SDXL forward.
# "delayed" (round 6; COMAT_FP8_SCALING=delayed, bench.py --config c5): a quantisation SITE (the input of one frozen layer) keeps
# a scale and a running abs-max in two device words (include/comat_hip.h, ABI 8).  Every tensor that passes the site during an
# optimizer step is quantised under the scale that is already there - the abs-max over ALL of the previous step's calls of that
# site (every denoise step, trained or not) - and folds its own abs-max into the running maximum; fp8_end_of_step() (after the
# optimizer) turns the maxima into the next step's scales.  One launch per tensor, and none where the producer emits the bytes
# itself: LayerNorm / GroupNorm(+SiLU) store the e4m3 bytes next to their output when told whom they feed (`fp8_for=`) - the
# same bits as quantising the stored output.  Values beyond the previous step's abs-max saturate at +-448 * scale, as in every
# delayed-scaling recipe.  Before the first step the scales come from fp8_calibration(): one no-grad pass in which every site
# quantises just in time AND records its abs-max.
# COMAT_FP8_KTAIL (default 1): the LoRA up projection of a frozen projection rides in the fp8 product's launch as a bf16 k-tail
# (comat_gemm_params::A2k); 0 = its own launch behind it (rounds 2-5), for A/B runs
_fp8_ktail = os.environ.get("COMAT_FP8_KTAIL", "1") != "0"
# COMAT_FP8_GEGLU_Q8 (default 1): `ff.net.0.proj` + GEGLU emits the e4m3 bytes for `ff.net.2` from its epilogue (comat_gemm_params::q8)
_fp8_geglu_q8 = os.environ.get("COMAT_FP8_GEGLU_Q8", "1") != "0"
# COMAT_FP8_FLASH_Q8 (default 1): the fused attention forward emits the e4m3 bytes for its output projection (comat_flash_attn_fwd_q)
_fp8_flash_q8 = os.environ.get("COMAT_FP8_FLASH_Q8", "1") != "0"
_FP8_MAX_SITES = 4096
_fp8_scaling = os.environ.get("COMAT_FP8_SCALING", "jit")
_fp8_calibrating = False
_fp8_states = {}


def set_fp8_scaling(mode: str):
    global _fp8_scaling
    assert mode in ("jit", "delayed"), mode
    _fp8_scaling = mode


def fp8_scaling():
    return _fp8_scaling


class _Fp8State:
    """the scale / running-maximum words of every quantisation site on one device (fixed addresses: captured graphs read them)"""

    def __init__(self, device):
        self.scale = torch.zeros(_FP8_MAX_SITES, dtype=torch.float32, device=device)
        self.amax = torch.zeros(_FP8_MAX_SITES, dtype=torch.int32, device=device)  # float bits of a non-negative value
        self.n = 0


def fp8_state(device):
    """allocate the site table of `device` (call once OUTSIDE any graph capture: UNet.__init__ does)"""
    key = str(torch.device(device))
    st = _fp8_states.get(key)
    if st is None:
        st = _fp8_states[key] = _Fp8State(device)
    return st


def _fp8_site(holder, device):
    """(scale [1], amax [1]) views of the site in front of `holder` (index assigned at first use; no device allocation)"""
    site = getattr(holder, "_fp8_site", None)
    if site is None:
        st = fp8_state(device)
        assert st.n < _FP8_MAX_SITES, "fp8: site table full"
        i = st.n
        st.n += 1
        site = holder._fp8_site = (st.scale[i:i + 1], st.amax[i:i + 1])
    return site


class fp8_calibration:
    """`with fp8_calibration():` every site quantises just in time (its own abs-max) and records the abs-max: run the sampler once
    under it, then fp8_end_of_step() (TrainableSDPipeline.fp8_calibrate does both)"""

    def __enter__(self):
        global _fp8_calibrating
        self.prev, _fp8_calibrating = _fp8_calibrating, True
        for st in _fp8_states.values():
            st.amax.zero_()
        return self

    def __exit__(self, *exc):
        global _fp8_calibrating
        _fp8_calibrating = self.prev
        return False


def fp8_end_of_step():
    """delayed scaling: the running maxima of this step become the next step's scales (one launch per device; a no-op otherwise)"""
    if _fp8_scaling != "delayed":
        return
    for st in _fp8_states.values():
        if st.n:
            kernels().fp8_scales_update(st.amax, st.scale, st.n)


def fp8_act(x, holder):
    """(e4m3 bytes, scale [1]) of the activation x entering the fp8 product of `holder`"""
    k = kernels()
    if _fp8_scaling != "delayed":
        return k.fp8_quantize(x)
    sc, am = _fp8_site(holder, x.device)
    if _fp8_calibrating:
        return k.fp8_quantize(x, scale=sc, amax=am)[0], sc
    pre = getattr(x, "_fp8", None)
    if pre is not None and pre[1] is sc:  # the producer of x stored the bytes for this very site
        return pre[0], sc
    return k.fp8_quantize_scaled(x, sc, am), sc


def _fp8_producer_site(holder, k_inner, device):
    """a norm that feeds `holder`: the site it should quantise for, or None (no fp8, not eligible, jit scales, calibration pass)"""
    if holder is None or _fp8_scaling != "delayed" or _fp8_calibrating or not _use_fp8(holder, k_inner):
        return None
    return _fp8_site(holder, device)


# ----------------------------------------------------------------------------------------------------------------
# parameter holders
# ----------------------------------------------------------------------------------------------------------------
class FrozenLinear:
    """A frozen nn.Linear: weight [N,K] and its transpose [K,N] in the compute dtype, bias fp32."""

    def __init__(self, weight: torch.Tensor, bias, dtype, device):
        w = weight.to(device=device, dtype=torch.float32)
        self.w = w.to(dtype).contiguous()
        self.wt = w.t().contiguous().to(dtype)
        self.bias = None if bias is None else bias.to(device=device, dtype=torch.float32).contiguous()
        self.out_features, self.in_features = self.w.shape


class FrozenGegluLinear:
    """`ff.net.0.proj` of a BasicTransformerBlock followed by its GEGLU (3P diffusers GEGLU: `value, gate = proj(x).chunk(2, -1);
    value * gelu(gate)`): a frozen Linear(in -> 2 D) whose output ROWS are stored interleaved in sixteens - rows 32 t .. 32 t + 15 =
    value channels 16 t .., rows 32 t + 16 .. 32 t + 31 = their gate channels - so that one lane of the GEMM epilogue owns value
    and gate of the same 8 channels and the product leaves the GEMM (comat_gemm_params::epi2).  `w` [2 D, in] / `wt` [in, 2 D]
    / `bias` [2 D] in that order; `out_features` = D."""

    def __init__(self, weight: torch.Tensor, bias, dtype, device):
        n2, k = weight.shape
        D = n2 // 2
        assert n2 % 32 == 0, "GEGLU projection: 2 D must be a multiple of 32"
        t = torch.arange(n2 // 32).reshape(-1, 1, 1)
        j = torch.arange(16).reshape(1, 1, -1)
        half = torch.arange(2).reshape(1, -1, 1)
        self.perm = (half * D + t * 16 + j).reshape(-1)  # interleaved position -> original row
        w = weight.to(device=device, dtype=torch.float32)[self.perm.to(device)]
        self.w = w.to(dtype).contiguous()
        self.wt = w.t().contiguous().to(dtype)
        self.bias = None if bias is None else bias.to(device=device, dtype=torch.float32)[self.perm.to(device)].contiguous()
        self.out_features, self.pre_features, self.in_features = D, n2, k


def frozen_linear_group(weights, biases, dtype, device):
    """FrozenLinears of projections that read the same input (same [N, K] each), allocated as ONE [G, N, K] buffer
    (+ one [G, K, N] for the transposes): the group's forward is then a single batched launch."""
    G = len(weights)
    shp = tuple(weights[0].shape)
    assert all(tuple(w.shape) == shp for w in weights)
    w32 = torch.stack([w.to(device=device, dtype=torch.float32) for w in weights])
    W = w32.to(dtype).contiguous()
    Wt = w32.transpose(1, 2).contiguous().to(dtype)
    out = []
    for i in range(G):
        lin = FrozenLinear.__new__(FrozenLinear)
        lin.w, lin.wt = W[i], Wt[i]
        b = biases[i]
        lin.bias = None if b is None else b.to(device=device, dtype=torch.float32).contiguous()
        lin.out_features, lin.in_features = shp
        out.append(lin)
    return out


class FrozenConv:
    """A frozen conv2d.  `w` is [Cout, KH, KW, Cin]; `wd` is the tap-flipped, channel-transposed weight
    [Cin, KH, KW, Cout] that turns the data-gradient into the same implicit-GEMM gather."""

    def __init__(self, weight_oihw: torch.Tensor, bias, dtype, device, stride=1, pad=1):
        w = weight_oihw.to(device=device, dtype=torch.float32)
        self.cout, self.cin, self.kh, self.kw = w.shape
        self.w = w.permute(0, 2, 3, 1).contiguous().to(dtype)
        self.wd = w.flip(2, 3).permute(1, 2, 3, 0).contiguous().to(dtype)
        self.bias = None if bias is None else bias.to(device=device, dtype=torch.float32).contiguous()
        self.stride, self.pad = stride, pad


class LoRAGroup:
    """The LoRA factors of G projections that read the SAME input (q/k/v of a self-attention, k/v of a
    cross-attention, or a single projection):  y_i = x W_i^T + b_i + s * (x D_i^T) U_i^T.
    The G down factors are adjacent in the store's flat buffers, so `down_cat` [G*r, in] is ONE matrix: one GEMM
    produces all low-rank activations and one GEMM all down-gradients."""

    def __init__(self, store, index, down_cat, ups, rank, scale=1.0):
        self.store, self.index = store, index
        self.down_cat, self.ups = down_cat, ups      # fp32 leaves (views of store.flat) with .grad views
        self.rank, self.scale, self.size = rank, scale, len(ups)

    def compute_copies(self):
        """(down_cat [G*r, in], [up_i [out_i, r]], down_cat^T [in, G*r], [up_i^T [r, out_i]]) in the compute dtype."""
        self.store.ensure_compute_copy()
        return self.store.group_views[self.index]


class LoRAStore:
    """All trainable LoRA factors of one model in ONE flat fp32 buffer (+ one flat gradient buffer).  Every factor
    is a leaf view whose .grad is a view of the flat gradient: the GEMM epilogues accumulate weight gradients in
    place, RCCL all-reduces the flat buffer and the fused clip+AdamW kernel consumes it.  Two derived buffers are
    refreshed by one kernel each after an optimizer step: `flat_c` (compute-dtype copy) and `flat_t` (compute-dtype
    TRANSPOSED copies of the grouped down factors, so their data-gradient runs through the k-contiguous GEMM path).

    spec: list of groups; a group is a list of (down_name, up_name, down [r, in], up [out, r]) sharing `in`."""

    def __init__(self, spec, dtype, device, scale=1.0):
        self.names, shapes, layout = [], [], []
        off = toff = 0
        for members in spec:
            r, cin = members[0][2].shape
            assert all(tuple(m[2].shape) == (r, cin) and m[3].shape[1] == r for m in members)
            g = dict(down_off=off, rank=r, cin=cin, n=len(members), t_off=toff, ups=[], ut_offs=[])
            for dn, _, d, _ in members:
                self.names.append(dn)
                shapes.append((off, tuple(d.shape)))
                off += r * cin
            for _, un, _, u in members:
                self.names.append(un)
                shapes.append((off, tuple(u.shape)))
                g["ups"].append((off, tuple(u.shape)))
                off += u.shape[0] * r
            toff += len(members) * r * cin
            for _, _, _, u in members:  # transposed up factors U^T [r, out] follow the group's transposed down block
                g["ut_offs"].append(toff)
                toff += u.shape[0] * r
            layout.append(g)
        total = off
        src = {}
        for members in spec:
            for dn, un, d, u in members:
                src[dn], src[un] = d, u
        self.dtype, self.device = dtype, device
        self.flat = torch.empty(total, dtype=torch.float32, device=device)
        self.flat.copy_(torch.cat([src[n].detach().reshape(-1).float() for n in self.names]).to(device))
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=device)
        self.flat_c = None if dtype == torch.float32 else torch.empty(total, dtype=dtype, device=device)
        self.flat_t = torch.empty(toff, dtype=dtype, device=device)
        self._fresh = False
        self._merged = {}

        def leaf(o, shp):
            n = shp[0] * shp[1]
            p = self.flat[o:o + n].view(shp).requires_grad_(True)
            p.grad = self.flat_grad[o:o + n].view(shp)
            return p

        self.params = {n: leaf(o, shp) for n, (o, shp) in zip(self.names, shapes)}
        comp = self.flat if self.flat_c is None else self.flat_c
        self.groups, self.group_views, self._leaves, tiles = [], [], list(self.params.values()), []
        for gi, g in enumerate(layout):
            r, cin, n = g["rank"], g["cin"], g["n"]
            dcat = leaf(g["down_off"], (n * r, cin))
            ups = [leaf(o, shp) for o, shp in g["ups"]]
            self._leaves += [dcat] + ups
            self.groups.append(LoRAGroup(self, gi, dcat, ups, r, scale))
            cv = lambda o, shp: comp[o:o + shp[0] * shp[1]].view(shp)
            uts = [self.flat_t[to:to + shp[0] * shp[1]].view(shp[1], shp[0]) for to, (_, shp) in zip(g["ut_offs"], g["ups"])]
            self.group_views.append((cv(g["down_off"], (n * r, cin)), [cv(o, shp) for o, shp in g["ups"]],
                                     self.flat_t[g["t_off"]:g["t_off"] + n * r * cin].view(cin, n * r), uts))
            for r0 in range(0, n * r, 32):
                for c0 in range(0, cin, 32):
                    tiles.append((g["down_off"], g["t_off"], n * r, cin, r0, c0))
            for to, (o, shp) in zip(g["ut_offs"], g["ups"]):
                for r0 in range(0, shp[0], 32):
                    for c0 in range(0, shp[1], 32):
                        tiles.append((o, to, shp[0], shp[1], r0, c0))
        self._tiles = torch.tensor(tiles, dtype=torch.int64).to(device)

    def ensure_compute_copy(self):
        if not self._fresh:
            k = kernels()
            if self.flat_c is not None:
                k.unary(UN_COPY, self.flat, self.flat_c, self.flat.numel())
            k.transpose_cast_tiles(self.flat, self.flat_t, self._tiles)
            self._fresh = True
            self.epoch = getattr(self, "epoch", 0) + 1  # consumers that cache products of the copies compare this
            if self._merged:  # merged weights that exist follow the parameters
                self._merge_entries()

    def mark_updated(self):
        """call after an in-place update of `flat` (optimizer kernel): the derived copies are refreshed lazily."""
        self._fresh = False

    # ---- merged weights W + s U D (see lora_group_linear / _LoRAMergedLinear) ---------------------------------------
    def merged_weights(self, grp, lins):
        """([W_i + s U_i D_i], [their transposes]) in the compute dtype for the projections `lins` of group `grp`: persistent
        buffers (one [G, N, K] + one [G, K, N] allocation when the frozen weights are co-allocated), created at the first
        use and refreshed in place whenever the compute copies are (once per optimizer step), so captured graphs can read
        them.  Memory: a second and third copy of every LoRA'd attention weight (SD1.5: 2 x 186 MB per UNet)."""
        self.ensure_compute_copy()
        key = (grp.index, tuple(id(l) for l in lins))
        ent = self._merged.get(key)
        if ent is None:
            if self.device.type == "cuda" and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("merged LoRA weights must exist before a capture begins (run the call eagerly once)")
            G = len(lins)
            shp = tuple(lins[0].w.shape)
            if G > 1 and all(tuple(l.w.shape) == shp for l in lins):
                wm = list(lins[0].w.new_empty((G,) + shp).unbind(0))
                wmt = list(lins[0].w.new_empty((G, shp[1], shp[0])).unbind(0))
            else:
                wm = [torch.empty_like(l.w) for l in lins]
                wmt = [torch.empty_like(l.wt) for l in lins]
            ent = self._merged[key] = dict(grp=grp, lins=tuple(lins), wm=wm, wmt=wmt)
            self._build_merge_table()
            self._merge_entries([ent])
        return ent["wm"], ent["wmt"]

    def _entry_problems(self, ent):
        """rows of comat_lora_merge's problem table for one entry, or None when the grouped kernel cannot take it"""
        grp, lins = ent["grp"], ent["lins"]
        _, ucs, dct, _ = self.group_views[grp.index]
        r, Gr = grp.rank, grp.size * grp.rank
        k = kernels()
        if not hasattr(k, "lora_merge"):
            return None
        rows = []
        for i, lin in enumerate(lins):
            dt_i = dct[:, i * r:(i + 1) * r]
            if not k.lora_merge_ok(lin.w, ucs[i], dt_i, r, r, Gr):
                return None
            N, Kd = lin.w.shape
            rows.append((lin.w.data_ptr(), ucs[i].data_ptr(), dt_i.data_ptr(), ent["wm"][i].data_ptr(), ent["wmt"][i].data_ptr(),
                         N, Kd, r, r, Gr))
        return rows

    def _build_merge_table(self):
        """device tables of ONE comat_lora_merge launch over every merged entry the grouped kernel takes (rebuilt whenever
        an entry is added - eagerly: the refresh inside a captured step only reads them)"""
        import numpy as np
        # one launch, one scale: the groups of a store share it (LoRAStore(scale=...) hands the same value to every group)
        assert all(g.scale == self.groups[0].scale for g in self.groups), "LoRA groups of one store must share their scale"
        probs, tiles, rest = [], [], []
        for ent in self._merged.values():
            rows = self._entry_problems(ent)
            if rows is None:
                rest.append(ent)
                continue
            for row in rows:
                pi, N, Kd = len(probs), row[5], row[6]
                probs.append(row)
                n0, k0 = np.meshgrid(np.arange(0, N, 64), np.arange(0, Kd, 64), indexing="ij")
                tiles.append(np.stack([np.full(n0.size, pi), n0.reshape(-1), k0.reshape(-1)], 1))
        self._merge_rest = rest
        # a graph captured earlier replays comat_lora_merge with the addresses of the table it saw: superseded tables stay alive
        # (a few KB each; entries are only ever added while the model's first eager step runs)
        old = getattr(self, "_merge_table", None)
        if old is not None:
            self._merge_tables_kept = getattr(self, "_merge_tables_kept", []) + [old]
        if probs:
            self._merge_table = (torch.tensor(probs, dtype=torch.int64).to(self.device),
                                 torch.from_numpy(np.concatenate(tiles).astype(np.int32)).to(self.device))
        else:
            self._merge_table = None

    def _merge_entries(self, ents=None):
        """refresh the merged weights: every entry (ents None: one grouped launch + the entries it cannot take) or just
        `ents` (a new entry).  One scale per store (LoRAStore(scale=...) hands the same value to every group)."""
        k = kernels()
        if ents is None:
            if getattr(self, "_merge_table", None) is not None:
                k.lora_merge(self._merge_table[0], self._merge_table[1], self.groups[0].scale)
            for ent in getattr(self, "_merge_rest", ()):
                self._merge_into(ent)
            return
        for ent in ents:
            rows = self._entry_problems(ent)
            if rows is None:
                self._merge_into(ent)
                continue
            import numpy as np
            tiles = []
            for pi, row in enumerate(rows):
                n0, k0 = np.meshgrid(np.arange(0, row[5], 64), np.arange(0, row[6], 64), indexing="ij")
                tiles.append(np.stack([np.full(n0.size, pi), n0.reshape(-1), k0.reshape(-1)], 1))
            k.lora_merge(torch.tensor(rows, dtype=torch.int64).to(self.device),
                         torch.from_numpy(np.concatenate(tiles).astype(np.int32)).to(self.device), ent["grp"].scale)

    def _merge_into(self, ent):
        """one entry through comat_gemm (fp32 parity mode, shapes the grouped kernel does not take): Wm_i = W_i + s U_i D_i
        and WmT_i = W_i^T + s D_i^T U_i^T, a batched launch each for a co-allocated group"""
        grp, lins, wm, wmt = ent["grp"], ent["lins"], ent["wm"], ent["wmt"]
        _, ucs, dct, _ = self.group_views[grp.index]
        G, r = grp.size, grp.rank
        Gr = G * r
        k = kernels()
        N, Kd = lins[0].w.shape
        sw, su, sm = _uniform_stride([l.w for l in lins]), _uniform_stride(ucs), _uniform_stride(wm)
        swt, smt = _uniform_stride([l.wt for l in lins]), _uniform_stride(wmt)
        if G > 1 and None not in (sw, su, sm, swt, smt):
            # Wm_i[N, K] = W_i + s * U_i[N, r] (D^T[K, G*r] columns i*r..)^T  for all i in one batched launch
            k.gemm(ucs[0], dct, wm[0], N, Kd, r, r, Gr, Kd, batch=(G, 1), sA=(su, 0), sB=(r, 0), sC=(sm, 0),
                   R=lins[0].w, ldr=Kd, sR=(sw, 0), alpha=grp.scale, beta=1.0)
            k.gemm(dct, ucs[0], wmt[0], Kd, N, r, Gr, r, N, batch=(G, 1), sA=(r, 0), sB=(su, 0), sC=(smt, 0),
                   R=lins[0].wt, ldr=N, sR=(swt, 0), alpha=grp.scale, beta=1.0)
        else:
            for i, lin in enumerate(lins):
                N, Kd = lin.w.shape
                k.gemm(ucs[i], dct[:, i * r:(i + 1) * r], wm[i], N, Kd, r, r, Gr, Kd, R=lin.w, ldr=Kd, alpha=grp.scale,
                       beta=1.0)
                k.gemm(dct[:, i * r:(i + 1) * r], ucs[i], wmt[i], Kd, N, r, Gr, r, N, R=lin.wt, ldr=N, alpha=grp.scale,
                       beta=1.0)

    def zero_grad(self):
        self.flat_grad.zero_()
        for p in self._leaves:  # keep the views bound (the GEMM epilogues accumulate into them in place)
            if p.grad is None:
                raise RuntimeError("LoRA .grad view was dropped")

    def set_requires_grad(self, flag: bool):
        for p in self._leaves:
            p.requires_grad_(flag)

    def state_dict(self):
        return {n: p.detach().clone() for n, p in self.params.items()}


# ----------------------------------------------------------------------------------------------------------------
# elementwise
# ----------------------------------------------------------------------------------------------------------------
def cast(x, dtype):
    """dtype conversion (no autograd)."""
    if x.dtype == dtype:
        return x
    x = _c(x)
    y = torch.empty_like(x, dtype=dtype)
    kernels().unary(UN_COPY, x, y, x.numel())
    return y


class _Cast(Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src_dtype = x.dtype
        return cast(x, dtype)

    @staticmethod
    def backward(ctx, g):
        return cast(_c(g), ctx.src_dtype), None


def cast_grad(x, dtype):
    return x if x.dtype == dtype else _Cast.apply(x, dtype)


class _Unary(Function):
    @staticmethod
    def forward(ctx, x, op):
        x = _c(x)
        y = torch.empty_like(x)
        kernels().unary(op, x, y, x.numel())
        ctx.save_for_backward(x)
        ctx.op = op
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        kernels().unary_bwd(ctx.op, _c(g), x, dx, x.numel())
        return dx, None


def silu(x):
    return _Unary.apply(x, UN_SILU)


def gelu(x):
    return _Unary.apply(x, UN_GELU)


class _Axpby(Function):
    @staticmethod
    def forward(ctx, x, y, a, b):
        x, y = _c(x), _c(y)
        out = torch.empty_like(x)
        kernels().axpby(a, x, b, y, out, x.numel())
        ctx.a, ctx.b = a, b
        ctx.ydtype = y.dtype
        return out

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        gx = gy = None
        if ctx.needs_input_grad[0]:
            if ctx.a == 1.0:
                gx = g
            else:
                gx = torch.empty_like(g)
                kernels().axpby(ctx.a, g, 0.0, None, gx, g.numel())
        if ctx.needs_input_grad[1]:
            if ctx.b == 1.0 and ctx.ydtype == g.dtype:
                gy = g
            else:
                gy = torch.empty_like(g, dtype=ctx.ydtype)
                kernels().axpby(ctx.b, g, 0.0, None, gy, g.numel())
        return gx, gy, None, None


def add(x, y, a=1.0, b=1.0):
    """a*x + b*y (same shape); result has x's dtype."""
    return _Axpby.apply(x, y, float(a), float(b))


class _Affine(Function):
    @staticmethod
    def forward(ctx, x, a, b):
        x = _c(x)
        y = torch.empty_like(x)
        kernels().unary(UN_AFFINE, x, y, x.numel(), a, b)
        ctx.a = a
        return y

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        dx = torch.empty_like(g)
        kernels().unary(UN_AFFINE, g, dx, g.numel(), ctx.a, 0.0)
        return dx, None, None


def affine(x, a, b):
    """a*x + b with scalars."""
    return _Affine.apply(x, float(a), float(b))


class _Geglu(Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        M, D2 = x.shape
        y = x.new_empty((M, D2 // 2))
        kernels().geglu_fwd(x, y, M, D2 // 2)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        kernels().geglu_bwd(_c(g), x, dx, x.shape[0], x.shape[1] // 2)
        return dx


def geglu(x):
    return _Geglu.apply(x)


# COMAT_GEGLU_FUSED=0: the projection and the GEGLU as two launches (A/B runs, tests); both forms use the interleaved layout
_geglu_fused = os.environ.get("COMAT_GEGLU_FUSED", "1") != "0"


# COMAT_GEGLU_BWD_FUSED=0: the GEGLU's gradient as its own launch behind ff.net.2's data-gradient GEMM (A/B runs)
_geglu_bwd_fused = os.environ.get("COMAT_GEGLU_BWD_FUSED", "1") != "0"


def set_geglu_fused(flag: bool):
    global _geglu_fused
    _geglu_fused = bool(flag)


def _geglu_linear_fwd(x, lin, need_pre, fp8_for=None):
    """(GEGLU(x W^T + b), pre-activations or None, None): ONE launch (the product leaves the GEMM epilogue; the pre-activations are
    stored only when a backward pass will read them) where the library's pipelined kernel takes the problem, else the GEMM and
    the interleaved-layout GEGLU kernel.
    fp8_for: the frozen layer that consumes the result.  Under the fp8 forward with delayed scaling the epilogue writes the e4m3
    bytes that layer multiplies INSTEAD of the bf16 product (comat_gemm_params::q8): -> (None, pre, (bytes, scale))."""
    M, Kd = x.shape
    D, N2 = lin.out_features, lin.pre_features
    k = kernels()
    if _use_fp8(lin, Kd):
        a, (w, sw) = fp8_act(x, lin), fp8_weight(lin)
        a, scales = a[0], (a[1], sw)
    else:
        a, w, scales = x, lin.w, None
    fused = _geglu_fused and x.dtype == torch.bfloat16 and k.geglu_gemm_ok(a, w, M, N2, Kd)
    site = _fp8_producer_site(fp8_for, D, x.device) if (fused and scales is not None and _fp8_geglu_q8) else None
    if site is not None:
        pre = x.new_empty((M, N2)) if need_pre else None
        q8 = torch.empty((M, D), dtype=torch.uint8, device=x.device)
        k.gemm(a, w, pre, M, N2, Kd, Kd, Kd, N2, bias=lin.bias, scales=scales, geglu=(None, need_pre), q8=(q8, site[0], site[1]))
        return None, pre, (q8, site[0])
    y = x.new_empty((M, D))
    if fused:
        pre = x.new_empty((M, N2)) if need_pre else None
        k.gemm(a, w, pre, M, N2, Kd, Kd, Kd, N2, bias=lin.bias, scales=scales, geglu=(y, need_pre))
    else:
        pre = x.new_empty((M, N2))
        k.gemm(a, w, pre, M, N2, Kd, Kd, Kd, N2, bias=lin.bias, scales=scales)
        k.geglu_il_fwd(pre, y, M, D)
    return y, (pre if need_pre else None), None


class _GegluLinear(Function):
    """y = GEGLU(x W^T + b) with the interleaved weight of FrozenGegluLinear (see _geglu_linear_fwd)."""

    @staticmethod
    def forward(ctx, x, lin):
        x = _c(x)
        y, pre, _ = _geglu_linear_fwd(x, lin, ctx.needs_input_grad[0])
        ctx.lin = lin
        if pre is not None:
            ctx.save_for_backward(pre)
        return y

    @staticmethod
    def backward(ctx, g):
        (pre,) = ctx.saved_tensors
        lin = ctx.lin
        M, N2 = pre.shape
        k = kernels()
        dpre = torch.empty_like(pre)
        k.geglu_il_bwd(_c(g), pre, dpre, M, lin.out_features)
        dx = pre.new_empty((M, lin.in_features))
        k.gemm(dpre, lin.wt, dx, M, lin.in_features, N2, N2, N2, lin.in_features)
        return dx, None


class _GegluFeedForward(Function):
    """h = GEGLU(x W1^T + b1) W2^T + b2 + residual - the feed-forward of a BasicTransformerBlock (`ff.net.0.proj` + GEGLU,
    `ff.net.2`; 3P diffusers FeedForward, reached from TrainableSDPipeline.py:144-150) as one autograd node, so that the
    backward pass can run the GEGLU's gradient in the epilogue of `ff.net.2`'s data-gradient GEMM (comat_gemm_params::epi2 = 3):
        forward   two launches (projection + GEGLU epilogue, projection + residual epilogue)  - as before
        backward  d pre = geglu'(g W2; pre) in ONE launch (was: the GEMM, a [M, D] round trip, the elementwise kernel), then
                  dx = d pre W1."""

    @staticmethod
    def forward(ctx, x, residual, ff1, ff2):
        x = _c(x)
        need = ctx.needs_input_grad[0]
        f, pre, f8q = _geglu_linear_fwd(x, ff1, need, fp8_for=ff2)  # (fp8 forward, delayed scales: e4m3 bytes instead of f)
        M, D = x.shape[0], ff1.out_features
        N = ff2.out_features
        y = x.new_empty((M, N))
        k = kernels()
        residual = _c(residual) if residual is not None else None
        beta = 1.0 if residual is not None else 0.0
        if _use_fp8(ff2, D):
            f8, sf = f8q if f8q is not None else fp8_act(f, ff2)
            w8, sw = fp8_weight(ff2)
            k.gemm(f8, w8, y, M, N, D, D, D, N, bias=ff2.bias, R=residual, ldr=N, beta=beta, scales=(sf, sw))
        else:
            k.gemm(f, ff2.w, y, M, N, D, D, D, N, bias=ff2.bias, R=residual, ldr=N, beta=beta)
        ctx.ff1, ctx.ff2 = ff1, ff2
        ctx.has_res = residual is not None
        if need:
            ctx.save_for_backward(pre)
        return y

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        ff1, ff2 = ctx.ff1, ctx.ff2
        dx = None
        if ctx.needs_input_grad[0]:
            (pre,) = ctx.saved_tensors
            M, N2 = pre.shape
            D, N = ff1.out_features, ff2.out_features
            k = kernels()
            dpre = torch.empty_like(pre)
            if _geglu_fused and _geglu_bwd_fused and pre.dtype == torch.bfloat16 and k.geglu_gemm_ok(g, ff2.wt, M, 2 * D, N):
                k.gemm(g, ff2.wt, dpre, M, D, N, N, N, N2, geglu=(pre, "bwd"))
            else:
                df = pre.new_empty((M, D))
                k.gemm(g, ff2.wt, df, M, D, N, N, N, D)
                k.geglu_il_bwd(df, pre, dpre, M, D)
            dx = pre.new_empty((M, ff1.in_features))
            k.gemm(dpre, ff1.wt, dx, M, ff1.in_features, N2, N2, N2, ff1.in_features)
        return dx, (g if ctx.has_res else None), None, None


def geglu_feed_forward(x, ff1: "FrozenGegluLinear", ff2: "FrozenLinear", residual=None):
    """GEGLU(x W1^T + b1) W2^T + b2 (+ residual): see _GegluFeedForward"""
    return _GegluFeedForward.apply(x, residual, ff1, ff2)


def geglu_linear(x, lin: "FrozenGegluLinear"):
    """GEGLU(x W^T + b) - the feed-forward's first projection and its gate in one operator (see _GegluLinear)"""
    return _GegluLinear.apply(x, lin)


def _copy_pair(items, rows):
    """two strided 2-D copies [(src, ld_src, dst, ld_dst, cols)] in one launch where the library takes them (16-byte rows)"""
    k = kernels()
    if hasattr(k, "copy2d_pair") and k.copy2d_pair_ok(items):
        k.copy2d_pair(items, rows)
    else:
        for src, ld_src, dst, ld_dst, cols in items:
            k.copy2d(src, ld_src, dst, ld_dst, rows, cols)


class _ConcatCols(Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a), _c(b)
        M, Ca = a.shape
        Cb = b.shape[1]
        out = a.new_empty((M, Ca + Cb))
        _copy_pair([(a, Ca, out, Ca + Cb, Ca), (b, Cb, out[:, Ca:], Ca + Cb, Cb)], M)
        ctx.ca, ctx.cb = Ca, Cb
        return out

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        M = g.shape[0]
        Ca, Cb = ctx.ca, ctx.cb
        k = kernels()
        ga = gb = None
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:
            ga, gb = g.new_empty((M, Ca)), g.new_empty((M, Cb))
            _copy_pair([(g, Ca + Cb, ga, Ca, Ca), (g[:, Ca:], Ca + Cb, gb, Cb, Cb)], M)
        elif ctx.needs_input_grad[0]:
            ga = g.new_empty((M, Ca))
            k.copy2d(g, Ca + Cb, ga, Ca, M, Ca)
        elif ctx.needs_input_grad[1]:
            gb = g.new_empty((M, Cb))
            k.copy2d(g[:, Ca:], Ca + Cb, gb, Cb, M, Cb)
        return ga, gb


def concat_cols(a, b):
    """channel concat of two [M, C*] token matrices (UNet skip connections)."""
    return _ConcatCols.apply(a, b)


class _ConcatRows(Function):
    """row concat: out = [a; b] (batch concat of token matrices)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a), _c(b)
        out = a.new_empty((a.shape[0] + b.shape[0], a.shape[1]))
        k = kernels()
        na, nb = a.numel(), b.numel()
        items = [(a, na, out[: a.shape[0]], na, na), (b, nb, out[a.shape[0]:], nb, nb)]
        if a.dtype == b.dtype and hasattr(k, "copy2d_pair") and k.copy2d_pair_ok(items):
            k.copy2d_pair(items, 1)  # both halves in one launch (one "row" each)
        else:
            k.unary(UN_COPY, a, out[: a.shape[0]], na)
            k.unary(UN_COPY, b, out[a.shape[0]:], nb)
        ctx.ma = a.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        return g[: ctx.ma], g[ctx.ma:]


def concat_rows(a, b):
    return _ConcatRows.apply(a, b)


class _AddRowVec(Function):
    @staticmethod
    def forward(ctx, x, v):
        x, v = _c(x), _c(v)
        out = torch.empty_like(x)
        kernels().add_rowvec(x, v, out, x.shape[0], x.shape[1])
        return out

    @staticmethod
    def backward(ctx, g):
        return g, None  # v is a frozen positional table


def add_rowvec(x, v):
    """x[r, :] + v[:] with a frozen vector/table v (x: [rows, cols], v: [cols])."""
    return _AddRowVec.apply(x, v)


# ----------------------------------------------------------------------------------------------------------------
# dense contractions
# ----------------------------------------------------------------------------------------------------------------
class _Linear(Function):
    @staticmethod
    def forward(ctx, x, residual, lin, act, out_dtype=None):
        x = _c(x)
        M, Kd = x.shape
        N = lin.out_features
        y = torch.empty((M, N), dtype=out_dtype or x.dtype, device=x.device)
        if residual is not None:
            residual = _c(residual)
        if _use_fp8(lin, Kd):
            k = kernels()
            x8, sx = fp8_act(x, lin)
            w8, sw = fp8_weight(lin)
            k.gemm(x8, w8, y, M, N, Kd, Kd, Kd, N, bias=lin.bias, R=residual, ldr=N,
                   beta=1.0 if residual is not None else 0.0, act=act, scales=(sx, sw))
        else:
            kernels().gemm(x, lin.w, y, M, N, Kd, Kd, Kd, N, bias=lin.bias, R=residual, ldr=N,
                           beta=1.0 if residual is not None else 0.0, act=act)
        ctx.lin = lin
        ctx.has_res = residual is not None
        ctx.shape = (M, N, Kd)
        assert act == ACT_NONE or not ctx.needs_input_grad[0], "fused activation is for no-grad calls only"
        return y

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        M, N, Kd = ctx.shape
        dx = None
        if ctx.needs_input_grad[0]:
            dx = g.new_empty((M, Kd))
            kernels().gemm(g, ctx.lin.wt, dx, M, Kd, N, N, N, Kd)
        return dx, (g if ctx.has_res else None), None, None, None


def linear(x, lin: FrozenLinear, residual=None, act=ACT_NONE, out_dtype=None):
    """y = x W^T + b (+ residual); W frozen.  `out_dtype` (no-grad use) lets the GEMM epilogue emit fp32 directly."""
    return _Linear.apply(x, residual, lin, act, out_dtype)


def _uniform_stride(ts):
    """Element stride between equally shaped, contiguous tensors laid out at a constant spacing inside ONE allocation
    (e.g. dQ / dK / dV of the fused attention backward, or the up factors of a LoRA group), else None."""
    if len(ts) < 2:
        return None
    t0 = ts[0]
    step = ts[1].data_ptr() - t0.data_ptr()
    if step <= 0 or step % t0.element_size():
        return None
    base = t0.untyped_storage().data_ptr()
    for i, t in enumerate(ts):
        if (t.shape != t0.shape or t.dtype != t0.dtype or not t.is_contiguous()
                or t.untyped_storage().data_ptr() != base or t.data_ptr() - t0.data_ptr() != i * step):
            return None
    return step // t0.element_size()


class _LoRAGroupLinear(Function):
    """(y_1 .. y_G) with y_i = x W_i^T + b_i + (h_i) U_i^T (+ residual),  h = s * x [D_1; ..; D_G]^T.
    Forward: one GEMM for h, then ONE K-segmented GEMM per projection ([x | h_i] . [W_i | U_i]^T).
    Backward: u_i = s * g_i U_i (G small GEMMs into one [M, G*r] buffer), dx = sum_i g_i W_i + u [D_1; ..; D_G] as
    ONE K-segmented GEMM, and the LoRA weight gradients in fp32 straight out of the GEMM epilogue, accumulated in
    place into the flat gradient buffer (training_utils/pipeline.py:123-144 keeps LoRA params in fp32) on the side
    stream: dU_i += g_i^T h_i, d[D_1; ..; D_G] += u^T x (one GEMM)."""

    @staticmethod
    def forward(ctx, x, residual, grp, lins, down_cat, *ups):
        x = _c(x)
        M, Kd = x.shape
        G, r = grp.size, grp.rank
        Gr = G * r
        dc, ucs, dct, uts = grp.compute_copies()
        k = kernels()
        h = x.new_empty((M, Gr))
        if residual is not None:
            assert G == 1
            residual = _c(residual)
        sw, su = _uniform_stride([lin.w for lin in lins]), _uniform_stride(ucs)
        use8 = [_use_fp8(lin, Kd) for lin in lins]
        k.gemm(x, dc, h, M, Gr, Kd, Kd, Kd, Gr, alpha=grp.scale)
        g8 = fp8_weight_group(lins) if (all(use8) and G > 1 and su is not None and all(lin.bias is None for lin in lins)
                                        and r % 16 == 0 and Gr % 8 == 0) else None
        if g8 is not None:
            # q / k / v (k / v) of one attention: ONE batched fp8 product (shared input bytes, a scale per frozen weight) and ONE
            # batched low-rank product on top of it - 3 launches for the group instead of 1 + 2 G
            x8, sx = fp8_act(x, lins[0])
            N = lins[0].out_features
            ys = x.new_empty((G, M, N))
            if _fp8_ktail:
                k.gemm(x8, g8[0], ys, M, N, Kd, Kd, Kd, N, batch=(G, 1), sB=(N * Kd, 0), sC=(M * N, 0), scales=(sx, g8[1], 1),
                       ktail=(h, ucs[0], r, Gr, r, r, su))
            else:
                k.gemm(x8, g8[0], ys, M, N, Kd, Kd, Kd, N, batch=(G, 1), sB=(N * Kd, 0), sC=(M * N, 0), scales=(sx, g8[1], 1))
                k.gemm(h, ucs[0], ys, M, N, r, Gr, r, N, batch=(G, 1), sA=(r, 0), sB=(su, 0), sC=(M * N, 0), R=ys, ldr=N,
                       sR=(M * N, 0), beta=1.0)
            ys = list(ys.unbind(0))
        elif any(use8):
            # frozen part on the fp8 MFMA (x quantised once for the whole group), low-rank part added in the storage dtype
            x8, sx = fp8_act(x, lins[use8.index(True)])
            ys = []
            for i, lin in enumerate(lins):
                N = lin.out_features
                y = x.new_empty((M, N))
                beta = 1.0 if residual is not None else 0.0
                if use8[i] and _fp8_ktail and r % 16 == 0 and Gr % 8 == 0:
                    # frozen product (e4m3 MFMA) + low-rank product (bf16 MFMA, k-tail) + bias + residual: one launch
                    w8, sw8 = fp8_weight(lin)
                    k.gemm(x8, w8, y, M, N, Kd, Kd, Kd, N, bias=lin.bias, R=residual, ldr=N, beta=beta, scales=(sx, sw8),
                           ktail=(h[:, i * r:(i + 1) * r], ucs[i], r, Gr, r, 0, 0))
                elif use8[i]:
                    w8, sw8 = fp8_weight(lin)
                    k.gemm(x8, w8, y, M, N, Kd, Kd, Kd, N, bias=lin.bias, R=residual, ldr=N, beta=beta, scales=(sx, sw8))
                    k.gemm(h[:, i * r:(i + 1) * r], ucs[i], y, M, N, r, Gr, r, N, R=y, ldr=N, beta=1.0)
                else:
                    k.gemm_segments([(x, lin.w, Kd, Kd, Kd), (h[:, i * r:(i + 1) * r], ucs[i], r, Gr, r)], y, M, N, N,
                                    bias=lin.bias, R=residual, ldr=N, beta=beta)
                ys.append(y)
        elif sw is not None and su is not None and residual is None and all(lin.bias is None for lin in lins):
            # co-allocated frozen weights (frozen_linear_group) + adjacent up factors: ONE batched launch for the group
            N = lins[0].out_features
            ys = x.new_empty((G, M, N))
            k.gemm_segments([(x, lins[0].w, Kd, Kd, Kd, 0, sw), (h, ucs[0], r, Gr, r, r, su)], ys, M, N, N, batch=G,
                            sC=M * N)
            ys = list(ys.unbind(0))
        else:
            ys = []
            for i, lin in enumerate(lins):
                N = lin.out_features
                y = x.new_empty((M, N))
                k.gemm_segments([(x, lin.w, Kd, Kd, Kd), (h[:, i * r:(i + 1) * r], ucs[i], r, Gr, r)], y, M, N, N,
                                bias=lin.bias, R=residual, ldr=N, beta=1.0 if residual is not None else 0.0)
                ys.append(y)
        ctx.save_for_backward(x, h, dct, *uts)
        ctx.grp, ctx.lins = grp, lins
        ctx.has_res = residual is not None
        assert down_cat.grad is not None and all(u.grad is not None for u in ups), \
            "LoRA factors need preallocated .grad views"
        return tuple(ys)

    @staticmethod
    def backward(ctx, *gs):
        x, h, dct, *uts = ctx.saved_tensors
        grp, lins = ctx.grp, ctx.lins
        M, Kd = x.shape
        G, r = grp.size, grp.rank
        Gr = G * r
        k = kernels()
        gs = [_c(g) if g is not None else x.new_zeros((M, lin.out_features)) for g, lin in zip(gs, lins)]
        u = x.new_empty((M, Gr))
        N0 = lins[0].out_features
        # when the incoming gradients sit at a constant spacing in one buffer (dQ/dK/dV of the fused attention
        # backward) and so do the up factors, the G per-projection GEMMs below are ONE batched launch each
        # u_i = s * g_i U_i through the transposed copies U_i^T [r, N] (refreshed with the other compute copies once per
        # optimizer step): both operands k-contiguous, i.e. the pipelined kernel instead of a k-major gather
        sg, su = _uniform_stride(gs), _uniform_stride(uts)
        sgu = _uniform_stride([grp.ups[i].grad for i in range(G)])
        batched = sg is not None and su is not None and sgu is not None
        # the input gradient's segments: every projection's frozen part, then the low-rank part that reads u
        segs = [(gs[i], lin.wt, lin.out_features, lin.out_features, lin.out_features) for i, lin in enumerate(lins)]
        segs.append((u, dct, Gr, Gr, Gr))
        dx = x.new_empty((M, Kd)) if ctx.needs_input_grad[0] else None
        if batched:  # u[:, i*r:(i+1)*r] = s * g_i U_i for all i
            k.gemm(gs[0], uts[0], u, M, r, N0, N0, N0, Gr, alpha=grp.scale, batch=(G, 1), sA=(sg, 0), sB=(su, 0),
                   sC=(r, 0))
        else:
            for i, lin in enumerate(lins):
                N = lin.out_features
                k.gemm(gs[i], uts[i], u[:, i * r:(i + 1) * r], M, r, N, N, N, Gr, alpha=grp.scale)
        want_down = ctx.needs_input_grad[4]
        want_ups = ctx.needs_input_grad[5:]

        # LoRA weight gradients: dU_i [N, r] += g_i^T h_i,  d[D_1; ..; D_G] [G*r, K] += u^T x
        probs = []
        for i, lin in enumerate(lins):
            if want_ups[i]:
                probs.append((gs[i], h[:, i * r:(i + 1) * r], grp.ups[i].grad, lin.out_features, r, M,
                              lin.out_features, Gr, r))
        if want_down:
            probs.append((u, x, grp.down_cat.grad, Gr, Kd, M, Gr, Kd, Kd))
        if probs and _tt_grouping and all(k.tt_group_ok(*pr) for pr in probs):
            _tt_enqueue(x.device, probs, (gs, h, u, x))
            probs = []

        def weight_grads():  # what the grouped kernel does not take (fp32 parity mode, odd shapes): one launch each
            if batched and all(want_ups):
                gu = grp.ups[0].grad
                k.gemm(gs[0], h, gu, N0, r, M, N0, Gr, r, transA=True, transB=True, R=gu, ldr=r, beta=1.0,
                       batch=(G, 1), sA=(sg, 0), sB=(r, 0), sC=(sgu, 0), sR=(sgu, 0))
            else:
                for i, lin in enumerate(lins):
                    if want_ups[i]:
                        N, gu = lin.out_features, grp.ups[i].grad
                        k.gemm(gs[i], h[:, i * r:(i + 1) * r], gu, N, r, M, N, Gr, r, transA=True, transB=True,
                               R=gu, ldr=r, beta=1.0)
            if want_down:
                gd = grp.down_cat.grad
                k.gemm(u, x, gd, Gr, Kd, M, Gr, Kd, Kd, transA=True, transB=True, R=gd, ldr=Kd, beta=1.0)

        if probs:
            side = _side_stream(x.device)
            if side is None:
                weight_grads()
            else:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    weight_grads()
                if not any(st is side for _, st in _side_dirty):
                    _side_dirty.append((x.device, side))
                _side_keep.append((gs, h, u, x))  # keep the operands alive until join_side_streams()
                _queue_join()
        if dx is not None:
            k.gemm_segments(segs, dx, M, Kd, Kd)
        return (dx, (gs[0] if ctx.has_res else None), None, None, None) + (None,) * G


# COMAT_LORA_TAIL (default 1, round 6): the rank-r products the factor gradients need ride in the launch that shares their A operand
# (comat_gemm_params::epi2 = 4, "tail columns"): h = s x D^T as r extra output columns of the forward projection, u = s g U as r
# extra columns of a single projection's data-gradient.  0 = both as launches of their own in front of the weight-gradient group
# (round 5).  Same operands, fp32 accumulation, same rounding: the results differ at most in summation order.
_lora_tail = os.environ.get("COMAT_LORA_TAIL", "1") != "0"


def set_lora_tail(flag: bool):
    global _lora_tail
    _lora_tail = bool(flag)


def _merged_forward(x, lins, grp, residual, h=None):
    """y_i = x (W_i + s U_i D_i)^T + b_i (+ residual): one plain GEMM per projection or one batched GEMM for a co-allocated
    group; no segments.  h [M, G r] (optional): receives s x [D_1; ..; D_G]^T from the SAME launches (tail columns of the
    products: the down factors' compute copies are the extra rows of B)."""
    x = _c(x)
    M, Kd = x.shape
    wm, _ = grp.store.merged_weights(grp, lins)
    k = kernels()
    G, r = len(lins), grp.rank
    Gr = G * r
    dc = grp.compute_copies()[0] if h is not None else None
    sm = _uniform_stride(wm)
    if G > 1 and sm is not None and residual is None and all(l.bias is None for l in lins):
        N = lins[0].out_features
        ys = x.new_empty((G, M, N))
        if h is None:
            k.gemm(x, wm[0], ys, M, N, Kd, Kd, Kd, N, batch=(G, 1), sA=(0, 0), sB=(sm, 0), sC=(M * N, 0))
        else:
            k.gemm(x, wm[0], ys, M, N + r, Kd, Kd, Kd, N, batch=(G, 1), sA=(0, 0), sB=(sm, 0), sC=(M * N, 0),
                   tail=(dc, h, r, Gr, r * Kd, r, grp.scale))
        return tuple(ys.unbind(0))
    ys = []
    if residual is not None:
        residual = _c(residual)
    for i, (lin, w) in enumerate(zip(lins, wm)):
        N = lin.out_features
        y = x.new_empty((M, N))
        beta = 1.0 if residual is not None else 0.0
        if h is None:
            k.gemm(x, w, y, M, N, Kd, Kd, Kd, N, bias=lin.bias, R=residual, ldr=N, beta=beta)
        else:
            k.gemm(x, w, y, M, N + r, Kd, Kd, Kd, N, bias=lin.bias, R=residual, ldr=N, beta=beta,
                   tail=(dc[i * r:(i + 1) * r], h[:, i * r:(i + 1) * r], r, Gr, 0, 0, grp.scale))
        ys.append(y)
    return tuple(ys)


# COMAT_TRAIN_MERGED (default 1, round 5): the TRAINED calls use the merged weights too.  0 = the low-rank form of rounds 1-4
# (_LoRAGroupLinear: h = s x D^T on the dependent chain, then a K-segmented product).
_train_merged = os.environ.get("COMAT_TRAIN_MERGED", "1") != "0"


def set_train_merged(flag: bool):
    global _train_merged
    _train_merged = bool(flag)


class _LoRAMergedLinear(Function):
    """(y_1 .. y_G) with y_i = x W_eff,i^T + b_i (+ residual),  W_eff,i = W_i + s U_i D_i  (LoRAStore.merged_weights: both
    orientations refreshed once per optimizer step) - the same function as _LoRAGroupLinear
    (training_utils/pipeline.py:94-115), arranged so that nothing of the low-rank branch sits on the dependent chain:
      forward   ONE plain (batched) GEMM;
      backward  dx = sum_i g_i W_eff,i on the issuing stream (one K-segmented GEMM over the g_i, no u segment);
                the factor gradients dU_i += g_i^T (s x D_i^T), d[D_1; ..] += (s g_i U_i)^T x need the two M x G r products
                h and u - nobody else reads them, so they are launched with their weight-gradient group on the side stream."""

    @staticmethod
    def forward(ctx, x, residual, grp, lins, down_cat, *ups):
        x = _c(x)
        # the up factors' gradients dU_i += g_i^T h_i need h = s x D^T: r extra columns of this launch (round 6) instead of a
        # launch of its own in the backward pass
        h = x.new_empty((x.shape[0], grp.size * grp.rank)) if _lora_tail and any(u.requires_grad for u in ups) else None
        ys = _merged_forward(x, lins, grp, residual, h)
        if h is None:
            ctx.save_for_backward(x)
        else:
            ctx.save_for_backward(x, h)
        ctx.grp, ctx.lins = grp, lins
        ctx.epoch = getattr(grp.store, "epoch", 0)  # the merged weights this forward multiplied by
        ctx.has_res = residual is not None
        assert down_cat.grad is not None and all(u.grad is not None for u in ups), \
            "LoRA factors need preallocated .grad views"
        return ys

    @staticmethod
    def backward(ctx, *gs):
        x, *rest = ctx.saved_tensors
        h = rest[0] if rest else None
        grp, lins = ctx.grp, ctx.lins
        M, Kd = x.shape
        G, r = grp.size, grp.rank
        Gr = G * r
        k = kernels()
        gs = [_c(g) if g is not None else x.new_zeros((M, lin.out_features)) for g, lin in zip(gs, lins)]
        want_down, want_ups = ctx.needs_input_grad[4], ctx.needs_input_grad[5:]
        dx = u = None
        u_done = False
        if ctx.needs_input_grad[0]:
            _, wmt = grp.store.merged_weights(grp, lins)
            # (merged_weights refreshes lazily: an optimizer step of this store between a forward and its backward would hand
            # the backward other weights than the forward used)
            assert getattr(grp.store, "epoch", 0) == ctx.epoch, \
                "LoRA factors were updated between a trained call's forward and its backward"
            dx = x.new_empty((M, Kd))
            if G == 1:
                N = lins[0].out_features
                if want_down and _lora_tail:  # u = s g U rides along: U^T [r, N] is the tail of W_eff^T [K, N]
                    u = x.new_empty((M, r))
                    k.gemm(gs[0], wmt[0], dx, M, Kd + r, N, N, N, Kd, tail=(grp.compute_copies()[3][0], u, r, r, 0, 0, grp.scale))
                    u_done = True
                else:
                    k.gemm(gs[0], wmt[0], dx, M, Kd, N, N, N, Kd)
            else:
                k.gemm_segments([(gs[i], wmt[i], lin.out_features, lin.out_features, lin.out_features)
                                 for i, lin in enumerate(lins)], dx, M, Kd, Kd)
        if want_down or any(want_ups):
            dc, _, _, uts = grp.compute_copies()
            h_done = h is not None
            if h is None and any(want_ups):
                h = x.new_empty((M, Gr))
            if u is None and want_down:
                u = x.new_empty((M, Gr))
            N0 = lins[0].out_features
            sg, su = _uniform_stride(gs), _uniform_stride(uts)

            def low_rank():  # whatever did not ride in a neighbour's launch: h = s x [D_1; ..]^T, u_i = s g_i U_i (through U_i^T [r, N])
                if h is not None and not h_done:
                    k.gemm(x, dc, h, M, Gr, Kd, Kd, Kd, Gr, alpha=grp.scale)
                if u is None or u_done:
                    return
                if G > 1 and sg is not None and su is not None:
                    k.gemm(gs[0], uts[0], u, M, r, N0, N0, N0, Gr, alpha=grp.scale, batch=(G, 1), sA=(sg, 0), sB=(su, 0),
                           sC=(r, 0))
                else:
                    for i, lin in enumerate(lins):
                        N = lin.out_features
                        k.gemm(gs[i], uts[i], u[:, i * r:(i + 1) * r], M, r, N, N, N, Gr, alpha=grp.scale)

            pre = low_rank if (not h_done and any(want_ups)) or (want_down and not u_done) else None
            probs = []
            for i, lin in enumerate(lins):
                if want_ups[i]:
                    probs.append((gs[i], h[:, i * r:(i + 1) * r], grp.ups[i].grad, lin.out_features, r, M,
                                  lin.out_features, Gr, r))
            if want_down:
                probs.append((u, x, grp.down_cat.grad, Gr, Kd, M, Gr, Kd, Kd))
            if _tt_grouping and all(k.tt_group_ok(*pr) for pr in probs):
                _tt_enqueue(x.device, probs, (gs, h, u, x), pre=pre)
            else:  # fp32 parity mode, odd shapes: one launch per gradient, still off the issuing stream

                def weight_grads():
                    if pre is not None:
                        pre()
                    for A, B, Cacc, Mp, Np, Kp, lda, ldb, ldc in probs:
                        k.gemm(A, B, Cacc, Mp, Np, Kp, lda, ldb, ldc, transA=True, transB=True, R=Cacc, ldr=ldc, beta=1.0)

                side = _side_stream(x.device)
                if side is None:
                    weight_grads()
                else:
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        weight_grads()
                    if not any(st is side for _, st in _side_dirty):
                        _side_dirty.append((x.device, side))
                    _side_keep.append((gs, h, u, x))  # keep the operands alive until join_side_streams()
                    _queue_join()
        return (dx, (gs[0] if ctx.has_res else None), None, None, None) + (None,) * G


def lora_group_linear(x, lins, grp: LoRAGroup | None, residual=None):
    """(x W_i^T + b_i + lora_i(x)) for the projections `lins` that share the input x; a tuple of len(lins).
    Merged weights W + s U D (refreshed once per optimizer step) serve the no-grad calls (COMAT_NOGRAD_MERGED, default 1 since
    round 4) and the trained calls (COMAT_TRAIN_MERGED, default 1 since round 5: _LoRAMergedLinear); 0 selects the low-rank
    products of rounds 1-3 (_LoRAGroupLinear)."""
    if grp is None:
        assert residual is None or len(lins) == 1
        return tuple(linear(x, lin, residual) for lin in lins)
    # Measured at full SD1.5 size (profiles/r04_f_nograd_merged.txt): against the fp32 forward the merged call is as accurate as
    # the unmerged one at every LoRA magnitude (1.32e-2 vs 1.34e-2 of the output; the share of the LoRA's own effect that is
    # lost: 3.6e-2 vs 3.7e-2 at |U| = 0.02, 0.297 vs 0.296 at a tenth of that - bf16 rounding of the ACTIVATIONS dominates both),
    # and a no-grad UNet forward takes 6.74 instead of 7.32 ms.
    # (not under fp8_forward: there the frozen part runs on e4m3 weights quantised once - a merged weight would have to be
    # re-quantised after every optimizer step)
    if _fp8_on:
        return _LoRAGroupLinear.apply(x, residual, grp, tuple(lins), grp.down_cat, *grp.ups)
    if not torch.is_grad_enabled():
        if os.environ.get("COMAT_NOGRAD_MERGED", "1") != "0":
            return _merged_forward(x, tuple(lins), grp, residual)
    elif _train_merged:
        return _LoRAMergedLinear.apply(x, residual, grp, tuple(lins), grp.down_cat, *grp.ups)
    return _LoRAGroupLinear.apply(x, residual, grp, tuple(lins), grp.down_cat, *grp.ups)


def lora_linear(x, lin: FrozenLinear, grp: LoRAGroup | None, residual=None):
    return lora_group_linear(x, (lin,), grp, residual)[0]


class _Conv(Function):
    @staticmethod
    def forward(ctx, x, residual, conv, B, H, W, ups, bias2):
        x = _c(x)
        Hs, Ws = H * ups, W * ups
        Ho = (Hs + 2 * conv.pad - conv.kh) // conv.stride + 1
        Wo = (Ws + 2 * conv.pad - conv.kw) // conv.stride + 1
        assert x.shape == (B * H * W, conv.cin), (x.shape, B, H, W, conv.cin)
        y = x.new_empty((B * Ho * Wo, conv.cout))
        if residual is not None:
            residual = _c(residual)
        if _use_fp8(conv, conv.cin):
            k = kernels()
            x8, sx = fp8_act(x, conv)
            w8, sw = fp8_weight(conv)
            k.conv2d(x8, w8, y, B, H, W, conv.cin, Ho, Wo, conv.cout, conv.kh, conv.kw, conv.stride, conv.pad, mode=0,
                     ups=ups, bias=conv.bias, bias2=bias2, R=residual, beta=1.0 if residual is not None else 0.0,
                     scales=(sx, sw))
        else:
            kernels().conv2d(x, conv.w, y, B, H, W, conv.cin, Ho, Wo, conv.cout, conv.kh, conv.kw, conv.stride, conv.pad,
                             mode=0, ups=ups, bias=conv.bias, bias2=bias2, R=residual,
                             beta=1.0 if residual is not None else 0.0)
        ctx.conv, ctx.geo = conv, (B, H, W, Ho, Wo, ups)
        ctx.has_res = residual is not None
        return y

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        conv = ctx.conv
        B, H, W, Ho, Wo, ups = ctx.geo
        dx = None
        if ctx.needs_input_grad[0]:
            k = kernels()
            padd = conv.kh - 1 - conv.pad
            if conv.stride == 1:
                Hs, Ws = H * ups, W * ups
                du = g.new_empty((B * Hs * Ws, conv.cin))
                k.conv2d(g, conv.wd, du, B, Ho, Wo, conv.cout, Hs, Ws, conv.cin, conv.kh, conv.kw, 1, padd, mode=0)
                if ups == 2:
                    dx = g.new_empty((B * H * W, conv.cin))
                    k.sumpool2x2(du, dx, B, H, W, conv.cin)
                else:
                    dx = du
            else:
                assert ups == 1
                dx = g.new_empty((B * H * W, conv.cin))
                k.conv2d(g, conv.wd, dx, B, Ho, Wo, conv.cout, H, W, conv.cin, conv.kh, conv.kw, conv.stride, padd,
                         mode=1)
        return dx, (g if ctx.has_res else None), None, None, None, None, None, None


def conv2d(x, conv: FrozenConv, B, H, W, ups=1, residual=None, bias2=None):
    """Channels-last conv (frozen weight).  x: [B*H*W, Cin] -> ([B*Ho*Wo, Cout]).  `ups=2` fuses a nearest 2x
    upsample of the input; `bias2` [B, Cout] fp32 is the per-sample time-embedding add (no gradient: the time
    embedding depends only on t and frozen weights); `residual` is added in the epilogue."""
    return _Conv.apply(x, residual, conv, B, H, W, ups, bias2)


def conv_out_hw(conv: FrozenConv, H, W, ups=1):
    return ((H * ups + 2 * conv.pad - conv.kh) // conv.stride + 1, (W * ups + 2 * conv.pad - conv.kw) // conv.stride + 1)


# ----------------------------------------------------------------------------------------------------------------
# normalisation
# ----------------------------------------------------------------------------------------------------------------
class _GroupNorm(Function):
    """y = GroupNorm(x) (+ SiLU).  With `fork`, the input is also handed back as a second output (an alias): a branch
    that bypasses the norm (a ResnetBlock's shortcut, the residual around a transformer block) reads THAT output, so
    both gradients of x arrive at this node and dx = norm_bwd(gy) + g_bypass is ONE kernel (the `add` operand of
    comat_groupnorm_bwd) instead of autograd's separate accumulation add."""

    @staticmethod
    def forward(ctx, x, gamma, beta, B, HW, G, eps, silu_, fork, fp8_for=None):
        x = _c(x)
        Cc = x.shape[1]
        assert x.shape[0] == B * HW
        y = torch.empty_like(x)
        stats = torch.empty((B, G, 2), dtype=torch.float32, device=x.device)
        k = kernels()
        site = _fp8_producer_site(fp8_for, Cc, x.device)
        if site is not None and k.groupnorm_fwd_q_ok(x, B, HW, Cc, G):
            # fp8 forward, delayed scaling: the e4m3 bytes of y for the layer it feeds leave the same launch (fp8_act finds them)
            q8 = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
            k.groupnorm_fwd_q(x, gamma, beta, y, stats, B, HW, Cc, G, eps, silu_, q8, site[0], site[1])
            y._fp8 = (q8, site[0])
        else:
            k.groupnorm_fwd(x, gamma, beta, y, stats, B, HW, Cc, G, eps, silu_)
        ctx.save_for_backward(x, gamma, beta, stats)
        ctx.cfg = (B, HW, Cc, G, silu_)
        ctx.set_materialize_grads(False)
        return (y, x.view_as(x)) if fork else y

    @staticmethod
    def backward(ctx, g, g_bypass=None):
        x, gamma, beta, stats = ctx.saved_tensors
        B, HW, Cc, G, silu_ = ctx.cfg
        if g is None:  # the normalised branch is unused: only the bypass gradient flows
            return (g_bypass,) + (None,) * 9
        dx = torch.empty_like(x)
        add = None if g_bypass is None else _c(g_bypass)
        kernels().groupnorm_bwd(_c(g), x, gamma, beta, stats, dx, B, HW, Cc, G, silu_, add=add)
        return (dx,) + (None,) * 9


def group_norm(x, gamma, beta, B, HW, G=32, eps=1e-5, silu=False, fp8_for=None):
    """fp8_for: the frozen layer (holder) this output feeds - under the fp8 forward with delayed scaling the kernel also stores
    the e4m3 bytes that layer will multiply (ops.fp8_act picks them up); ignored otherwise"""
    return _GroupNorm.apply(x, gamma, beta, B, HW, G, float(eps), bool(silu), False, fp8_for)


def group_norm_fork(x, gamma, beta, B, HW, G=32, eps=1e-5, silu=False, fp8_for=None):
    """(GroupNorm(x), x'): use x' for the branch that bypasses the norm (see _GroupNorm)."""
    return _GroupNorm.apply(x, gamma, beta, B, HW, G, float(eps), bool(silu), True, fp8_for)


class _LayerNorm(Function):
    """y = LayerNorm(x); `fork` as in _GroupNorm (the residual connection around a pre-norm sub-layer)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, fork, fp8_for=None):
        x = _c(x)
        M, Cc = x.shape
        y = torch.empty_like(x)
        stats = torch.empty((M, 2), dtype=torch.float32, device=x.device)
        k = kernels()
        site = _fp8_producer_site(fp8_for, Cc, x.device)
        if site is not None and k.layernorm_fwd_q_ok(x):
            q8 = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
            k.layernorm_fwd_q(x, gamma, beta, y, stats, M, Cc, eps, q8, site[0], site[1])
            y._fp8 = (q8, site[0])
        else:
            k.layernorm_fwd(x, gamma, beta, y, stats, M, Cc, eps)
        ctx.save_for_backward(x, gamma, stats)
        ctx.set_materialize_grads(False)
        return (y, x.view_as(x)) if fork else y

    @staticmethod
    def backward(ctx, g, g_bypass=None):
        x, gamma, stats = ctx.saved_tensors
        if g is None:
            return g_bypass, None, None, None, None, None
        dx = torch.empty_like(x)
        add = None if g_bypass is None else _c(g_bypass)
        kernels().layernorm_bwd(_c(g), x, gamma, stats, dx, x.shape[0], x.shape[1], add=add)
        return dx, None, None, None, None, None


def layer_norm(x, gamma, beta, eps=1e-5, fp8_for=None):
    return _LayerNorm.apply(x, gamma, beta, float(eps), False, fp8_for)


def layer_norm_fork(x, gamma, beta, eps=1e-5, fp8_for=None):
    """(LayerNorm(x), x'): use x' for the residual that bypasses the norm (see _GroupNorm); fp8_for as in group_norm."""
    return _LayerNorm.apply(x, gamma, beta, float(eps), True, fp8_for)


# ----------------------------------------------------------------------------------------------------------------
# attention with materialised probabilities (the map the reference's AttentionStore captures)
# ----------------------------------------------------------------------------------------------------------------
class _Attention(Function):
    """softmax(scale * Q K^T) V per (batch, head), heads addressed in place inside [tokens, heads*dim] matrices.
    Returns (O [B*Nq, H*d], P [B, H, Nq, Nk]).  P is a differentiable output: an upstream gradient on it (from the
    attribute-concentration loss) is added to dP before the softmax backward (attn_utils/tc_attn_utils.py:140-146)."""

    @staticmethod
    def forward(ctx, q, k_, v, B, Nq, Nk, H, d, scale, causal, key_mask):
        q, k_, v = _c(q), _c(k_), _c(v)
        K = kernels()
        dev = q.device
        HD = H * d
        S = torch.empty((B, H, Nq, Nk), dtype=torch.float32, device=dev)
        K.gemm(q, k_, S, Nq, Nk, d, HD, HD, Nk, batch=(B, H), sA=(Nq * HD, d), sB=(Nk * HD, d),
               sC=(H * Nq * Nk, Nq * Nk), alpha=scale)
        P = torch.empty((B, H, Nq, Nk), dtype=q.dtype, device=dev)
        K.softmax_fwd(S, P, B * H * Nq, Nk, q_len=Nq, causal=causal, causal_offset=Nk - Nq, key_mask=key_mask,
                      rows_per_mask=H * Nq)
        del S
        O = q.new_empty((B * Nq, HD))
        K.gemm(P, v, O, Nq, d, Nk, Nk, HD, HD, transB=True, batch=(B, H), sA=(H * Nq * Nk, Nq * Nk),
               sB=(Nk * HD, d), sC=(Nq * HD, d))
        ctx.save_for_backward(q, k_, v, P)
        ctx.cfg = (B, Nq, Nk, H, d, scale)
        ctx.set_materialize_grads(False)  # an unused probability output must not cost a zero-filled gradient
        return O, P

    @staticmethod
    def backward(ctx, gO, gP):
        q, k_, v, P = ctx.saved_tensors
        B, Nq, Nk, H, d, scale = ctx.cfg
        K = kernels()
        dev = q.device
        HD = H * d
        sP = (H * Nq * Nk, Nq * Nk)
        dV = dQ = dK = None
        if gO is None and gP is None:
            return (None,) * 11
        if gO is None:
            gO = torch.zeros((B * Nq, HD), dtype=q.dtype, device=dev)
        gO = _c(gO)
        # dP = gO V^T  (fp32)
        dP = torch.empty((B, H, Nq, Nk), dtype=torch.float32, device=dev)
        K.gemm(gO, v, dP, Nq, Nk, d, HD, HD, Nk, batch=(B, H), sA=(Nq * HD, d), sB=(Nk * HD, d), sC=sP)
        if gP is not None:
            gP = _c(gP)
            K.axpby(1.0, dP, 1.0, gP, dP, dP.numel())
        if ctx.needs_input_grad[2]:  # dV [Nk, d] = P^T gO
            dV = torch.empty_like(v)
            K.gemm(P, gO, dV, Nk, d, Nq, Nk, HD, HD, transA=True, transB=True, batch=(B, H), sA=sP,
                   sB=(Nq * HD, d), sC=(Nk * HD, d))
        dS = torch.empty((B, H, Nq, Nk), dtype=q.dtype, device=dev)
        K.softmax_bwd(P, dP, dS, B * H * Nq, Nk, scale)
        del dP
        if ctx.needs_input_grad[0]:  # dQ = dS K
            dQ = torch.empty_like(q)
            K.gemm(dS, k_, dQ, Nq, d, Nk, Nk, HD, HD, transB=True, batch=(B, H), sA=sP, sB=(Nk * HD, d),
                   sC=(Nq * HD, d))
        if ctx.needs_input_grad[1]:  # dK = dS^T Q
            dK = torch.empty_like(k_)
            K.gemm(dS, q, dK, Nk, d, Nq, Nk, HD, HD, transA=True, transB=True, batch=(B, H), sA=sP,
                   sB=(Nq * HD, d), sC=(Nk * HD, d))
        return dQ, dK, dV, None, None, None, None, None, None, None, None


class _FlashAttention(Function):
    """Fused attention (scores stay on chip); saves Q, K, V, O and the per-row log-sum-exp for the backward kernels."""

    @staticmethod
    def forward(ctx, q, k_, v, B, Nq, Nk, H, d, scale, fp8_for=None):
        q, k_, v = _c(q), _c(k_), _c(v)
        HD = H * d
        O = q.new_empty((B * Nq, HD))
        lse = torch.empty((B, H, Nq), dtype=torch.float32, device=q.device)
        site = _fp8_producer_site(fp8_for, HD, q.device) if _fp8_flash_q8 else None
        if site is not None:  # fp8 forward, delayed scaling: the e4m3 bytes for the output projection leave the same launch
            q8 = torch.empty((B * Nq, HD), dtype=torch.uint8, device=q.device)
            kernels().flash_attn_fwd(q, k_, v, O, lse, B, H, Nq, Nk, d, HD, HD, HD, HD, scale, q8=(q8, site[0], site[1]))
            O._fp8 = (q8, site[0])
        else:
            kernels().flash_attn_fwd(q, k_, v, O, lse, B, H, Nq, Nk, d, HD, HD, HD, HD, scale)
        ctx.save_for_backward(q, k_, v, O, lse)
        ctx.cfg = (B, Nq, Nk, H, d, scale)
        return O

    @staticmethod
    def backward(ctx, gO):
        q, k_, v, O, lse = ctx.saved_tensors
        B, Nq, Nk, H, d, scale = ctx.cfg
        HD = H * d
        gO = _c(gO)
        # one allocation, constant spacing: the q/k/v (or k/v) projections' backward batches over these gradients
        if q.shape == k_.shape:
            dQ, dK, dV = q.new_empty((3,) + tuple(q.shape)).unbind(0)
        else:
            dQ = torch.empty_like(q)
            dK, dV = k_.new_empty((2,) + tuple(k_.shape)).unbind(0)
        dbuf = torch.empty((B, H, Nq), dtype=torch.float32, device=q.device)
        kernels().flash_attn_bwd(q, k_, v, O, gO, lse, dbuf, dQ, dK, dV, B, H, Nq, Nk, d, HD, HD, HD, HD, scale)
        return dQ, dK, dV, None, None, None, None, None, None, None


def flash_ok(dim, dtype):
    return dim <= 160 and dim % (8 if dtype == torch.bfloat16 else 4) == 0


class _FusedQKVAttention(Function):
    """Self-attention whose q / k / v projections are one frozen Linear(d -> 3d) (BLIP's ViT, modeling_blip
    BlipAttention.qkv): ONE GEMM writes [M, 3D], the fused attention kernels read q / k / v as column slices of it
    (leading dimension 3D), the backward kernels write dQ / dK / dV into the column slices of one [M, 3D] buffer and
    ONE GEMM (K = 3D) turns it into the input gradient — 2 launches instead of 6, no gradient-accumulation adds."""

    @staticmethod
    def forward(ctx, x, lin, B, N, H, d, scale):
        x = _c(x)
        M, Kd = x.shape
        D = H * d
        assert lin.out_features == 3 * D
        k = kernels()
        qkv = x.new_empty((M, 3 * D))
        k.gemm(x, lin.w, qkv, M, 3 * D, Kd, Kd, Kd, 3 * D, bias=lin.bias)
        O = x.new_empty((M, D))
        lse = torch.empty((B, H, N), dtype=torch.float32, device=x.device)
        k.flash_attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], O, lse, B, H, N, N, d, 3 * D, 3 * D, 3 * D, D, scale)
        ctx.save_for_backward(qkv, O, lse)
        ctx.cfg = (lin, B, N, H, d, scale, Kd)
        return O

    @staticmethod
    def backward(ctx, gO):
        qkv, O, lse = ctx.saved_tensors
        lin, B, N, H, d, scale, Kd = ctx.cfg
        D = H * d
        M = qkv.shape[0]
        k = kernels()
        gO = _c(gO)
        dqkv = torch.empty_like(qkv)
        dbuf = torch.empty((B, H, N), dtype=torch.float32, device=qkv.device)
        k.flash_attn_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], O, gO, lse, dbuf, dqkv[:, :D], dqkv[:, D:2 * D],
                         dqkv[:, 2 * D:], B, H, N, N, d, 3 * D, 3 * D, 3 * D, D, scale)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = qkv.new_empty((M, Kd))
            k.gemm(dqkv, lin.wt, dx, M, Kd, 3 * D, 3 * D, 3 * D, Kd)
        return dx, None, None, None, None, None, None


def fused_qkv_attention(x, lin_qkv: "FrozenLinear", B, N, heads, scale=None):
    """x: [B*N, in] -> attention output [B*N, heads*dim] with q, k, v = split(x W_qkv^T + b_qkv)."""
    D = lin_qkv.out_features // 3
    dim = D // heads
    assert flash_ok(dim, x.dtype), "fused qkv attention needs a head dim the fused kernels support"
    return _FusedQKVAttention.apply(x, lin_qkv, B, N, heads, dim, float(scale if scale is not None else dim ** -0.5))


def attention(q, k, v, B, Nq, Nk, heads, dim, scale=None, causal=False, key_mask=None, need_probs=True, fp8_for=None):
    """q: [B*Nq, heads*dim], k/v: [B*Nk, heads*dim] -> (out [B*Nq, heads*dim], probs [B, heads, Nq, Nk] or None).
    `need_probs=False` selects the fused kernel (no probability map in HBM) when the layer allows it.
    fp8_for: the frozen projection that consumes the output (ops.group_norm): the fused kernel then also emits its e4m3 bytes."""
    if scale is None:
        scale = dim ** -0.5
    if not need_probs and not causal and key_mask is None and flash_ok(dim, q.dtype):
        return _FlashAttention.apply(q, k, v, B, Nq, Nk, heads, dim, float(scale), fp8_for), None
    return _Attention.apply(q, k, v, B, Nq, Nk, heads, dim, float(scale), bool(causal), key_mask)


# ----------------------------------------------------------------------------------------------------------------
# scheduler step
# ----------------------------------------------------------------------------------------------------------------
class _CfgDdpm(Function):
    @staticmethod
    def forward(ctx, x, eps2, z, s, cx, ce, sigma):
        x, eps2 = _c(x), _c(eps2)
        assert x.dtype == torch.float32 and eps2.numel() == 2 * x.numel()
        xp = torch.empty_like(x)
        kernels().cfg_ddpm_fwd(x, eps2, None if z is None else _c(z), xp, x.numel(), s, cx, ce, sigma)
        ctx.cfg = (s, cx, ce, eps2.dtype, eps2.shape)
        return xp

    @staticmethod
    def backward(ctx, g):
        s, cx, ce, edt, eshape = ctx.cfg
        g = _c(g)
        dx = torch.empty_like(g) if ctx.needs_input_grad[0] else None
        deps = torch.empty(eshape, dtype=edt, device=g.device)
        kernels().cfg_ddpm_bwd(g, dx, deps, g.numel(), s, cx, ce)
        return dx, (deps if ctx.needs_input_grad[1] else None), None, None, None, None, None


def cfg_ddpm_step(x, eps2, z, guidance, cx, ce, sigma):
    """x_prev = cx*x + ce*(e_u + s(e_c - e_u)) + sigma*z;  x fp32 [n], eps2 = [uncond; cond] in compute dtype."""
    return _CfgDdpm.apply(x, eps2, z, float(guidance), float(cx), float(ce), float(sigma))


# ----------------------------------------------------------------------------------------------------------------
# image path
# ----------------------------------------------------------------------------------------------------------------
class ResampleTables:
    """Device-resident sparse tap tables of a separable linear resampling operator and of its transpose."""

    def __init__(self, fwd, bwd, Hin, Win, Hout, Wout, device):
        # fwd / bwd: dicts with ystart, ywt, xstart, xwt (numpy), KT
        def put(d):
            return dict(ystart=torch.from_numpy(d["ystart"]).to(device), ywt=torch.from_numpy(d["ywt"]).to(device),
                        xstart=torch.from_numpy(d["xstart"]).to(device), xwt=torch.from_numpy(d["xwt"]).to(device),
                        KT=int(d["KT"]))
        self.fwd, self.bwd = put(fwd), put(bwd)
        self.Hin, self.Win, self.Hout, self.Wout = Hin, Win, Hout, Wout

    def static_copy(self, extra_taps=2):
        """A second table set with `extra_taps` spare (zero-weight) taps per row: the fixed-address tables that a
        captured hipGraph of the training step reads.  `load()` refills it with another crop's operator before a
        replay (the resampling kernel skips zero weights, so padded taps cost nothing)."""
        new = ResampleTables.__new__(ResampleTables)
        new.Hin, new.Win, new.Hout, new.Wout = self.Hin, self.Win, self.Hout, self.Wout

        def grow(d):
            KT = d["KT"] + extra_taps
            out = dict(ystart=d["ystart"].clone(), xstart=d["xstart"].clone(), KT=KT)
            for k in ("ywt", "xwt"):
                w = torch.zeros((d[k].shape[0], KT), dtype=d[k].dtype, device=d[k].device)
                w[:, : d["KT"]] = d[k]
                out[k] = w
            return out
        new.fwd, new.bwd = grow(self.fwd), grow(self.bwd)
        return new

    def load(self, other):
        """copy `other`'s operator into these (fixed-address) tables; raises if it needs more taps than there is room"""
        for mine, theirs in ((self.fwd, other.fwd), (self.bwd, other.bwd)):
            if theirs["KT"] > mine["KT"]:
                raise ValueError("resampling operator needs more taps than the static tables hold")
            mine["ystart"].copy_(theirs["ystart"])
            mine["xstart"].copy_(theirs["xstart"])
            for k in ("ywt", "xwt"):
                mine[k].zero_()
                mine[k][:, : theirs["KT"]] = theirs[k]


class _Resample(Function):
    @staticmethod
    def forward(ctx, img, tab, B, Cc, scale, shift, out_dtype):
        img = _c(img)
        out = torch.empty((B * tab.Hout * tab.Wout, Cc), dtype=out_dtype, device=img.device)
        t = tab.fwd
        kernels().resample2d(img, out, B, tab.Hin, tab.Win, tab.Hout, tab.Wout, Cc, t["ystart"], t["ywt"], t["xstart"],
                             t["xwt"], t["KT"], scale, shift)
        ctx.tab, ctx.B, ctx.C, ctx.scale = tab, B, Cc, scale
        ctx.in_dtype = img.dtype
        return out

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        tab = ctx.tab
        t = tab.bwd
        dimg = torch.empty((ctx.B * tab.Hin * tab.Win, ctx.C), dtype=ctx.in_dtype, device=g.device)
        kernels().resample2d(g, dimg, ctx.B, tab.Hout, tab.Wout, tab.Hin, tab.Win, ctx.C, t["ystart"], t["ywt"],
                             t["xstart"], t["xwt"], t["KT"], ctx.scale, None)
        return dimg, None, None, None, None, None, None


def resample(img, tab: ResampleTables, B, C, scale=None, shift=None, out_dtype=None):
    """out = scale[c] * R(img) + shift[c] with R the separable resampling operator of `tab`."""
    return _Resample.apply(img, tab, B, C, scale, shift, out_dtype or img.dtype)


class _Patchify(Function):
    @staticmethod
    def forward(ctx, img, B, H, W, Cc, P):
        img = _c(img)
        out = img.new_empty((B * (H // P) * (W // P), P * P * Cc))
        kernels().patchify(img, out, B, H, W, Cc, P, False)
        ctx.cfg = (B, H, W, Cc, P)
        return out

    @staticmethod
    def backward(ctx, g):
        B, H, W, Cc, P = ctx.cfg
        g = _c(g)
        dimg = g.new_empty((B * H * W, Cc))
        kernels().patchify(dimg, g, B, H, W, Cc, P, True)
        return dimg, None, None, None, None, None


def patchify(img, B, H, W, C, P):
    return _Patchify.apply(img, B, H, W, C, P)


def embedding(ids, table):
    """rows of a frozen table (no gradient)."""
    ids = _c(ids.reshape(-1))
    out = table.new_empty((ids.numel(), table.shape[1]))
    kernels().embedding(ids, table, out, ids.numel(), table.shape[1], table.shape[0])
    return out


class _Permute(Function):
    @staticmethod
    def forward(ctx, x, B, Cc, H, W, to_nhwc, out_dtype):
        x = _c(x)
        y = torch.empty(((B * H * W, Cc) if to_nhwc else (B, Cc, H, W)), dtype=out_dtype, device=x.device)
        kernels().permute_nchw_nhwc(x, y, B, Cc, H, W, to_nhwc)
        ctx.cfg = (B, Cc, H, W, to_nhwc, x.dtype)
        return y

    @staticmethod
    def backward(ctx, g):
        B, Cc, H, W, to_nhwc, xdt = ctx.cfg
        g = _c(g)
        dx = torch.empty(((B, Cc, H, W) if to_nhwc else (B * H * W, Cc)), dtype=xdt, device=g.device)
        kernels().permute_nchw_nhwc(g, dx, B, Cc, H, W, not to_nhwc)
        return dx, None, None, None, None, None, None


def nchw_to_tokens(x, out_dtype=None):
    B, Cc, H, W = x.shape
    return _Permute.apply(x, B, Cc, H, W, True, out_dtype or x.dtype)


def tokens_to_nchw(x, B, H, W, out_dtype=None):
    return _Permute.apply(x, B, x.shape[1], H, W, False, out_dtype or x.dtype)


# ----------------------------------------------------------------------------------------------------------------
# losses
# ----------------------------------------------------------------------------------------------------------------
class _CrossEntropy(Function):
    @staticmethod
    def forward(ctx, logits, labels, ignore_index, ls):
        logits = _c(logits)
        T, V = logits.shape
        dev = logits.device
        logp = torch.empty((T,), dtype=torch.float32, device=dev)
        lse = torch.empty((T,), dtype=torch.float32, device=dev)
        acc = torch.empty((2,), dtype=torch.float32, device=dev)
        kernels().cross_entropy_fwd(logits, labels, logp, lse, acc, T, V, V, ignore_index, ls)
        ctx.save_for_backward(logits, labels, lse, acc)
        ctx.cfg = (ignore_index, ls)
        ctx.mark_non_differentiable(logp)
        return acc[0] / acc[1], logp

    @staticmethod
    def backward(ctx, g, _):
        logits, labels, lse, acc = ctx.saved_tensors
        ignore_index, ls = ctx.cfg
        T, V = logits.shape
        dl = torch.empty_like(logits)
        g_up = _c(g.reshape(1).to(torch.float32))  # device scalar: no host sync in the middle of backward
        kernels().cross_entropy_bwd(logits, labels, lse, dl, T, V, V, ignore_index, ls, g_up, acc)
        return dl, None, None, None


def cross_entropy(logits, labels, ignore_index=-100, label_smoothing=0.0):
    """mean token CE over labels != ignore_index.  Returns (loss, per-token log-prob of the label)."""
    return _CrossEntropy.apply(logits, labels, int(ignore_index), float(label_smoothing))


class _DiscHead(Function):
    @staticmethod
    def forward(ctx, x, w, b, target, pix_per_sample):
        x = _c(x)
        P = x.shape[0]
        assert x.shape[1] == 4
        loss = torch.empty((1,), dtype=torch.float32, device=x.device)
        kernels().disc_head_fwd(x, w, b, target, loss, P, pix_per_sample)
        ctx.save_for_backward(x, w, b, target)
        ctx.pps = pix_per_sample
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        x, w, b, target = ctx.saved_tensors
        P = x.shape[0]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dwb = dw = db = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dwb = torch.zeros((5,), dtype=torch.float32, device=x.device)
            dw, db = dwb[:4], dwb[4:]
        kernels().disc_head_bwd(x, w, b, target, _c(g.reshape(1).to(torch.float32)), dx, dwb, P, ctx.pps)
        return dx, dw, db, None, None


def disc_head_loss(x, w, b, target, pix_per_sample):
    """mean BCE-with-logits of Linear(4,1)(x) against target[pixel // pix_per_sample] (gan_sdxl.py:83-88)."""
    return _DiscHead.apply(x, w, b, target, int(pix_per_sample))


class _AttnMapGather(Function):
    @staticmethod
    def forward(ctx, amap, mask, tok_idx, tok_obj):
        amap = _c(amap)
        H, npix, L = amap.shape
        n_tok = tok_idx.numel()
        dev = amap.device
        num = torch.zeros((H, n_tok), dtype=torch.float32, device=dev)
        den = torch.zeros((H, n_tok), dtype=torch.float32, device=dev)
        avg = torch.zeros((n_tok, npix), dtype=torch.float32, device=dev)
        kernels().attnmap_gather_fwd(amap, mask, tok_idx, tok_obj, num, den, avg, H, npix, L, n_tok)
        ctx.save_for_backward(mask, tok_idx, tok_obj)
        ctx.cfg = (H, npix, L, n_tok, amap.dtype)
        return num, den, avg

    @staticmethod
    def backward(ctx, g_num, g_den, g_avg):
        mask, tok_idx, tok_obj = ctx.saved_tensors
        H, npix, L, n_tok, adt = ctx.cfg
        dev = mask.device
        z = lambda t, shape: torch.zeros(shape, dtype=torch.float32, device=dev) if t is None else _c(t)
        g_num, g_den = z(g_num, (H, n_tok)), z(g_den, (H, n_tok))
        g_avg = None if g_avg is None else _c(g_avg)
        damap = torch.empty((H, npix, L), dtype=adt, device=dev)  # the kernel writes every element
        kernels().attnmap_gather_bwd(g_num, g_den, g_avg, mask, tok_idx, tok_obj, damap, H, npix, L, n_tok)
        return damap, None, None, None


def attnmap_gather(amap, mask, tok_idx, tok_obj):
    """amap [heads, npix, L] -> (num [heads, n_tok], den [heads, n_tok], avg [n_tok, npix])."""
    return _AttnMapGather.apply(amap, mask, tok_idx, tok_obj)
