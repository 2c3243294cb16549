"""Grouped k-major weight gradients (comat_gemm_tt_grouped) against one comat_gemm launch per factor, on the weight-gradient
problem set of one SD1.5 trained UNet backward (CFG batch 2, 64x64 latents, LoRA rank 128 on all 32 attentions).

    python tools/mb_tt_group.py > gpurun_out/mb_tt_group.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from comat_amd import _hip  # noqa: E402

k = _hip.HipKernels()
dev = torch.device("cuda:0")
T = torch.bfloat16
R = 128


def unet_backward_problems():
    """(M, N, K) of every LoRA weight-gradient product of one trained SD1.5 UNet call, in backward order is irrelevant
    here: per transformer block with C channels over `tok` tokens (B = 2): self-attention q/k/v (dU x3, dD [3r, C]), out
    (dU, dD), cross-attention q (dU, dD), out (dU, dD)  [the text k/v projections run once per sampler call]"""
    probs = []
    for C, hw, nblk in ((320, 64, 5), (640, 32, 5), (1280, 16, 5), (1280, 8, 1)):
        tok = 2 * hw * hw
        for _ in range(nblk):
            probs += [(C, R, tok)] * 3 + [(3 * R, C, tok)]      # attn1 q/k/v
            probs += [(C, R, tok), (R, C, tok)]                 # attn1 out
            probs += [(C, R, tok), (R, C, tok)]                 # attn2 q
            probs += [(C, R, tok), (R, C, tok)]                 # attn2 out
    return probs


def graph_time(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fn()
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


shapes = unet_backward_problems()
ops = []
for M, N, K in shapes:
    A = torch.randn(K, M, device=dev).to(T)
    B = torch.randn(K, N, device=dev).to(T)
    C = torch.zeros(M, N, device=dev)
    ops.append((A, B, C, M, N, K, M, N, N))
fl = sum(2.0 * M * N * K for M, N, K in shapes)
by = sum((M + N) * K * 2 + 2 * M * N * 4 for M, N, K in shapes)
print(f"# {len(shapes)} weight-gradient problems of one SD1.5 trained UNet backward: {fl / 1e9:.1f} GFLOP, {by / 1e6:.0f} MB algorithmic")


def single():
    for A, B, C, M, N, K, lda, ldb, ldc in ops:
        k.gemm(A, B, C, M, N, K, lda, ldb, ldc, transA=True, transB=True, R=C, ldr=ldc, beta=1.0)


t1 = graph_time(single)
print(f"one launch per factor (gemm2_tt_kernel): {t1:8.3f} ms  = {fl / t1 / 1e9:7.1f} TFLOP/s, {t1 * 1e3 / len(ops):6.1f} us per problem")
for grp in (8, 16, 32, 48, len(ops)):
    def grouped():
        for i in range(0, len(ops), grp):
            k.gemm_tt_grouped(ops[i:i + grp])
    t = graph_time(grouped)
    print(f"groups of {grp:3d} ({(len(ops) + grp - 1) // grp:3d} calls):            {t:8.3f} ms  = {fl / t / 1e9:7.1f} TFLOP/s, "
          f"{by / t / 1e6:7.1f} GB/s algorithmic, {t * 1e3 / len(ops):6.1f} us per problem", flush=True)
# correctness spot check against fp32 matmul on the first and last problem
for idx in (0, len(ops) - 1):
    A, B, C, M, N, K = ops[idx][:6]
    C.zero_()
    k.gemm_tt_grouped([ops[idx]])
    ref = A.float().t() @ B.float()
    print(f"check problem {idx} {M}x{N}x{K}: max rel err {float((C - ref).abs().max() / ref.abs().max()):.2e}")
