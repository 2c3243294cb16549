"""Kernels for the rocprofv3 --pmc passes kept under profiles/: the dominant MFMA kernel (3x3 implicit-GEMM conv at the
SD1.5 64x64 level), a projection GEMM, and the attention-map path (softmax write-back of the cross-attention map +
the attribute-concentration gather), each launched a few times."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from comat_amd import _hip  # noqa: E402

K = _hip.HipKernels()
dev = torch.device("cuda:0")
T = torch.bfloat16
N = int(os.environ.get("PMC_N", "5"))

# calibration stream for the byte counters: out = 1.0 * x over 256 Mi bf16 elements = 512 MiB read + 512 MiB written
# (larger than the 256 MiB Infinity Cache)
CAL_N = 256 * 1024 * 1024
Xc = torch.ones(CAL_N, device=dev, dtype=T)
Yc = torch.empty_like(Xc)
for _ in range(N):
    K.axpby(1.0, Xc, 0.0, None, Yc, CAL_N)
del Xc, Yc
# conv 3x3, B=2, 64x64, 320 -> 320 (ResnetBlock2D at the top UNet level)
X = torch.randn(2 * 64 * 64, 320, device=dev).to(T)
W = torch.randn(320, 3, 3, 320, device=dev).to(T)
Y = torch.empty(2 * 64 * 64, 320, device=dev, dtype=T)
for _ in range(N):
    K.conv2d(X, W, Y, 2, 64, 64, 320, 64, 64, 320, 3, 3, 1, 1)
# conv 3x3, B=1, 128x128, 512 -> 512 (VAE decoder up block: the largest convs of the step)
Xv = torch.randn(128 * 128, 512, device=dev).to(T)
Wv = torch.randn(512, 3, 3, 512, device=dev).to(T)
Yv = torch.empty(128 * 128, 512, device=dev, dtype=T)
for _ in range(N):
    K.conv2d(Xv, Wv, Yv, 1, 128, 128, 512, 128, 128, 512, 3, 3, 1, 1)
# LoRA weight gradient dU [320, 128] += g^T h over 8192 tokens (k-major operands)
G_ = torch.randn(8192, 320, device=dev).to(T)
H_ = torch.randn(8192, 128, device=dev).to(T)
DU = torch.zeros(320, 128, device=dev)
for _ in range(N):
    K.gemm(G_, H_, DU, 320, 128, 8192, 320, 128, 128, transA=True, transB=True, R=DU, ldr=128, beta=1.0)
# fused attention, SD1.5 self-attention at the 64x64 level (2 x 8 heads, 4096 tokens, d = 40): forward + backward
HD = 8 * 40
Qa, Ka, Va, Ga = (torch.randn(2 * 4096, HD, device=dev).to(T) for _ in range(4))
Oa = torch.empty_like(Qa)
lse = torch.empty(2, 8, 4096, device=dev)
dbuf = torch.empty(2, 8, 4096, device=dev)
dQ, dK, dV = torch.empty_like(Qa), torch.empty_like(Ka), torch.empty_like(Va)
for _ in range(N):
    K.flash_attn_fwd(Qa, Ka, Va, Oa, lse, 2, 8, 4096, 4096, 40, HD, HD, HD, HD, 40 ** -0.5)
    K.flash_attn_bwd(Qa, Ka, Va, Oa, Ga, lse, dbuf, dQ, dK, dV, 2, 8, 4096, 4096, 40, HD, HD, HD, HD, 40 ** -0.5)
# GEGLU projection GEMM 8192 x 2560 x 320
A = torch.randn(8192, 320, device=dev).to(T)
Bw = torch.randn(2560, 320, device=dev).to(T)
C = torch.empty(8192, 2560, device=dev, dtype=T)
for _ in range(N):
    K.gemm(A, Bw, C, 8192, 2560, 320, 320, 320, 2560)
# cross-attention map path at up_64: scores [1 sample, 8 heads, 4096, 77] -> probabilities (the captured map), then gather
S = torch.randn(8, 4096, 77, device=dev)
P = torch.empty(8, 4096, 77, device=dev, dtype=T)
mask = (torch.rand(2, 4096, device=dev) > 0.5).float()
tok_idx = torch.tensor([2, 3, 6, 7], dtype=torch.int32, device=dev)
tok_obj = torch.tensor([0, 0, 1, 1], dtype=torch.int32, device=dev)
num = torch.zeros(8, 4, device=dev); den = torch.zeros(8, 4, device=dev); avg = torch.zeros(4, 4096, device=dev)
for _ in range(N):
    K.softmax_fwd(S, P, 8 * 4096, 77)
    K.attnmap_gather_fwd(P, mask, tok_idx, tok_obj, num, den, avg, 8, 4096, 77, 4)
torch.cuda.synchronize()
print("done")
