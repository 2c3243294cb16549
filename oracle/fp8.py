"""ORACLE (test infrastructure, never imported by the product): CPU restatement of the per-tensor fp8 (OCP e4m3fn)
quantisation behind the fp8-forward configuration (BASELINE.json configs[4]: "fp8 MFMA UNet forward with bf16
backward").  The reference itself has no fp8 path (it trains under fp16 autocast, training_script.py:449-456): this
file restates the arithmetic of include/comat_hip.h (comat_fp8_scale / comat_fp8_quantize, comat_gemm with
COMAT_FP8_E4M3 operands), not a reference file.

Pinned by: (1) e4m3fn_round below is an independent numpy statement of OCP e4m3fn round-to-nearest-even (4 exponent
bits, bias 7, 3 mantissa bits, subnormals down to 2^-9, largest finite 448, no infinities) and is checked element by
element against torch's float8_e4m3fn cast over every representable value and their midpoints (tests/test_fp8.py);
(2) the GPU kernels are checked bit for bit against `quantize`.

  scale = max(amax|x|, 2^-100) / 448           (fp32)
  byte  = e4m3fn_rne(x * (1 / scale))          (fp32 reciprocal and product, saturating at +-448)
  value = scale * byte
Delayed scaling (round 6): the same byte rule under a scale taken from the abs-max of the tensors that passed the site during
the PREVIOUS optimizer step (`quantize_with_scale`); the scale rule itself is unchanged.
"""
from __future__ import annotations

import numpy as np
import torch

FP8_MAX = 448.0


def e4m3fn_round(v: np.ndarray) -> np.ndarray:
    """fp32 -> nearest e4m3fn value (ties to even), saturating; returned as fp32"""
    v = np.asarray(v, dtype=np.float32)
    a = np.minimum(np.abs(v).astype(np.float64), FP8_MAX)
    e = np.floor(np.log2(np.maximum(a, 2.0 ** -20)))
    e = np.maximum(e, -6.0)                      # below 2^-6 the format is subnormal: fixed quantum 2^-9
    quantum = 2.0 ** (e - 3.0)
    q = np.rint(a / quantum) * quantum           # np.rint rounds half to even
    q = np.minimum(q, FP8_MAX)
    return (np.sign(v) * q).astype(np.float32)


def scale_of(x: torch.Tensor) -> torch.Tensor:
    return torch.clamp(x.detach().float().abs().max(), min=2.0 ** -100) / FP8_MAX


def quantize(x: torch.Tensor):
    """-> (e4m3fn bytes as uint8 [same shape], scale fp32 scalar)"""
    s = scale_of(x)
    q = (x.detach().float() * (1.0 / s)).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn)
    return q.view(torch.uint8), s


def quantize_with_scale(x: torch.Tensor, s) -> torch.Tensor:
    """delayed scaling (include/comat_hip.h, ABI 8: comat_fp8_quantize_scaled and the *_fwd_q producers): the e4m3fn bytes of x
    under a scale that was fixed BEFORE x was seen (the site's abs-max of the previous optimizer step); values beyond
    448 * s saturate"""
    s = torch.as_tensor(s, dtype=torch.float32)
    q = (x.detach().float() * (1.0 / s)).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn)
    return q.view(torch.uint8)


def dequantize(q: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    return q.view(torch.float8_e4m3fn).float() * s


def fake_quant(x: torch.Tensor) -> torch.Tensor:
    """the fp32 values the fp8 MFMA multiplies (no gradient)"""
    q, s = quantize(x)
    return dequantize(q, s)
