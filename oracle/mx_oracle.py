"""CPU ORACLE (test infrastructure only) for the MXFP8 format of the fp8 QKV/proj path (BASELINE config
#5): OCP Microscaling Formats v1.0 - blocks of 32 consecutive K elements share one e8m0 scale, the
smallest power of two with amax / scale <= 448 (= 2^(floor(log2 amax) - 8), one more when amax's mantissa
exceeds 1.75, so nothing saturates); elements are e4m3fn, round-to-nearest-even.  The reference has no fp8 path (fp32 throughout), so there is no golden vector for this format:
the restatement is pinned by (a) torch's own float8_e4m3fn conversion for the element rounding and (b)
exact-integer GEMM identities in tests/test_fp8.py; model-level parity of the fp8 mode is measured
against the same fp32 goldens as the bf16 mode."""
import numpy as np
import torch


def quantize_mx8(x):
    """x [..., K] float -> (q uint8 e4m3 bytes [..., K], scales uint8 e8m0 [..., K/32])."""
    x = torch.as_tensor(x, dtype=torch.float32)
    K = x.shape[-1]
    blk = x.reshape(*x.shape[:-1], K // 32, 32)
    amax = blk.abs().amax(dim=-1)
    bits = amax.view(torch.int32)
    e = ((bits >> 23) & 0xFF) + ((bits & 0x7FFFFF) > 0x600000).to(torch.int32)   # see the module docstring
    sb = torch.clamp(e - 8, min=0)
    inv = ((254 - sb) << 23).to(torch.int32).view(torch.float32)  # 2^(127 - sb)
    scaled = torch.clamp(blk * inv[..., None], -448.0, 448.0)
    q = scaled.to(torch.float8_e4m3fn).view(torch.uint8)
    return q.reshape(x.shape), sb.to(torch.uint8)


def dequantize_mx8(q, scales):
    q = torch.as_tensor(q).view(torch.float8_e4m3fn).to(torch.float32)
    K = q.shape[-1]
    s = torch.pow(2.0, torch.as_tensor(scales).to(torch.float32) - 127.0)
    return (q.reshape(*q.shape[:-1], K // 32, 32) * s[..., None]).reshape(q.shape)


def gemm_mx8(qa, sa, qb, sb, bias=None, alpha=1.0):
    """alpha * A . B^T + bias in float64 on the dequantised operands (what the scaled MFMA computes exactly,
    up to fp32 accumulation order)."""
    a, b = dequantize_mx8(qa, sa).double(), dequantize_mx8(qb, sb).double()
    c = alpha * (a @ b.transpose(-1, -2))
    return c + bias.double() if bias is not None else c
