"""CPU restatement of the split-fp16 arithmetic ("f16x3") of the halo-patch kernels — TEST INFRASTRUCTURE, like everything under oracle/
(only tests/ may import it; the product path is auto-dynamic-deeplab_amd/csrc/conv3b.h, common.h).

An fp32 operand x is carried as two fp16 terms under an exact power-of-two scale s:  h = fp16(s x),  l = fp16(s x - h);  a product is the sum of its
three largest terms  l*wh + h*wl + h*wh  (fp16 x fp16 products are exact in fp32; accumulation in fp32 on the device, in fp64 here so that the SPLIT error
is seen alone).  The scale takes the operand's largest magnitude into [2^14, 2^15): `scale_field` restates common.h f16_scale_field bit for bit.
"""
import numpy as np


def scale_field(amax):
    """Exponent field (biased, 8 bits) of the scale of an operand whose largest magnitude is `amax` (float32): 268 - exponent field of amax, at most 253."""
    bits = np.float32(amax).view(np.uint32)
    return int(min(268 - int(bits >> 23), 253))


def pow2(field):
    return np.uint32(field << 23).view(np.float32)


def split(x, s):
    """(h, l) planes of s*x as float32 arrays holding fp16 values."""
    r = (x.astype(np.float32) * np.float32(s)).astype(np.float32)
    h = r.astype(np.float16).astype(np.float32)
    l = (r - h).astype(np.float16).astype(np.float32)
    return h, l


def matmul(w, x):
    """w [M, K] @ x [K, N] in the three-term split-fp16 arithmetic (exact accumulation), scales from the operands' maxima, unscaled result."""
    kw, kx = scale_field(np.abs(w).max()), scale_field(np.abs(x).max())
    wh, wl = split(w, pow2(kw))
    xh, xl = split(x, pow2(kx))
    acc = wl.astype(np.float64) @ xh.astype(np.float64) + wh.astype(np.float64) @ xl.astype(np.float64) + wh.astype(np.float64) @ xh.astype(np.float64)
    return acc * float(pow2(254 - kw)) * float(pow2(254 - kx))
