"""The split-fp16 arithmetic ("f16x3") restated on the CPU (oracle/f16x3.py): scale rule, range and split error.  No GPU."""
import numpy as np
import pytest

from oracle import f16x3


def test_scale_rule_keeps_every_magnitude_inside_fp16():
    """f16_scale_field: the largest magnitude lands in [2^14, 2^15) for every normal float32; zero / denormal maxima take the largest scale (2^126),
    Inf / NaN bit patterns a finite one; scale and inverse are normal numbers."""
    for e in range(-125, 128):
        for m in (1.0, 1.5, 1.9999999):
            a = np.float32(m) * np.float32(2.0) ** np.float32(e) if e < 127 else np.float32(m * 2.0 ** 126) * np.float32(2.0)
            if not np.isfinite(a):
                continue
            k = f16x3.scale_field(a)
            assert 13 <= k <= 253
            if k < 253:
                v = float(a) * float(f16x3.pow2(k))
                assert 2.0 ** 14 <= v < 2.0 ** 15, (a, k, v)
            assert float(f16x3.pow2(k)) * float(f16x3.pow2(254 - k)) == 1.0
    assert f16x3.scale_field(0.0) == 253 and f16x3.scale_field(np.float32(1e-42)) == 253
    assert f16x3.scale_field(np.float32(np.inf)) == 13


@pytest.mark.parametrize('case', ['unit', 'tiny_gradients', 'huge_activations', 'octaves'])
def test_split_error_is_below_fp32_accumulation_noise(case):
    """Three product terms of the two-term split: error / sum|a b| rms < 1e-8 and max < 5e-8 at K = 2736 — below what ONE fp32 rounding of the
    result costs (6e-8) — for operands of magnitude 1, 1e-9, 3e5; spread over 24 octaves (a few large terms carry each sum) rms < 4e-8, max < 2e-7;
    the fp32 BLAS product of the same data is worse in every case."""
    rng = np.random.default_rng(5)
    K, M, N = 2736, 32, 64
    x = rng.standard_normal((K, N)); w = rng.standard_normal((M, K)) * 0.03
    if case == 'tiny_gradients':
        x = x * 1e-9
    elif case == 'huge_activations':
        x = np.maximum(x, 0) * 3e5
    elif case == 'octaves':
        x = x * np.exp2(rng.integers(-12, 13, x.shape)); w = w * np.exp2(rng.integers(-12, 13, w.shape))
    x = x.astype(np.float32); w = w.astype(np.float32)
    ref = w.astype(np.float64) @ x.astype(np.float64)
    den = np.abs(w).astype(np.float64) @ np.abs(x).astype(np.float64)
    e = (f16x3.matmul(w, x) - ref) / den
    e32 = ((w @ x).astype(np.float64) - ref) / den
    rms_max, abs_max = (4e-8, 2e-7) if case == 'octaves' else (1e-8, 5e-8)
    assert np.sqrt(np.mean(e * e)) < rms_max and np.abs(e).max() < abs_max, (np.sqrt(np.mean(e * e)), np.abs(e).max())
    assert np.sqrt(np.mean(e * e)) < np.sqrt(np.mean(e32 * e32))


def test_unscaled_fp16_would_fail():
    """Why the scales exist: the same split WITHOUT them loses gradients of 1e-9 entirely (fp16's smallest subnormal is 6e-8)."""
    rng = np.random.default_rng(6)
    x = (rng.standard_normal((256, 8)) * 1e-9).astype(np.float32)
    h, l = f16x3.split(x, 1.0)
    assert not h.any() and not l.any()
    k = f16x3.scale_field(np.abs(x).max())
    h, l = f16x3.split(x, f16x3.pow2(k))
    back = (h.astype(np.float64) + l.astype(np.float64)) * float(f16x3.pow2(254 - k))
    assert np.abs(back - x).max() <= 2.0 ** -22 * np.abs(x).max()
