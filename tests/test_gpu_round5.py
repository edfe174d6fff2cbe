"""GPU parity added in round 5 (VERDICT r04 items 2, 5 and ADVICE r04):

 * config 2 at 2x1024x2048 in TRAIN mode (batch statistics in every BatchNorm, `train.py:227-240`): the 14 sentinel conv-weight gradients
   against the REFERENCE's own fp64 gradients (tests/golden/grads64.npz `full_train_sentinels`, written by tests/golden/make_golden_fp64.py from
   the real `modeling.ADD` in double precision), relative to the live fp32 oracle's error — the backward of the headline step held to a
   number instead of to the chaos argument;
 * the wide 1x1 weight gradient with 128 output channels (a shape both the split-bf16 head kernel and the register-streaming kernel accept:
   ADVICE r04, wgrad.hip wg_fill) against fp64, with the launch geometry the chosen kernel expects."""
import ctypes as C
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import test_gpu_round3 as R3          # noqa: E402  (sentinel_gate, the shared report)

REPORT = []


def _log(fmt, *a):
    REPORT.append(fmt % a)


def teardown_module(module):
    os.makedirs('gpurun_out', exist_ok=True)
    with open('gpurun_out/parity_report_round5.txt', 'w') as f:
        f.write('\n'.join(REPORT) + '\n')


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    import addk
    addk.load()
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    return torch.device('cuda:0')


def test_full_size_train_mode_gradients_on_sentinel_convs(dev):
    """The reference's own fp32 arithmetic sits 4e-2..6e-2 (max-abs and rms) from its fp64 gradients on the stems and the early cells at this
    shape, 1e-3..2e-2 on the heads (the fixture's d32 / stat arrays; profiles/r05_parity_report_round5.txt) — train-mode BatchNorm amplifies
    rounding differences by 10^4..10^5, so no fp32 implementation can be held elementwise to 1e-3 against fp64 here.  Asserted: addk's error
    against fp64 within max(3x the live fp32 oracle's, 1e-3) per sentinel in both metrics, and every element within the spread two fp32
    realisations show against each other."""
    res = R3.sentinel_gate(dev, 'full_train_sentinels', True, 3.0, 1e-3, 0.15, 0.10, log=_log)      # every element vs the fp32 oracle: measured max 6.0e-2, rms 5.4e-2 (two fp32 realisations, each 5e-2 from fp64)
    import math
    gm = math.exp(sum(math.log(max(ea, 1e-30) / max(eo, 1e-30)) for ea, eo, _ in res.values()) / len(res))
    _log('config2 2x1024x2048 train-mode sentinel gradients: geometric mean of addk / fp32-oracle error ratios %.2f', gm)
    assert gm <= 1.6, gm


@pytest.mark.parametrize('prec', ['bf16x6', 'f16x3'])
@pytest.mark.parametrize('shape', [(2, 64, 128, 320, 128), (1, 70, 130, 200, 128), (2, 64, 128, 400, 256)], ids=['c320_o128', 'c200_o128_odd', 'c400_o256'])
def test_wide_pointwise_weight_gradient_with_128_outputs(dev, shape, prec):
    """1x1, C -> 128 (F = 32's heads): kind_of() picks the split-bf16 head kernel (9) and wg_fill must give it ITS geometry —
    (Cout / 128) * cdiv(C, 64) tiles of 64-pixel row segments — not the register-streaming kernel's, which the same shape also passes."""
    import addk
    from addk import _lib as L
    lib = L.load()
    N, H, W, Ci, Cout = shape
    prev = addk.get_precision()
    try:
        addk.set_precision(prec)
        gen = torch.Generator().manual_seed(Ci + Cout + H)
        rnd = lambda *sh: torch.randn(*sh, generator=gen).to(dev)
        P = N * H * W
        x, a, b, dy = rnd(P, Ci), rnd(Ci), 0.3 * rnd(Ci), rnd(P, Cout)
        z = F.relu(a.double() * x.double() + b.double())
        ref = dy.double().t() @ z                              # [Cout][Ci]
        wa = L.ConvWgradArgs()
        wa.dy, wa.lddy, wa.Cout = dy.data_ptr(), Cout, Cout
        wa.N, wa.H, wa.W, wa.OH, wa.OW, wa.KH, wa.KW, wa.stride, wa.pad, wa.dil = N, H, W, H, W, 1, 1, 1, 0, 1
        wa.src.x, wa.src.a, wa.src.b, wa.src.ld, wa.src.C, wa.src.relu = x.data_ptr(), a.data_ptr(), b.data_ptr(), Ci, Ci, 1
        dw = torch.full((Cout, Ci), float('nan'), device=dev)
        wa.dw, wa.ldw, wa.cin_total, wa.w_choff, wa.accumulate = dw.data_ptr(), Ci, Ci, 0, 0
        wa.ws_floats = lib.addk_conv_wgrad_ws(P, Cout, Ci, 1)
        ws = torch.empty(int(wa.ws_floats), device=dev)
        wa.ws = ws.data_ptr()
        cfg = (C.c_int32 * 4)()
        L.check(lib.addk_conv_wgrad_config(C.byref(wa), cfg), 'wgrad_config')
        tiles = (Cout // 128) * ((Ci + 63) // 64)
        assert cfg[0] == 9 and cfg[1] == 8 and cfg[2] == 4 and cfg[3] % tiles == 0 and cfg[3] // tiles <= N * H * ((W + 63) // 64), list(cfg)
        L.check(lib.addk_conv_wgrad(C.byref(wa), torch.cuda.current_stream().cuda_stream), 'conv_wgrad')
        torch.cuda.synchronize()
        e = float((dw.double().cpu() - ref.cpu()).abs().max() / ref.abs().max())
        _log('wide 1x1 weight gradient %s [%s]: kind %d, %d workgroups, error vs fp64 %.2e', shape, prec, cfg[0], cfg[3], e)
        assert e <= 2e-5, e
    finally:
        addk.set_precision(prev)
