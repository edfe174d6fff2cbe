"""GPU input pipeline for Cityscapes samples (reference dataloaders/datasets/cityscapes.py:64-91,
dataloaders/custom_transforms.py:238-286 `train_preprocess`, :322-347 `full_image_eval_preprocess`).

The PNG decode stays on the host (there is no decoder on the device); everything after it runs on the GPU from the decoded
8-bit planes: `encode_segmap` (labelId -> train id, void -> 255), random left-right flip, random scale in [0.5, 2] with the
image resized by PIL's ANTIALIAS (= LANCZOS) filter and the labels by NEAREST, ToTensor + Normalize, zero / 255 padding to
the crop size and the random crop.  Results are BIT-EXACT with the reference's PIL calls: the resampling tables below are a
restatement of Pillow's `precompute_coeffs` / `normalize_coeffs_8bpc` (libImaging/Resample.c: fixed point with 22
fractional bits, rounding and clipping after the horizontal AND after the vertical pass) and of `ImagingScaleAffine`'s
nearest-neighbour index walk (libImaging/Geometry.c: the source coordinate is ACCUMULATED in double, not recomputed), and the
kernels (csrc/data.hip) only apply them.  tests/test_data_pipeline.py pins tables and kernels against PIL itself.

Random numbers: the reference draws `random.random()` (flip), `random.random()` (log-scale), `random.randint` x2 (crop) from
Python's global generator, in that order; pass a `random.Random` to reproduce a sequence.
"""
import ctypes as C
import math
import random as _random

import numpy as np
import torch

from . import _lib as L

PRECISION_BITS = 32 - 8 - 2
VOID_CLASSES = [0, 1, 2, 3, 4, 5, 6, 9, 10, 14, 15, 16, 18, 29, 30, -1]          # cityscapes.py:38
VALID_CLASSES = [7, 8, 11, 12, 13, 17, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 31, 32, 33]   # cityscapes.py:39
MEAN = (0.29866842, 0.30135223, 0.30561872)                                     # cityscapes.py:52-53
STD = (0.23925215, 0.23859318, 0.2385942)


def encode_segmap_lut(ignore_index=255):
    """cityscapes.py:83-90 as a 256-entry table: the reference's two in-place loops (void ids -> ignore_index, then valid id ->
    train id, in list order) run on the identity plane, so any chaining of the in-place assignments is reproduced."""
    m = np.arange(256, dtype=np.int64)
    for v in VOID_CLASSES:
        m[m == v] = ignore_index                # -1 never matches an 8-bit value, as in the reference
    for i, v in enumerate(VALID_CLASSES):
        m[m == v] = i
    return m.astype(np.uint8)


def _sinc(x):
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def _lanczos(x):
    return _sinc(x) * _sinc(x / 3.0) if -3.0 <= x < 3.0 else 0.0


def lanczos_tables(in_size, out_size):
    """Pillow precompute_coeffs + normalize_coeffs_8bpc for the LANCZOS filter (support 3) over the whole axis.
    Returns (bounds int32 [out, 2], coef int32 [out, ksize])."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 3.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coef = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [_lanczos((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        if ww != 0.0:
            k = [w / ww for w in k]
        for x, w in enumerate(k):
            coef[xx, x] = int(-0.5 + w * (1 << PRECISION_BITS)) if w < 0 else int(0.5 + w * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, coef


def nearest_table(in_size, out_size):
    """Pillow ImagingScaleAffine: xo starts at a*0.5 and is incremented by a = in/out per output pixel; index = int(xo)."""
    a = float(in_size) / out_size
    tab = np.zeros(out_size, dtype=np.int32)
    xo = a * 0.5
    for x in range(out_size):
        xin = -1 if xo < 0.0 else int(xo)
        tab[x] = min(max(xin, 0), in_size - 1)
        xo += a
    return tab


def resize_u8_host(img, new_w, new_h, mirror=False):
    """numpy restatement of `Image.resize((new_w, new_h), Image.ANTIALIAS)` on an [H, W, C] uint8 array (horizontal pass,
    then vertical, 8-bit intermediate) — the checker of the GPU kernels where PIL is not at hand, itself pinned to PIL by
    tests/test_data_pipeline.py."""
    H, W, Cc = img.shape
    src = img[:, ::-1] if mirror else img
    out = src
    if new_w != W:
        b, k = lanczos_tables(W, new_w)
        tmp = np.empty((H, new_w, Cc), dtype=np.uint8)
        for x in range(new_w):
            f, n = b[x]
            acc = (src[:, f:f + n].astype(np.int64) * k[x, :n].astype(np.int64)[None, :, None]).sum(1) + (1 << (PRECISION_BITS - 1))
            tmp[:, x] = np.clip(acc >> PRECISION_BITS, 0, 255)
        out = tmp
    if new_h != H:
        b, k = lanczos_tables(H, new_h)
        res = np.empty((new_h, out.shape[1], Cc), dtype=np.uint8)
        for y in range(new_h):
            f, n = b[y]
            acc = (out[f:f + n].astype(np.int64) * k[y, :n].astype(np.int64)[:, None, None]).sum(0) + (1 << (PRECISION_BITS - 1))
            res[y] = np.clip(acc >> PRECISION_BITS, 0, 255)
        out = res
    return out


def _dev_i32(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)


class GpuPreprocess:
    """train_preprocess / full_image_eval_preprocess on the device.  Inputs: decoded image [H, W, 3] uint8 and labelIds
    [H, W] uint8 (device tensors, or host arrays that are uploaded)."""

    def __init__(self, crop_size, mean=MEAN, std=STD, scale=0, device='cuda:0', rng=None):
        self.crop_size, self.scale = tuple(crop_size), scale
        self.dev = torch.device(device)
        self.lib = L.load()
        self.mean = torch.tensor(mean, dtype=torch.float32)
        self.std = torch.tensor(std, dtype=torch.float32)
        self.lut = torch.from_numpy(encode_segmap_lut()).to(self.dev)
        self.rng = rng if rng is not None else _random
        self._tables = {}

    def _tab(self, kind, a, b):
        key = (kind, a, b)
        t = self._tables.get(key)
        if t is None:
            if kind == 'lanczos':
                bn, k = lanczos_tables(a, b)
                t = (_dev_i32(bn, self.dev), _dev_i32(k, self.dev), k.shape[1])
            else:
                t = _dev_i32(nearest_table(a, b), self.dev)
            if len(self._tables) > 64:
                self._tables.clear()
            self._tables[key] = t
        return t

    def _upload(self, a):
        if not isinstance(a, torch.Tensor):
            a = torch.from_numpy(np.ascontiguousarray(a))
        return a.to(self.dev, non_blocking=True).contiguous()

    def encode_segmap(self, label_ids):
        lab = self._upload(label_ids)
        out = torch.empty_like(lab)
        st = torch.cuda.current_stream().cuda_stream
        L.check(self.lib.addk_lut_u8(lab.data_ptr(), out.data_ptr(), lab.numel(), self.lut.data_ptr(), st), 'lut_u8')
        return out

    def _finish(self, img, lab, i0, j0):
        ch, cw = self.crop_size
        H, W = img.shape[0], img.shape[1]
        out_img = torch.empty((3, ch, cw), dtype=torch.float32, device=self.dev)
        out_lbl = torch.empty((ch, cw), dtype=torch.int64, device=self.dev)
        st = torch.cuda.current_stream().cuda_stream
        m = (C.c_float * 3)(*self.mean.tolist()); s = (C.c_float * 3)(*self.std.tolist())
        L.check(self.lib.addk_finish_sample(img.data_ptr(), lab.data_ptr(), H, W, i0, j0, ch, cw, m, s, out_img.data_ptr(),
                                            out_lbl.data_ptr(), st), 'finish_sample')
        return {'image': out_img, 'label': out_lbl}

    def eval_sample(self, image, label_ids):
        """full_image_eval_preprocess (custom_transforms.py:322-347): normalise, pad to crop_size (image 0, label 255)."""
        img = self._upload(image)
        lab = self.encode_segmap(label_ids)
        H, W = img.shape[0], img.shape[1]
        ch, cw = max(self.crop_size[0], H), max(self.crop_size[1], W)
        keep = self.crop_size
        self.crop_size = (ch, cw)
        try:
            return self._finish(img, lab, 0, 0)
        finally:
            self.crop_size = keep

    def train_sample(self, image, label_ids):
        """train_preprocess (custom_transforms.py:238-286)."""
        img = self._upload(image)
        lab = self.encode_segmap(label_ids)
        H, W = img.shape[0], img.shape[1]
        flip = self.rng.random() < 0.5
        if self.scale == 0:
            lo, hi = 0.5, 2.0
            rand_log_scale = math.log(lo, 2) + self.rng.random() * (math.log(hi, 2) - math.log(lo, 2))
            s = math.pow(2, rand_log_scale)
        else:
            s = self.scale
        nw, nh = int(round(W * s)), int(round(H * s))
        st = torch.cuda.current_stream().cuda_stream
        lib = self.lib
        cur, cw_, ch_ = img, W, H
        if nw != W:
            bn, k, ks = self._tab('lanczos', W, nw)
            tmp = torch.empty((H, nw, 3), dtype=torch.uint8, device=self.dev)
            L.check(lib.addk_resample_u8(cur.data_ptr(), H, W, tmp.data_ptr(), H, nw, 3, bn.data_ptr(), k.data_ptr(), ks, 0, int(flip), st), 'resample_h')
            cur, cw_ = tmp, nw
        elif flip:
            cur = torch.flip(cur, dims=[1]).contiguous()
        if nh != H:
            bn, k, ks = self._tab('lanczos', H, nh)
            tmp = torch.empty((nh, cw_, 3), dtype=torch.uint8, device=self.dev)
            L.check(lib.addk_resample_u8(cur.data_ptr(), H, cw_, tmp.data_ptr(), nh, cw_, 3, bn.data_ptr(), k.data_ptr(), ks, 1, 0, st), 'resample_v')
            cur, ch_ = tmp, nh
        if (nw, nh) != (W, H):
            lab2 = torch.empty((nh, nw), dtype=torch.uint8, device=self.dev)
            L.check(lib.addk_nearest_u8(lab.data_ptr(), H, W, lab2.data_ptr(), nh, nw, self._tab('nearest', W, nw).data_ptr(),
                                        self._tab('nearest', H, nh).data_ptr(), int(flip), st), 'nearest')
            lab = lab2
        elif flip:
            lab = torch.flip(lab, dims=[1]).contiguous()
        ch, cw = self.crop_size
        hp, wp = max(nh, ch), max(nw, cw)                      # size after padding
        i0 = self.rng.randint(0, hp - ch)
        j0 = self.rng.randint(0, wp - cw)
        return self._finish(cur, lab, i0, j0)
