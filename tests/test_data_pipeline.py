"""Input pipeline (reference dataloaders/datasets/cityscapes.py:64-91, dataloaders/custom_transforms.py:238-286,322-347).
The reference's transform module cannot be imported here (torchvision is absent and Pillow >= 10 dropped Image.ANTIALIAS), so
the parity anchor is PIL itself — the third-party code whose arithmetic the reference's calls execute: Pillow resize with
LANCZOS (== the old ANTIALIAS) and NEAREST.  CPU: the host restatement of Pillow's resampling tables is bit-exact with
PIL; encode_segmap equals the reference's two in-place loops.  GPU: the kernels reproduce a PIL-built train / eval sample
bit for bit (labels) and to the last float bit (normalised image)."""
import math
import random

import numpy as np
import pytest
import torch
from PIL import Image

from addk import data as D


def _img(h, w, seed):
    g = np.random.default_rng(seed)
    base = g.integers(0, 256, (h // 4 + 2, w // 4 + 2, 3), dtype=np.uint8)
    img = np.asarray(Image.fromarray(base).resize((w, h), Image.BILINEAR)).copy()      # smooth + noise: realistic 8-bit content
    img[::7, ::5] = g.integers(0, 256, img[::7, ::5].shape, dtype=np.uint8)
    lab = g.choice(np.array([0, 1, 7, 8, 11, 12, 13, 17, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 31, 32, 33, 4, 5, 6, 9, 10, 14, 15, 16, 18, 29, 30], dtype=np.uint8),
                   size=(h // 8 + 1, w // 8 + 1))
    lab = np.kron(lab, np.ones((8, 8), dtype=np.uint8))[:h, :w].copy()
    return img, lab


@pytest.mark.parametrize('size,new', [((37, 53), (26, 19)), ((37, 53), (53, 37)), ((64, 80), (160, 128)), ((40, 60), (61, 40)), ((33, 47), (47, 66))])
def test_lanczos_tables_are_bit_exact_with_pillow(size, new):
    h, w = size
    nw, nh = new
    img, _ = _img(h, w, 5)
    ref = np.asarray(Image.fromarray(img).resize((nw, nh), Image.LANCZOS))
    assert np.array_equal(D.resize_u8_host(img, nw, nh), ref)
    flipped = Image.fromarray(img).transpose(Image.FLIP_LEFT_RIGHT)
    assert np.array_equal(D.resize_u8_host(img, nw, nh, mirror=True), np.asarray(flipped.resize((nw, nh), Image.LANCZOS)))


@pytest.mark.parametrize('size,new', [((37, 53), (26, 19)), ((64, 80), (160, 128)), ((1024, 2048), (1399, 700)), ((33, 47), (95, 66))])
def test_nearest_table_is_bit_exact_with_pillow(size, new):
    h, w = size
    nw, nh = new
    _, lab = _img(h, w, 6)
    ref = np.asarray(Image.fromarray(lab).resize((nw, nh), Image.NEAREST))
    xt, yt = D.nearest_table(w, nw), D.nearest_table(h, nh)
    assert np.array_equal(lab[yt][:, xt], ref)


def test_encode_segmap_equals_the_reference_loops():
    lut = D.encode_segmap_lut()
    mask = np.arange(256, dtype=np.uint8).reshape(16, 16).copy()
    ref = mask.copy()
    for v in D.VOID_CLASSES:                    # cityscapes.py:83-90, verbatim semantics on a uint8 plane
        ref[ref == v] = 255
    cm = dict(zip(D.VALID_CLASSES, range(19)))
    for v in D.VALID_CLASSES:
        ref[ref == v] = cm[v]
    assert np.array_equal(lut[mask], ref)
    assert sorted(set(lut[D.VALID_CLASSES].tolist())) == list(range(19)) and all(lut[v] == 255 for v in D.VOID_CLASSES if v >= 0)


def _host_train_sample(img, lab_ids, crop, rng, mean, std):
    """train_preprocess (custom_transforms.py:238-286) with PIL and torch, ToTensor / Normalize written out."""
    lab = D.encode_segmap_lut()[lab_ids]
    image, mask = Image.fromarray(img), Image.fromarray(lab)
    if rng.random() < 0.5:
        image = image.transpose(Image.FLIP_LEFT_RIGHT)
        mask = mask.transpose(Image.FLIP_LEFT_RIGHT)
    w, h = image.size
    rls = math.log(0.5, 2) + rng.random() * (math.log(2.0, 2) - math.log(0.5, 2))
    s = math.pow(2, rls)
    new_size = (int(round(w * s)), int(round(h * s)))
    image = image.resize(new_size, Image.LANCZOS)
    mask = mask.resize(new_size, Image.NEAREST)
    t = torch.from_numpy(np.asarray(image).copy()).permute(2, 0, 1).float().div(255)
    t = t.sub(torch.tensor(mean).view(3, 1, 1)).div(torch.tensor(std).view(3, 1, 1))
    m = torch.from_numpy(np.asarray(mask).astype(np.int64))
    h, w = t.shape[1], t.shape[2]
    pt, pl = max(0, crop[0] - h), max(0, crop[1] - w)
    t = torch.nn.ZeroPad2d((0, pl, 0, pt))(t)
    m = torch.nn.ConstantPad2d((0, pl, 0, pt), 255)(m)
    h, w = t.shape[1], t.shape[2]
    i = rng.randint(0, h - crop[0]); j = rng.randint(0, w - crop[1])
    return t[:, i:i + crop[0], j:j + crop[1]], m[i:i + crop[0], j:j + crop[1]]


@pytest.mark.gpu
@pytest.mark.parametrize('seed', [1, 2, 3, 4, 5, 6])
def test_gpu_train_sample_is_bit_exact_with_a_pil_built_one(seed):
    import addk  # noqa: F401
    img, lab = _img(256, 512, 40 + seed)
    crop = (193, 193)
    ref_img, ref_lab = _host_train_sample(img, lab, crop, random.Random(seed), D.MEAN, D.STD)
    pp = D.GpuPreprocess(crop, rng=random.Random(seed))
    out = pp.train_sample(img, lab)
    torch.cuda.synchronize()
    assert torch.equal(out['label'].cpu(), ref_lab)
    assert torch.equal(out['image'].cpu(), ref_img)


@pytest.mark.gpu
def test_gpu_eval_sample_pads_to_1025x2049():
    import addk  # noqa: F401
    img, lab = _img(1024, 2048, 77)
    pp = D.GpuPreprocess((1025, 2049))
    out = pp.eval_sample(img, lab)
    torch.cuda.synchronize()
    t = torch.from_numpy(img).permute(2, 0, 1).float().div(255)
    t = t.sub(torch.tensor(D.MEAN).view(3, 1, 1)).div(torch.tensor(D.STD).view(3, 1, 1))
    t = torch.nn.ZeroPad2d((0, 1, 0, 1))(t)
    m = torch.nn.ConstantPad2d((0, 1, 0, 1), 255)(torch.from_numpy(D.encode_segmap_lut()[lab].astype(np.int64)))
    assert tuple(out['image'].shape) == (3, 1025, 2049) and torch.equal(out['image'].cpu(), t) and torch.equal(out['label'].cpu(), m)
