"""TEST INFRASTRUCTURE (oracle): a numpy restatement of the Pillow operations the reference's training augmentation puts a
scene image through (/root/reference/mggan/data_utils/trajectories_scene.py:286-333,350-357 -> Pillow 12.2.0, the version in
this image; Pillow's source is not vendored under /root/reference):
    Image.transpose(FLIP_LEFT_RIGHT / FLIP_TOP_BOTTOM)
    Image.rotate(angle, expand=True)            libImaging Geometry.c: affine_fixed (nearest neighbour, 16.16 fixed point)
    Image.resize(size, LANCZOS)                 libImaging Resample.c: ImagingResample, 8-bit two-pass, 22-bit coefficients
    Image.crop(box) beyond the image            zero fill
PINNED against the installed Pillow itself, whole images, bit for bit (tests/test_aug_crops.py).  Only tests/ may import this;
the product computes the crops on the GPU (csrc/crop.hip) from the integer tables of mggan/data_utils/aug_geometry.py."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mg-gan_amd"))
from mggan.data_utils import aug_geometry as AG  # noqa: E402


def flip(img, code):
    """img (h, w, 3) uint8; code 0 none, 1 FLIP_LEFT_RIGHT, 2 FLIP_TOP_BOTTOM."""
    return img[:, ::-1] if code == 1 else img[::-1] if code == 2 else img


def rotate_expand(img, alpha):
    """Image.rotate(alpha / pi * 180, expand=True): nearest neighbour through affine_fixed, zero outside the source."""
    h, w = img.shape[:2]
    m, (nw, nh) = AG.rotate_matrix(w, h, alpha)
    if m == [1.0, 0.0, 0.0, 0.0, 1.0, 0.0]:
        return img.copy()
    a0, a1, a2, a3, a4, a5 = AG.affine_fixed(m, nw, nh)
    x = np.arange(nw, dtype=np.int64)[None, :]
    y = np.arange(nh, dtype=np.int64)[:, None]
    xin = (a2 + y * a1 + x * a0) >> 16
    yin = (a5 + y * a4 + x * a3) >> 16
    ok = (xin >= 0) & (xin < w) & (yin >= 0) & (yin < h)
    out = np.zeros((nh, nw, 3), np.uint8)
    out[ok] = img[yin[ok], xin[ok]]
    return out


def _clip8(v):
    return np.clip(v >> AG.PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_lanczos(img, size):
    """Image.resize((w2, h2), LANCZOS) of an RGB image: horizontal pass over the rows the vertical pass needs, then vertical."""
    h, w = img.shape[:2]
    w2, h2 = size
    if (w2, h2) == (w, h):
        return img.copy()
    src = img.astype(np.int64)
    half = 1 << (AG.PRECISION_BITS - 1)
    if w2 != w:
        bh, kh, _ = AG.resample_coeffs(w, w2)
        tmp = np.zeros((h, w2, 3), np.uint8)
        for X in range(w2):
            x0, n = bh[X]
            tmp[:, X] = _clip8(half + (src[:, x0:x0 + n] * kh[X, :n, None].astype(np.int64)).sum(1))
        src = tmp.astype(np.int64)
    if h2 != h:
        bv, kv, _ = AG.resample_coeffs(h, h2)
        out = np.zeros((h2, src.shape[1], 3), np.uint8)
        for Y in range(h2):
            y0, n = bv[Y]
            out[Y] = _clip8(half + (src[y0:y0 + n] * kv[Y, :n, None, None].astype(np.int64)).sum(0))
        return out
    return src.astype(np.uint8)


def crop(img, box):
    """Image.crop((x0, y0, x1, y1)): pixels outside the image read 0."""
    x0, y0, x1, y1 = box
    h, w = img.shape[:2]
    out = np.zeros((y1 - y0, x1 - x0, 3), np.uint8)
    sx0, sy0, sx1, sy1 = max(x0, 0), max(y0, 0), min(x1, w), min(y1, h)
    if sx1 > sx0 and sy1 > sy0:
        out[sy0 - y0:sy1 - y0, sx0 - x0:sx1 - x0] = img[sy0:sy1, sx0:sx1]
    return out


def augmented_small_image(scaled, code, alpha, f):
    """The `small_image` of a training item: flip -> rotate(expand) -> Lanczos resize by f."""
    rot = rotate_expand(np.ascontiguousarray(flip(scaled, code)), alpha)
    return resize_lanczos(rot, AG.small_size(rot.shape[1], rot.shape[0], f))
