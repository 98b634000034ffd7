"""Host-side geometry of the training augmentation's image path, for the device crop kernel (csrc/crop.hip:
crop_patches_aug_kernel).  The reference transforms the scene image per ITEM with Pillow
(/root/reference/mggan/data_utils/trajectories_scene.py:276-333): flip -> `img.rotate(alpha / pi * 180, expand=True)` (nearest
neighbour) -> `img.resize(..., ANTIALIAS)` (Lanczos, 8-bit fixed point) -> one 33 x 33 crop per pedestrian (:350-357).  The GPU
computes only the crops; what it needs per item are the integers Pillow itself would use:

  * the inverse affine map of the rotation in 16.16 fixed point -- Image.rotate's matrix (Python floats, cosine / sine rounded to
    15 decimals, the expand translation) put through the FIX() of Pillow's nearest-neighbour affine loop (libImaging
    Geometry.c: affine_fixed), and the size of the expanded canvas;
  * the Lanczos coefficient tables of the two resize passes -- libImaging Resample.c: precompute_coeffs + normalize_coeffs_8bpc
    (f64 weights from libm's sin, normalised, rounded to 22-bit fixed point), one row of `ksize` integers and a
    (first source index, tap count) pair per output column / row.  They depend on (source size, output size) only and are cached.

Pillow is a third-party dependency of the reference (no source under /root/reference); the restatement is pinned against the
installed Pillow itself: tests/test_aug_crops.py compares whole rotated / resized images bit for bit."""
import math
from functools import lru_cache

import numpy as np

PRECISION_BITS = 32 - 8 - 2  # Resample.c: coefficients of the 8-bit paths are 22-bit fixed point
LANCZOS_SUPPORT = 3.0


def rotate_matrix(w, h, alpha):
    """Image.rotate(alpha / pi * 180, expand=True) of a (w, h) image -> (matrix [a, b, c, d, e, f] mapping OUTPUT pixel
    coordinates to source coordinates, (nw, nh) of the expanded canvas).  alpha == 0 -> identity, same size (Pillow's
    fast path returns a copy); the other fast paths (exactly 90 / 180 / 270 degrees) are transposes and not built."""
    angle = (alpha / np.pi * 180) % 360.0
    if angle == 0:
        return [1.0, 0.0, 0.0, 0.0, 1.0, 0.0], (w, h)
    if angle in (90, 180, 270):
        raise NotImplementedError("rotation by exactly {} degrees is a transpose in Pillow".format(angle))
    center = (w / 2, h / 2)
    angle = -math.radians(angle)
    m = [round(math.cos(angle), 15), round(math.sin(angle), 15), 0.0, round(-math.sin(angle), 15), round(math.cos(angle), 15), 0.0]

    def transform(x, y, m):
        a, b, c, d, e, f = m
        return a * x + b * y + c, d * x + e * y + f

    m[2], m[5] = transform(-center[0], -center[1], m)
    m[2] += center[0]
    m[5] += center[1]
    xx, yy = [], []
    for x, y in ((0, 0), (w, 0), (w, h), (0, h)):
        tx, ty = transform(x, y, m)
        xx.append(tx)
        yy.append(ty)
    nw = math.ceil(max(xx)) - math.floor(min(xx))
    nh = math.ceil(max(yy)) - math.floor(min(yy))
    m[2], m[5] = transform(-(nw - w) / 2.0, -(nh - h) / 2.0, m)
    return m, (nw, nh)


def _floor_c(v):
    """Geometry.c: #define FLOOR(v) ((v) < 0.0 ? ((int)floor(v)) : ((int)(v)))"""
    return int(math.floor(v)) if v < 0.0 else int(v)


def affine_fixed(m, nw, nh):
    """The six 16.16 integers of affine_fixed (Geometry.c) for an (nw, nh) output: source x of output pixel (x, y) is
    (a2 + y * a1 + x * a0) >> 16, source y is (a5 + y * a4 + x * a3) >> 16.  Raises when Pillow itself would leave the
    fixed-point loop (coordinates beyond 32,768) or take the pure-scaling loop (no rotation)."""
    a = m
    if a[1] == 0 and a[3] == 0:
        raise NotImplementedError("a map without rotation takes Pillow's scaling loop")

    def check(x, y):
        return abs(x * a[0] + y * a[1] + a[2]) < 32768.0 and abs(x * a[3] + y * a[4] + a[5]) < 32768.0

    if not (check(0, 0) and check(nw, nh) and check(0, nh) and check(nw, 0)):
        raise NotImplementedError("image too large for Pillow's fixed-point affine loop")
    fix = lambda v: _floor_c(v * 65536.0 + 0.5)
    return (fix(a[0]), fix(a[1]), fix(a[2] + a[0] * 0.5 + a[1] * 0.5), fix(a[3]), fix(a[4]), fix(a[5] + a[3] * 0.5 + a[4] * 0.5))


def _sin_libm(a):
    """sin of every element through libm (math.sin) -- the function Pillow calls; numpy's vectorised sine may differ in the
    last bit, and a coefficient that sits on a rounding boundary would flip."""
    return np.array(list(map(math.sin, a.ravel().tolist())), np.float64).reshape(a.shape)


def _sinc(t):
    tp = t * math.pi
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.where(t == 0.0, 1.0, _sin_libm(tp) / tp)


@lru_cache(maxsize=4096)
def resample_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs(in_size, 0, in_size, out_size, LANCZOS) + normalize_coeffs_8bpc ->
    (bounds int32 (out_size, 2) = [first source index, taps], kk int32 (out_size, ksize), ksize).  Every operation is the C
    loop's, element by element, in its order (IEEE f64: sums accumulate left to right -- np.cumsum --, sin is libm's)."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = LANCZOS_SUPPORT * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)       # (int) truncates; the value is >= -support > INT_MIN
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    x = np.arange(ksize, dtype=np.int64)[None, :]
    live = x < xmax[:, None]
    arg = ((x + xmin[:, None]).astype(np.float64) - center[:, None] + 0.5) * ss
    w = np.where(live & (arg >= -3.0) & (arg < 3.0), _sinc(arg) * _sinc(arg / 3), 0.0)  # truncated sinc (lanczos_filter)
    ww = np.cumsum(w, axis=1)[:, -1:]                                      # ww += w, left to right (dead taps add 0.0)
    w = np.where(ww != 0.0, w / np.where(ww != 0.0, ww, 1.0), w)
    half = float(1 << PRECISION_BITS)
    kk = np.where(w < 0, np.trunc(-0.5 + w * half), np.trunc(0.5 + w * half)).astype(np.int32)
    kk[~live] = 0
    bounds = np.stack([xmin, xmax], 1).astype(np.int32)
    bounds.setflags(write=False)
    kk.setflags(write=False)
    return bounds, kk, ksize


def small_size(nw, nh, f):
    """Size of the resized scene image (trajectories_scene.py:320-326: int(round(width * scale_factor_small)))."""
    return int(round(nw * f)), int(round(nh * f))
