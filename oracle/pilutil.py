"""TEST INFRASTRUCTURE ONLY (oracle) -- `scipy.misc.imresize` as the reference's custom_transforms.py:5,106,135 imports it.

The function was removed from SciPy in 1.3 and does not exist in this image, so the reference cannot be run for this call.
This is a restatement in plain numpy, independent of cc_amd (which goes through PIL itself):
  * SciPy 1.1 `scipy/misc/pilutil.py`: imresize = toimage(arr) -> PIL resize -> array; toimage byte-scales non-uint8 arrays to
    THEIR OWN min..max (`bytescale`: (x - cmin) * 255/(cmax-cmin), clip, +0.5, truncate);
  * Pillow `src/libImaging/Resample.c` (8-bit path): separable horizontal-then-vertical passes, triangle filter of support
    1 * max(scale, 1) around centre (i + 0.5) * scale, coefficients normalised and quantised to 22-bit fixed point
    (PRECISION_BITS = 32 - 8 - 2), accumulation from 1 << 21, shift, clip to 0..255, uint8 intermediate between the passes.
Pinned by tests/test_transforms.py against Pillow itself (the dependency the reference's call lands in) on seeded images, bit for
bit; the fixture tests/golden/transforms.npz is generated with THIS function serving the reference's import."""
import numpy as np

PRECISION_BITS = 32 - 8 - 2


def bytescale(data, high=255, low=0):
    """scipy 1.1 bytescale with cmin/cmax = the array's own extrema."""
    data = np.asarray(data)
    if data.dtype == np.uint8:
        return data
    cmin, cmax = data.min(), data.max()
    cscale = cmax - cmin
    if cscale == 0:
        cscale = 1
    scale = float(high - low) / cscale
    return (((data - cmin) * scale + low).clip(low, high) + 0.5).astype(np.uint8)


def _coefficients(in_size, out_size):
    """Resample.c precompute_coeffs for the bilinear (triangle) filter -> per output index (first input index, int32 weights)."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ss = 1.0 / filterscale
    out = []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        n = xmax - xmin
        w = np.empty(n, dtype=np.float64)
        for x in range(n):
            t = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - t if t < 1.0 else 0.0
        tot = w.sum()
        if tot != 0.0:
            w = w / tot
        kk = np.where(w < 0, np.trunc(-0.5 + w * (1 << PRECISION_BITS)), np.trunc(0.5 + w * (1 << PRECISION_BITS))).astype(np.int64)
        out.append((xmin, kk))
    return out


def _pass(img, out_size, axis):
    """one separable pass over `axis` of a uint8 [H,W,C] image"""
    in_size = img.shape[axis]
    if in_size == out_size:
        return img
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    res = np.empty((out_size,) + src.shape[1:], dtype=np.uint8)
    for xx, (xmin, kk) in enumerate(_coefficients(in_size, out_size)):
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(kk, src[xmin:xmin + len(kk)], axes=(0, 0))
        res[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(res, 0, axis)


def imresize(arr, size, interp='bilinear'):
    """scipy.misc.imresize(arr, (h, w)) for HxW or HxWx3 arrays, bilinear only (what custom_transforms.py uses)."""
    assert interp == 'bilinear'
    img = bytescale(arr)
    two_d = img.ndim == 2
    if two_d:
        img = img[:, :, None]
    h, w = int(size[0]), int(size[1])
    img = _pass(img, w, 1)         # ImagingResample: horizontal pass first
    img = _pass(img, h, 0)
    return img[:, :, 0] if two_d else img


# ----------------------------------------------------------------------------------------------------------------------
# scipy.misc.imrotate (custom_transforms.py:5,84: RandomRotate, composed first in train.py:178-184's pipeline)
#   * SciPy 1.1 pilutil.imrotate = toimage(arr) (byte-scale as above) -> Image.rotate(angle, resample=BILINEAR) -> array;
#   * Pillow Image.rotate (expand=False, centre = (w/2, h/2)): the inverse affine map
#         [ cos(-a)  sin(-a) | cx - (cos(-a)*cx + sin(-a)*cy) ]      cos / sin rounded to 15 decimals
#         [-sin(-a)  cos(-a) | cy - (-sin(-a)*cx + cos(-a)*cy) ]
#   * Pillow src/libImaging/Geometry.c ImagingGenericTransform + affine_transform + bilinear_filter32RGB, all in double:
#     source position of output pixel (x, y) = M . (x + 0.5, y + 0.5); outside [0, w) x [0, h) -> 0 (black); else shift by
#     -0.5, floor, blend the 2x2 neighbourhood (indices clamped to the image, the lower row replaced by the upper one beyond the
#     last row) as v = a + (b - a) * d horizontally then vertically, truncate to uint8.
# Pinned bit for bit against Pillow itself by tests/test_transforms.py::test_oracle_imrotate_is_pillow_bilinear.
def rotate_matrix(w, h, angle):
    """the six coefficients Image.rotate hands to the affine transform (Python floats = C doubles)"""
    import math
    angle = angle % 360.0
    a = -math.radians(angle)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
    cx, cy = w / 2, h / 2
    m[2] = m[0] * -cx + m[1] * -cy + m[2]
    m[5] = m[3] * -cx + m[4] * -cy + m[5]
    m[2] += cx
    m[5] += cy
    return m


def affine_bilinear(img, m):
    """Pillow's affine transform with the bilinear filter on a uint8 [H,W,C] (or [H,W]) image, output size = input size"""
    two_d = img.ndim == 2
    if two_d:
        img = img[:, :, None]
    H, W, C = img.shape
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float64) + 0.5, np.arange(W, dtype=np.float64) + 0.5, indexing="ij")
    xin = m[0] * xs + m[1] * ys + m[2]
    yin = m[3] * xs + m[4] * ys + m[5]
    inside = ~((xin < 0.0) | (xin >= W) | (yin < 0.0) | (yin >= H))
    xin = xin - 0.5
    yin = yin - 0.5
    x = np.floor(xin).astype(np.int64)          # FLOOR(v): floor for negatives, truncation otherwise = floor
    y = np.floor(yin).astype(np.int64)
    dx = (xin - x)[:, :, None]
    dy = (yin - y)[:, :, None]
    x0, x1 = np.clip(x, 0, W - 1), np.clip(x + 1, 0, W - 1)
    yc = np.clip(y, 0, H - 1)
    src = img.astype(np.float64)
    a, b = src[yc, x0], src[yc, x1]
    v1 = a + (b - a) * dx
    y1ok = ((y + 1 >= 0) & (y + 1 < H))[:, :, None]
    y1 = np.clip(y + 1, 0, H - 1)
    a2, b2 = src[y1, x0], src[y1, x1]
    v2 = np.where(y1ok, a2 + (b2 - a2) * dx, v1)
    v = v1 + (v2 - v1) * dy
    out = np.where(inside[:, :, None], v, 0.0).astype(np.uint8)          # (UINT8) v: truncation
    return out[:, :, 0] if two_d else out


def imrotate(arr, angle, interp='bilinear'):
    """scipy.misc.imrotate(arr, angle) for HxW or HxWx3 arrays, bilinear only (what custom_transforms.py uses)."""
    assert interp == 'bilinear'
    img = bytescale(arr)
    a = angle % 360.0
    if a == 0:
        return img.copy()
    assert a not in (90.0, 180.0, 270.0), "quarter turns take Pillow's transpose fast paths (RandomRotate draws from (0, 10))"
    return affine_bilinear(img, rotate_matrix(img.shape[1], img.shape[0], angle))
