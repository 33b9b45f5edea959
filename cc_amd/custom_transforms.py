"""Input pipeline of the reference (custom_transforms.py).

Host side: the transform classes train.py:166-190 composes, same names, arguments, random-number draws (so the same seeds
give the same augmentations) and intrinsics updates.  ``scipy.misc.imresize`` / ``imrotate`` -- removed from SciPy in 1.3
and absent here -- are restated from SciPy 1.1's ``scipy/misc/pilutil.py`` (``toimage`` byte-scales float arrays to their
own min..max before PIL's resize / rotate; bilinear resampling).  Both are pinned by tests/test_transforms.py against Pillow
and against independent numpy restatements of Pillow's 8-bit resampler and of its affine-bilinear transform (oracle/pilutil.py,
which also serves the reference's imports when the fixture is generated).  `RandomRotate` is the FIRST transform of the
pipeline train.py:178-184 composes when the flow network is trained (the all-trainable configuration the metric is quoted on);
train.py:171-176 (--fix-flownet) leaves it out.

Device side: ``DeviceFrames`` fuses ArrayToTensor + Normalize (+ the mirror of RandomHorizontalFlip and the crop of
RandomScaleCrop) for a whole batch of frames into one HIP launch (``cc_frames_to_tensor``) -- HWC uint8 / float32 frames go
to HBM as they are (4x less PCIe traffic for uint8) and come out as the normalised fp32 NCHW tensors the nets consume."""
import random

import numpy as np
import torch

from ._lib import engine, STREAM


# ----------------------------------------------------------------------------- scipy.misc restatements (pinned: tests/test_transforms.py)
def _bytescale(data, cmin=None, cmax=None, high=255, low=0):
    """scipy 1.1 misc/pilutil.py bytescale."""
    if data.dtype == np.uint8:
        return data
    if cmin is None:
        cmin = data.min()
    if cmax is None:
        cmax = data.max()
    cscale = cmax - cmin
    if cscale == 0:
        cscale = 1
    scale = float(high - low) / cscale
    bytedata = (data - cmin) * scale + low
    return (bytedata.clip(low, high) + 0.5).astype(np.uint8)


def _toimage(arr):
    from PIL import Image
    data = np.asarray(arr)
    if data.ndim == 2:
        return Image.fromarray(_bytescale(data), mode='L')
    assert data.ndim == 3 and data.shape[2] in (3, 4), "imresize: expected HxW or HxWx{3,4}"
    return Image.fromarray(np.ascontiguousarray(_bytescale(data)), mode='RGB' if data.shape[2] == 3 else 'RGBA')


def imresize(arr, size, interp='bilinear'):
    """scipy.misc.imresize(arr, (h, w)) -> uint8 array."""
    from PIL import Image
    func = {'nearest': Image.NEAREST, 'lanczos': Image.LANCZOS, 'bilinear': Image.BILINEAR, 'bicubic': Image.BICUBIC}
    im = _toimage(arr)
    if isinstance(size, int):
        size = tuple((np.array(im.size) * (size / 100.0)).astype(int))
    elif isinstance(size, float):
        size = tuple((np.array(im.size) * size).astype(int))
    else:
        size = (size[1], size[0])
    return np.array(im.resize(size, resample=func[interp]))


def imrotate(arr, angle, interp='bilinear'):
    """scipy.misc.imrotate(arr, angle) -> uint8 array (counter-clockwise, same size)."""
    from PIL import Image
    func = {'nearest': Image.NEAREST, 'bilinear': Image.BILINEAR, 'bicubic': Image.BICUBIC}
    return np.array(_toimage(arr).rotate(angle, resample=func[interp]))


def rotate_matrix(w, h, angle):
    """The six affine coefficients PIL.Image.rotate(angle, expand=False) hands to its transform (inverse map, centre (w/2, h/2),
    cos / sin rounded to 15 decimals) -- the arithmetic of Pillow's Image.py, in Python floats = C doubles."""
    import math
    a = -math.radians(angle % 360.0)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
    cx, cy = w / 2, h / 2
    m[2] = m[0] * -cx + m[1] * -cy + m[2]
    m[5] = m[3] * -cx + m[4] * -cy + m[5]
    m[2] += cx
    m[5] += cy
    return m


# ----------------------------------------------------------------------------- host transforms (custom_transforms.py)
class Compose(object):
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, images, intrinsics):
        for t in self.transforms:
            images, intrinsics = t(images, intrinsics)
        return images, intrinsics


class Normalize(object):
    def __init__(self, mean, std):
        self.mean = mean
        self.std = std

    def __call__(self, images, intrinsics):
        for tensor in images:
            for t, m, s in zip(tensor, self.mean, self.std):
                t.sub_(m).div_(s)
        return images, intrinsics


class NormalizeLocally(object):
    def __call__(self, images, intrinsics):
        image_tensor = torch.stack(images)
        assert image_tensor.size(1) == 3
        mean = image_tensor.transpose(0, 1).contiguous().view(3, -1).mean(1)
        std = image_tensor.transpose(0, 1).contiguous().view(3, -1).std(1)
        for tensor in images:
            for t, m, s in zip(tensor, mean, std):
                t.sub_(m).div_(s)
        return images, intrinsics


class ArrayToTensor(object):
    """list of HxWxC arrays -> list of CxHxW float tensors / 255."""

    def __call__(self, images, intrinsics):
        return [torch.from_numpy(np.transpose(im, (2, 0, 1))).float() / 255 for im in images], intrinsics


class RandomHorizontalFlip(object):
    def __call__(self, images, intrinsics):
        assert intrinsics is not None
        if random.random() < 0.5:
            output_intrinsics = np.copy(intrinsics)
            output_images = [np.copy(np.fliplr(im)) for im in images]
            w = output_images[0].shape[1]
            output_intrinsics[0, 2] = w - output_intrinsics[0, 2]
        else:
            output_images, output_intrinsics = images, intrinsics
        return output_images, output_intrinsics


class RandomRotate(object):
    """custom_transforms.py:75-86: with probability 1/2 rotate every frame of the sample by one angle from U(0, 10) degrees."""

    @staticmethod
    def draw():
        """the random decisions of __call__ in its order -> angle in degrees, or None (no rotation)"""
        if np.random.random() > 0.5:
            return None
        return np.random.uniform(0, 10)

    def __call__(self, images, intrinsics):
        rot = self.draw()
        if rot is None:
            return images, intrinsics
        assert intrinsics is not None
        return [imrotate(im, rot) for im in images], intrinsics


class RandomScaleCrop(object):
    def __init__(self, h=0, w=0):
        self.h, self.w = h, w

    def draw(self, in_h, in_w):
        """The random decisions of __call__ in its order: -> (scaled_h, scaled_w, x_scaling, y_scaling, off_y, off_x, out_h, out_w)."""
        x_scaling, y_scaling = np.random.uniform(1, 1.1, 2)
        scaled_h, scaled_w = int(in_h * y_scaling), int(in_w * x_scaling)
        out_h, out_w = (self.h, self.w) if (self.h and self.w) else (in_h, in_w)
        offset_y = np.random.randint(scaled_h - out_h + 1)
        offset_x = np.random.randint(scaled_w - out_w + 1)
        return scaled_h, scaled_w, x_scaling, y_scaling, offset_y, offset_x, out_h, out_w

    def __call__(self, images, intrinsics):
        assert intrinsics is not None
        output_intrinsics = np.copy(intrinsics)
        in_h, in_w, _ = images[0].shape
        scaled_h, scaled_w, xs, ys, offset_y, offset_x, out_h, out_w = self.draw(in_h, in_w)
        output_intrinsics[0] *= xs
        output_intrinsics[1] *= ys
        scaled_images = [imresize(im, (scaled_h, scaled_w)) for im in images]
        cropped_images = [im[offset_y:offset_y + out_h, offset_x:offset_x + out_w] for im in scaled_images]
        output_intrinsics[0, 2] -= offset_x
        output_intrinsics[1, 2] -= offset_y
        return cropped_images, output_intrinsics


class Scale(object):
    def __init__(self, h, w):
        self.h, self.w = h, w

    def __call__(self, images, intrinsics):
        assert intrinsics is not None
        output_intrinsics = np.copy(intrinsics)
        in_h, in_w, _ = images[0].shape
        output_intrinsics[0] *= (self.w / in_w)
        output_intrinsics[1] *= (self.h / in_h)
        return [imresize(im, (self.h, self.w)) for im in images], output_intrinsics


# ----------------------------------------------------------------------------- device side
class DeviceFrames(object):
    """ArrayToTensor + Normalize(mean, std) (+ per-frame mirror / crop window) for a batch of equally sized HWC frames in
    one launch.  frames: list of N arrays [H,W,3] (uint8 or float32 0..255) or one [N,H,W,3] array / tensor;
    flips / offsets: per-frame (bool, (off_y, off_x)) or None.  -> fp32 [N,3,h,w] on `device`."""

    def __init__(self, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5), device="cuda"):
        self.mean, self.std, self.device = [float(m) for m in mean], [float(s) for s in std], torch.device(device)

    def _to_device(self, frames):
        if not torch.is_tensor(frames):
            frames = torch.from_numpy(np.ascontiguousarray(np.stack([np.asarray(f) for f in frames])
                                                           if isinstance(frames, (list, tuple)) else frames))
        assert frames.dim() == 4 and frames.shape[3] == 3 and frames.dtype in (torch.uint8, torch.float32)
        return frames.to(self.device, non_blocking=True).contiguous()

    def __call__(self, frames, out_hw=None, flips=None, offsets=None):
        src = self._to_device(frames)
        N, H, W, _ = src.shape
        h, w = out_hw if out_hw is not None else (H, W)
        geo = np.zeros((N, 3), dtype=np.int32)
        if flips is not None:
            geo[:, 0] = np.asarray(flips, dtype=np.int32)
        if offsets is not None:
            geo[:, 1:] = np.asarray(offsets, dtype=np.int32)
        assert (geo[:, 1] >= 0).all() and (geo[:, 2] >= 0).all() and (geo[:, 1] + h <= H).all() and (geo[:, 2] + w <= W).all()
        geo_d = torch.from_numpy(geo).to(self.device)
        dst = torch.empty(N, 3, h, w, device=self.device, dtype=torch.float32)
        engine().call("cc_frames_to_tensor", src, int(src.dtype == torch.uint8), dst, geo_d, N, H, W, h, w, self.mean[0],
                      self.mean[1], self.mean[2], self.std[0], self.std[1], self.std[2], STREAM)
        return dst

    # ---- RandomRotate on the device
    def rotate(self, frames, angles):
        """scipy.misc.imrotate of every frame (angles[n] in degrees, None = not rotated: byte-scaled only, which is what the
        resize that follows does first) -> uint8 [N,H,W,3] on the device, bit-exact with the host classes."""
        src = self._to_device(frames)
        N, H, W, _ = src.shape
        assert len(angles) == N
        rot = np.zeros((N, 8), dtype=np.float64)
        for n, a in enumerate(angles):
            if a is not None and (a % 360.0) != 0:
                assert (a % 360.0) not in (90.0, 180.0, 270.0), "quarter turns: Pillow transposes instead of resampling"
                rot[n, 0] = 1.0
                rot[n, 1:7] = rotate_matrix(W, H, a)
        E = engine()
        ws = torch.empty(int(E.call("cc_frames_rotate_ws_bytes", N)), dtype=torch.uint8, device=self.device)
        dst = torch.empty(N, H, W, 3, dtype=torch.uint8, device=self.device)
        E.call("cc_frames_rotate", src, int(src.dtype == torch.uint8), dst, torch.from_numpy(rot).to(self.device), ws, N, H, W, STREAM)
        return dst

    # ---- RandomScaleCrop's resize on the device
    def resize_crop(self, frames, scaled_hw, out_hw, flips=None, offsets=None):
        """RandomHorizontalFlip -> imresize to scaled_hw -> crop out_hw at offsets -> ArrayToTensor -> Normalize, all on the
        device and bit-exact with the host classes.  scaled_hw: (sh, sw) or one pair per frame (the frames of one sample share
        a draw, the samples of a batch do not).  -> fp32 [N,3,h,w]."""
        src = self._to_device(frames)
        N, H, W, _ = src.shape
        h, w = out_hw
        sizes = [tuple(scaled_hw)] * N if isinstance(scaled_hw[0], (int, np.integer)) else [tuple(s) for s in scaled_hw]
        assert len(sizes) == N
        tabs = {}
        for sh, sw in sizes:
            tabs.setdefault((H, sh), resample_table(H, sh))
            tabs.setdefault((W, sw), resample_table(W, sw))
        KT = max(t[1].shape[1] for t in tabs.values())
        chunks, offs, pos = [], {}, 0
        for key, (first, wts) in tabs.items():
            pad = np.zeros((wts.shape[0], KT), dtype=np.int32)
            pad[:, :wts.shape[1]] = wts
            offs[key] = pos
            chunks += [first.astype(np.int32), pad.reshape(-1)]
            pos += first.size + pad.size
        geo = np.zeros((N, 8), dtype=np.int32)
        if flips is not None:
            geo[:, 0] = np.asarray(flips, dtype=np.int32)
        if offsets is not None:
            geo[:, 1:3] = np.asarray(offsets, dtype=np.int32)
        for n, (sh, sw) in enumerate(sizes):
            geo[n, 3:7] = (sh, sw, offs[(W, sw)], offs[(H, sh)])
            assert 0 <= geo[n, 1] and geo[n, 1] + h <= sh and 0 <= geo[n, 2] and geo[n, 2] + w <= sw, "crop window outside the scaled frame"
        max_sw = max(sw for _, sw in sizes)
        tmp_w = (max_sw + 3) // 4 * 4
        E = engine()
        ws = torch.empty(int(E.call("cc_frames_resize_ws_bytes", N, H, tmp_w)), dtype=torch.uint8, device=self.device)
        geo_d = torch.from_numpy(geo).to(self.device)
        tab_d = torch.from_numpy(np.concatenate(chunks)).to(self.device)
        dst = torch.empty(N, 3, h, w, device=self.device, dtype=torch.float32)
        E.call("cc_frames_resize_to_tensor", src, int(src.dtype == torch.uint8), dst, geo_d, tab_d, KT, ws, N, H, W, tmp_w, max_sw, h, w,
               self.mean[0], self.mean[1], self.mean[2], self.std[0], self.std[1], self.std[2], STREAM)
        return dst


_TABLES = {}


def resample_table(in_size, out_size):
    """Pillow Resample.c precompute_coeffs + normalize_coeffs_8bpc for the bilinear filter: -> (first input index [out_size],
    int32 weights [out_size, taps]) -- triangle filter of support max(scale, 1) around (i + 0.5) * scale, normalised in double,
    quantised to 22-bit fixed point (PRECISION_BITS = 32 - 8 - 2)."""
    key = (int(in_size), int(out_size))
    if key not in _TABLES:
        scale = float(in_size) / out_size
        fscale = max(scale, 1.0)
        support, ss = 1.0 * fscale, 1.0 / fscale
        center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
        first = np.maximum((center - support + 0.5).astype(np.int64), 0)             # C (int) cast: truncation
        last = np.minimum((center + support + 0.5).astype(np.int64), in_size)
        n = last - first
        idx = np.arange(int(n.max()), dtype=np.float64)[None, :]
        t = np.abs((idx + first[:, None] - center[:, None] + 0.5) * ss)
        wts = np.where((t < 1.0) & (idx < n[:, None]), 1.0 - t, 0.0)
        tot = np.cumsum(wts, axis=1)[:, -1:]                                          # sequential sum, as the C loop
        wts = np.where(tot != 0.0, wts / np.where(tot != 0.0, tot, 1.0), wts)
        q = np.trunc(np.where(wts < 0, -0.5 + wts * (1 << 22), 0.5 + wts * (1 << 22))).astype(np.int32)
        _TABLES[key] = (first.astype(np.int32), q)
    return _TABLES[key]


class DeviceTrainTransform(object):
    """train.py:171-176 `Compose([RandomHorizontalFlip(), RandomScaleCrop(), ArrayToTensor(), Normalize(mean, std)])`, or with
    rotate=True train.py:178-184's pipeline (RandomRotate first: `cc_frames_rotate`), for a whole batch of samples: the random draws and the intrinsics arithmetic happen on the host in the reference's order (Python
    `random` for the flip, then `np.random` uniform(1, 1.1, 2) and the two randint of RandomScaleCrop -- the same seeds give
    the same augmentations), the pixels never leave the device (`DeviceFrames.resize_crop`).
    samples: list of (frames, intrinsics) with frames = list of [H,W,3] arrays, all samples of one size.
    -> (fp32 [B, n_frames, 3, h, w] on the device, list of updated intrinsics)."""

    def __init__(self, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5), device="cuda", h=0, w=0, rotate=False):
        self.frames = DeviceFrames(mean, std, device)
        self.crop = RandomScaleCrop(h, w)
        self.rotate = rotate          # True: train.py:178-184 (RandomRotate first: the pipeline when the flow network is trained)

    def __call__(self, samples):
        flat, flips, offsets, sizes, Ks, angles = [], [], [], [], [], []
        out_hw = None
        for frames, K in samples:
            in_h, in_w, _ = frames[0].shape
            K = np.copy(K)
            rot = RandomRotate.draw() if self.rotate else None                        # RandomRotate, :78-82 (intrinsics unchanged)
            angles += [rot] * len(frames)
            flip = random.random() < 0.5                                              # RandomHorizontalFlip, :62
            if flip:
                K[0, 2] = in_w - K[0, 2]
            sh, sw, xs, ys, oy, ox, oh, ow = self.crop.draw(in_h, in_w)               # RandomScaleCrop, :97-121
            K[0] *= xs
            K[1] *= ys
            K[0, 2] -= ox
            K[1, 2] -= oy
            assert out_hw in (None, (oh, ow)), "DeviceTrainTransform: samples of one batch must share the output size"
            out_hw = (oh, ow)
            for f in frames:
                flat.append(f)
                flips.append(int(flip))
                offsets.append((oy, ox))
                sizes.append((sh, sw))
            Ks.append(K)
        if self.rotate:
            flat = self.frames.rotate(flat, angles)           # uint8 frames on the device; flip / resize / crop follow there
        out = self.frames.resize_crop(flat, sizes, out_hw, flips, offsets)
        nf = len(samples[0][0])
        return out.view(len(samples), nf, 3, out_hw[0], out_hw[1]), Ks
