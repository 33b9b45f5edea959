"""Input pipeline: cc_amd.custom_transforms against fixtures the unmodified reference
custom_transforms.py produced (tests/golden/transforms.npz; its removed `scipy.misc.imresize` import is served by
oracle/pilutil.py -- a numpy restatement pinned to Pillow below and independent of cc_amd -- so the fixture pins the
reference's own arithmetic: RNG draw order, flip, scale, crop, intrinsics updates, /255, normalise), and the fused device kernel
against the host classes, bit for bit."""
import os
import random

import numpy as np
import pytest
import torch

from cc_amd import custom_transforms as CT
from oracle.make_golden import ROT_SEEDS, run_train_transform, run_train_transform_rotate, transform_inputs


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "transforms.npz"))


def test_oracle_imresize_is_pillow_bilinear():
    """oracle/pilutil.py (SciPy 1.1 imresize = bytescale + Pillow 8-bit bilinear resample, restated in numpy) against Pillow
    itself -- up- and down-scaling, both axes, float and uint8 input -- and against the product's imresize."""
    from PIL import Image
    from oracle import pilutil
    r = np.random.RandomState(7)
    for (H, W, h, w) in [(40, 56, 46, 64), (40, 56, 24, 32), (37, 53, 37, 80), (64, 64, 20, 100), (128, 416, 147, 478)]:
        a = (r.rand(H, W, 3) * 300 - 20).astype(np.float32)                  # float input: byte-scaled to its own range first
        ref = np.array(Image.fromarray(pilutil.bytescale(a), mode='RGB').resize((w, h), resample=Image.BILINEAR))
        assert np.array_equal(pilutil.imresize(a, (h, w)), ref), (H, W, h, w)
        assert np.array_equal(CT.imresize(a, (h, w)), ref), (H, W, h, w)
        u8 = r.randint(0, 256, size=(H, W)).astype(np.uint8)
        ref = np.array(Image.fromarray(u8, mode='L').resize((w, h), resample=Image.BILINEAR))
        assert np.array_equal(pilutil.imresize(u8, (h, w)), ref) and np.array_equal(CT.imresize(u8, (h, w)), ref)


def test_oracle_imrotate_is_pillow_bilinear():
    """oracle/pilutil.py imrotate (SciPy 1.1 imrotate = bytescale + Pillow Image.rotate(BILINEAR): the affine map of Image.py and
    Geometry.c's bilinear_filter in double, restated in numpy) against Pillow itself, bit for bit -- small and large angles,
    RGB float (byte-scaled to its own range) and 8-bit grey input -- and against the product's imrotate / rotate_matrix."""
    from PIL import Image
    from oracle import pilutil
    r = np.random.RandomState(3)
    for (H, W) in [(40, 56), (37, 53), (64, 64), (128, 416)]:
        for ang in [0.3, 1.7, 5.0, 9.99, 3.14159, 45.0, 123.4, 359.5]:
            a = (r.rand(H, W, 3) * 300 - 20).astype(np.float32)
            ref = np.array(Image.fromarray(pilutil.bytescale(a), mode='RGB').rotate(ang, resample=Image.BILINEAR))
            assert np.array_equal(pilutil.imrotate(a, ang), ref), (H, W, ang)
            assert np.array_equal(CT.imrotate(a, ang), ref), (H, W, ang)
            assert CT.rotate_matrix(W, H, ang) == pilutil.rotate_matrix(W, H, ang)
            u8 = r.randint(0, 256, size=(H, W)).astype(np.uint8)
            ref = np.array(Image.fromarray(u8, mode='L').rotate(ang, resample=Image.BILINEAR))
            assert np.array_equal(pilutil.imrotate(u8, ang), ref) and np.array_equal(CT.imrotate(u8, ang), ref)


def test_host_rotate_pipeline_matches_reference(gold):
    """train.py:178-184's pipeline (RandomRotate first) through the product's host classes against the fixture the unmodified
    reference classes wrote: same draws (three of the six seeds rotate), pixels and intrinsics bit for bit."""
    rotated = 0
    for seed in ROT_SEEDS:
        imgs, K = run_train_transform_rotate(CT, seed)
        assert np.array_equal(np.asarray(K, dtype=np.float32), gold["rot.seed%d.K" % seed])
        for i, im in enumerate(imgs):
            assert np.array_equal(im.numpy(), gold["rot.seed%d.img%d" % (seed, i)]), (seed, i)
        np.random.seed(seed)
        rotated += CT.RandomRotate.draw() is not None
    assert 0 < rotated < len(ROT_SEEDS)
    frames, _ = transform_inputs()
    assert np.array_equal(CT.imrotate(frames[0], 7.25), gold["rot.direct"])


def test_host_transforms_match_reference(gold):
    for seed in (0, 1, 2, 3):
        imgs, K = run_train_transform(CT, seed)
        assert np.array_equal(np.asarray(K, dtype=np.float32), gold["seed%d.K" % seed])
        for i, im in enumerate(imgs):
            assert np.array_equal(im.numpy(), gold["seed%d.img%d" % (seed, i)]), (seed, i)
    frames, K = transform_inputs()
    sc, Ks = CT.Compose([CT.Scale(h=24, w=32), CT.ArrayToTensor()])([f.copy() for f in frames], np.copy(K))
    assert np.array_equal(np.asarray(Ks, dtype=np.float32), gold["scale.K"]) and np.array_equal(sc[0].numpy(), gold["scale.img0"])


def _device_vs_host(dev):
    frames, K = transform_inputs(seed=5, n=4, H=36, W=52)
    u8 = [f.astype(np.uint8) for f in frames]
    prep = CT.DeviceFrames(mean=(0.5, 0.45, 0.4), std=(0.5, 0.25, 0.2), device=dev)
    flips, offs, hw = [1, 0, 1, 0], [(3, 5), (0, 0), (4, 20), (2, 1)], (32, 32)
    for src in (u8, frames):
        out = prep(src, out_hw=hw, flips=flips, offsets=offs).cpu()
        for n, f in enumerate(src):
            a = np.copy(np.fliplr(f)) if flips[n] else f                              # RandomHorizontalFlip, :66
            a = a[offs[n][0]:offs[n][0] + hw[0], offs[n][1]:offs[n][1] + hw[1]]      # RandomScaleCrop's crop, :116
            t, _ = CT.Compose([CT.ArrayToTensor(), CT.Normalize(prep.mean, prep.std)])([np.ascontiguousarray(a)], None)
            assert torch.equal(out[n], t[0]), n


def _device_train_transform_vs_host(dev, sizes=((40, 56), (36, 52)), rotate=False):
    """DeviceTrainTransform (flip -> Pillow-exact resize -> crop -> /255 -> normalise on the device) against the host classes
    with the same seeds: pixels and intrinsics bit for bit, float32 frames (byte-scaled to their own range, the reference's
    loader) and uint8 frames, several samples with different draws in one call."""
    mean, std = (0.5, 0.45, 0.4), (0.5, 0.25, 0.2)
    for (H, W) in sizes:
        for as_u8 in (False, True):
            samples = []
            for k in range(3):
                frames, K = transform_inputs(seed=50 + k, n=3, H=H, W=W)
                if not as_u8:
                    frames = [f * 0.8 + 10.0 for f in frames]            # min > 0, max < 255: bytescale stretches the contrast
                else:
                    frames = [f.astype(np.uint8) for f in frames]
                samples.append((frames, K))
            random.seed(11)
            np.random.seed(11)
            host = []
            t = CT.Compose(([CT.RandomRotate()] if rotate else []) +
                           [CT.RandomHorizontalFlip(), CT.RandomScaleCrop(), CT.ArrayToTensor(), CT.Normalize(mean=mean, std=std)])
            for frames, K in samples:
                host.append(t([f.copy() for f in frames], np.copy(K)))
            random.seed(11)
            np.random.seed(11)
            out, Ks = CT.DeviceTrainTransform(mean, std, device=dev, rotate=rotate)(samples)
            assert out.shape == (3, 3, 3, H, W)
            flipped = 0
            for b, (imgs, K) in enumerate(host):
                assert np.array_equal(np.asarray(K), np.asarray(Ks[b])), b
                for i, im in enumerate(imgs):
                    assert torch.equal(out[b, i].cpu(), im), (H, W, as_u8, b, i, float((out[b, i].cpu() - im).abs().max()))


def test_resample_table_matches_oracle():
    from oracle import pilutil
    for (a, b) in [(40, 46), (56, 61), (375, 256), (1242, 832), (128, 140), (64, 20), (20, 64)]:
        first, wts = CT.resample_table(a, b)
        for i, (xmin, kk) in enumerate(pilutil._coefficients(a, b)):
            assert first[i] == xmin and np.array_equal(wts[i, :len(kk)], kk) and not wts[i, len(kk):].any(), (a, b, i)


def test_device_train_transform_emulated():
    from hipemu.emu import emulated_engine
    with emulated_engine():
        _device_train_transform_vs_host("cpu")


def _device_rotate_vs_host(dev, sizes=((40, 56), (37, 53))):
    """cc_frames_rotate against scipy.misc.imrotate as the host classes run it (PIL), bit for bit: float32 frames (byte-scaled to
    their own range) and uint8 frames, rotated and unrotated frames in one call."""
    r = np.random.RandomState(9)
    prep = CT.DeviceFrames(device=dev)
    for (H, W) in sizes:
        for as_u8 in (False, True):
            frames = [(r.rand(H, W, 3) * 300 - 20).astype(np.float32) for _ in range(5)]
            if as_u8:
                frames = [np.clip(f, 0, 255).astype(np.uint8) for f in frames]
            angles = [0.37, None, 9.75, 4.2, None]
            out = prep.rotate(frames, angles).cpu().numpy()
            for n, (f, a) in enumerate(zip(frames, angles)):
                ref = CT.imrotate(f, a) if a is not None else CT._bytescale(f)
                assert np.array_equal(out[n], ref), (H, W, as_u8, n, int(np.abs(out[n].astype(int) - ref.astype(int)).max()))


def test_device_rotate_emulated():
    from hipemu.emu import emulated_engine
    with emulated_engine():
        _device_rotate_vs_host("cpu")
        _device_train_transform_vs_host("cpu", sizes=((40, 56),), rotate=True)


@pytest.mark.gpu
def test_device_rotate_gpu():
    _device_rotate_vs_host("cuda", sizes=((40, 56), (128, 416), (256, 832)))
    _device_train_transform_vs_host("cuda", sizes=((40, 56), (128, 416)), rotate=True)


@pytest.mark.gpu
def test_device_train_transform_gpu():
    _device_train_transform_vs_host("cuda", sizes=((40, 56), (128, 416), (256, 832)))


def test_device_frames_emulated():
    from hipemu.emu import emulated_engine
    with emulated_engine():
        _device_vs_host("cpu")


@pytest.mark.gpu
def test_device_frames_gpu():
    _device_vs_host("cuda")
    # one training sample's worth of full-size frames (5 x 256 x 832 uint8) in a single launch
    r = np.random.RandomState(0)
    big = r.randint(0, 256, size=(5, 256, 832, 3)).astype(np.uint8)
    out = CT.DeviceFrames(device="cuda")(big)
    ref = (torch.from_numpy(big).permute(0, 3, 1, 2).float() / 255 - 0.5) / 0.5
    assert torch.equal(out.cpu(), ref)
