"""Seeded synthetic inputs for benchmarks and parity tests.

numpy ``RandomState`` is used instead of the torch generator so the very same
tensors can be re-created on any box / torch build from a seed alone (the
golden fixtures under tests/golden/ store outputs only).

A sample is one target frame + 4 reference frames in the order (t-2, t-1, t+1,
t+2) (datasets/sequence_folders.py:16-21), images in [-1, 1], plus the
KITTI-like pinhole intrinsics the reference's loader would produce for that
resolution (data/kitti_raw_loader.py:97-115 scaling of P_rect).
"""
import numpy as np
import torch


def kitti_intrinsics(batch, height, width, dtype=torch.float32):
    K = np.array([[0.5809 * width, 0.0, 0.4909 * width],
                  [0.0, 1.9242 * height, 0.4609 * height],
                  [0.0, 0.0, 1.0]], dtype=np.float32)
    Kinv = np.linalg.inv(K.astype(np.float64)).astype(np.float32)   # datasets/sequence_folders.py:61
    K = torch.from_numpy(K).to(dtype).unsqueeze(0).repeat(batch, 1, 1).contiguous()
    Kinv = torch.from_numpy(Kinv).to(dtype).unsqueeze(0).repeat(batch, 1, 1).contiguous()
    return K, Kinv


def _smooth(a, passes):
    """cheap separable [1 2 1]/4 low-pass, `passes` times (wrap-around borders)."""
    for _ in range(passes):
        a = 0.25 * np.roll(a, 1, -1) + 0.5 * a + 0.25 * np.roll(a, -1, -1)
        a = 0.25 * np.roll(a, 1, -2) + 0.5 * a + 0.25 * np.roll(a, -1, -2)
    return a


def frames(batch, height, width, seed=1, n_frames=5, smooth=0):
    """-> list of n_frames float32 tensors [B,3,H,W] in [-1,1].

    smooth=0: plain uniform noise (parity tests); smooth>0: low-pass textured
    frames where consecutive frames are small translations of one texture plus
    noise, so warps look like real sequences (timing runs)."""
    rs = np.random.RandomState(seed)
    if smooth <= 0:
        return [torch.from_numpy((rs.rand(batch, 3, height, width) * 2 - 1).astype(np.float32))
                for _ in range(n_frames)]
    base = _smooth(rs.rand(batch, 3, height, width + 64).astype(np.float32), smooth)
    base = (base - base.min()) / (base.max() - base.min()) * 2 - 1
    out = []
    for i in range(n_frames):
        sh = 4 * i
        f = base[..., sh:sh + width] + 0.02 * (rs.rand(batch, 3, height, width).astype(np.float32) - 0.5)
        out.append(torch.from_numpy(np.clip(f, -1, 1).astype(np.float32)).contiguous())
    return out


def sample(batch, height=256, width=832, seed=1, smooth=0, device="cpu"):
    """-> (tgt, [ref t-2, t-1, t+1, t+2], K, Kinv) on `device`."""
    fr = frames(batch, height, width, seed, 5, smooth)
    tgt = fr[2]
    refs = [fr[0], fr[1], fr[3], fr[4]]
    K, Kinv = kitti_intrinsics(batch, height, width)
    mv = lambda t: t.to(device)
    return mv(tgt), [mv(r) for r in refs], mv(K), mv(Kinv)


def kernel_inputs(batch, height, width, seed=2):
    """Inputs for kernel-level tests that bypass the nets:
    disp ~ U(0.01,10.01) smooth -> depth = 1/disp; pose ~ N(0, 0.01^2) [B,4,6];
    flows ~ N(0, 2^2) px smooth [B,2,H,W]; masks = sigmoid(N(0,1)) [B,4,H,W]."""
    rs = np.random.RandomState(seed)
    disp = _smooth(rs.rand(batch, 1, height, width).astype(np.float32), 2) * 10 + 0.01
    depth = (1.0 / disp).astype(np.float32)
    pose = (rs.randn(batch, 4, 6) * 0.01).astype(np.float32)
    flow_f = (_smooth(rs.randn(batch, 2, height, width).astype(np.float32), 2) * 8).astype(np.float32)
    flow_b = (_smooth(rs.randn(batch, 2, height, width).astype(np.float32), 2) * 8).astype(np.float32)
    mask = (1 / (1 + np.exp(-rs.randn(batch, 4, height, width)))).astype(np.float32)
    t = torch.from_numpy
    return dict(depth=t(depth), pose=t(pose), flow_fwd=t(flow_f), flow_bwd=t(flow_b), mask=t(mask))


def seeded_state_dict(module, seed=0):
    """Fill every floating-point entry of module.state_dict() from a numpy stream,
    in key order: xavier-uniform-like weights, small non-zero biases, BN stats
    left at their defaults.  Portable across torch builds (unlike manual_seed +
    init_weights) and identical for any module with the same keys/shapes."""
    rs = np.random.RandomState(seed)
    sd = module.state_dict()
    out = {}
    for k, v in sd.items():
        if not v.is_floating_point() or "running_" in k:
            out[k] = v.clone()
            continue
        if v.dim() == 4:
            fan_in = v.shape[1] * v.shape[2] * v.shape[3]
            fan_out = v.shape[0] * v.shape[2] * v.shape[3]
            a = float(np.sqrt(6.0 / (fan_in + fan_out)))
            arr = (rs.rand(*v.shape) * 2 - 1) * a
        elif k.endswith("bias") and v.dim() == 1:
            arr = (rs.rand(*v.shape) - 0.5) * 0.1
        else:   # BN weight
            arr = 1.0 + (rs.rand(*v.shape) - 0.5) * 0.2
        out[k] = torch.from_numpy(arr.astype(np.float32))
    return out
