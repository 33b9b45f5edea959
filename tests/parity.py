"""Shared parity cases: the HIP engine (cc_amd, through the C ABI) against the oracle on identical seeded
inputs.  `dev` is "cuda" for the real library (-m gpu) or "cpu" for the x86 emulation build of the same
kernel sources (tests/hipemu).  Tolerances are the ones SURVEY.md section 8c / BASELINE.json state."""
import numpy as np
import pytest
import torch

from cc_amd import synthetic as syn, inverse_warp as IW, loss_functions as LF, ssim as SS
from oracle import geometry as G, losses as L
from oracle.make_golden import pyramid_inputs


def to(dev, *ts):
    out = [t.to(dev) if torch.is_tensor(t) else [x.to(dev) for x in t] for t in ts]
    return out if len(out) > 1 else out[0]


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def frac_bad(a, b, atol=1e-5, rtol=1e-4):
    """fraction of elements outside |a-b| <= atol + rtol*|b| (robust to isolated bilinear-tap flips: the host CPU
    the oracle runs on may round a sampling coordinate to the other side of an integer than the GPU does)."""
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float(((a - b).abs() > atol + rtol * b.abs()).double().mean())


def leaf(t, dev):
    return t.detach().clone().to(dev).requires_grad_(True)


def check_warps(dev, B=2, H=24, W=40, smooth=0, atol=1e-5):
    """a8/a9/a10 + Back2Future.warp vs the oracle run on THIS host's CPU: forward 1e-5 abs, gradients 2e-5 rel
    (bit-exactness is checked against the reference's own outputs in check_warps_bit_exact_vs_golden)."""
    tgt, refs, K, Kinv = syn.sample(B, H, W, seed=1, smooth=smooth)
    ki = syn.kernel_inputs(B, H, W)
    pose = ki["pose"] * 3
    Kd, Kinvd = to(dev, K, Kinv)
    for ac in (False, True):
        for pad in ("zeros", "border"):
            d, p, im = leaf(ki["depth"][:, 0], dev), leaf(pose[:, 0], dev), leaf(refs[0], dev)
            d0, p0, im0 = leaf(ki["depth"][:, 0], "cpu"), leaf(pose[:, 0], "cpu"), leaf(refs[0], "cpu")
            # the kernel boundary is P = K.[R|t]: feed the oracle's P so coordinates can be compared bit-for-bit
            o = IW.inverse_warp(im, d, p, Kd, Kinvd, padding_mode=pad, align_corners=ac)
            r = G.inverse_warp(im0, d0, p0, K, Kinv, padding_mode=pad, align_corners=ac)
            go = torch.randn(r.shape, generator=torch.Generator().manual_seed(5))
            o.backward(go.to(dev))
            r.backward(go)
            assert frac_bad(o, r, atol) < 2e-3, (frac_bad(o, r, atol), float((o.detach().cpu() - r.detach()).abs().max()))
            assert frac_bad(d.grad, d0.grad, max(1e-6, atol * 0.1) * float(d0.grad.abs().max())) < 2e-3
            # d(out)/d(img) is the bilinear weight itself: a coordinate difference of atol px shows up as ~4*atol*|go|
            tol_im = 1e-5 if atol <= 1e-5 else 8 * atol
            assert frac_bad(im.grad, im0.grad, tol_im) < 2e-3, frac_bad(im.grad, im0.grad, tol_im)
            assert rel(p.grad, p0.grad) < 1e-2, rel(p.grad, p0.grad)
        d, p = leaf(ki["depth"][:, 0], dev), leaf(pose[:, 1], dev)
        d0, p0 = leaf(ki["depth"][:, 0], "cpu"), leaf(pose[:, 1], "cpu")
        f, f0 = IW.pose2flow(d, p, Kd, Kinvd), G.pose2flow(d0, p0, K, Kinv)
        assert float((f.detach().cpu() - f0.detach()).abs().max()) < 1e-3
        gf = torch.randn(f0.shape, generator=torch.Generator().manual_seed(6))
        g1 = torch.autograd.grad(f, [d, p], gf.to(dev))
        g0 = torch.autograd.grad(f0, [d0, p0], gf)
        assert rel(g1[0], g0[0]) < 1e-4 and rel(g1[1], g0[1]) < 1e-4
        fl, im = leaf(ki["flow_fwd"], dev), leaf(refs[1], dev)
        fl0, im0 = leaf(ki["flow_fwd"], "cpu"), leaf(refs[1], "cpu")
        o, r = IW.flow_warp(im, fl, align_corners=ac), G.flow_warp(im0, fl0, align_corners=ac)
        assert frac_bad(o, r, atol) < 2e-3
        go = torch.randn(r.shape, generator=torch.Generator().manual_seed(7))
        g1 = torch.autograd.grad(o, [im, fl], go.to(dev))
        g0 = torch.autograd.grad(r, [im0, fl0], go)
        assert frac_bad(g1[0], g0[0], (1e-5 if atol <= 1e-5 else 8 * atol)) < 2e-3 and frac_bad(g1[1], g0[1], (1e-4 if atol <= 1e-5 else 8 * atol)) < 2e-3
        for FC in (8, 18):          # 18 >= 16 channels: the channel-group parallel backward (4 groups of 5,5,5,3)
            ft = torch.randn(B, FC, H, W, generator=torch.Generator().manual_seed(8))
            fe, fe0 = leaf(ft, dev), leaf(ft, "cpu")
            o, r = IW.feature_warp(fe, fl, align_corners=ac), G.feature_warp(fe0, fl0, align_corners=ac)
            assert frac_bad(o, r, atol) < 2e-3
            go = torch.randn(r.shape, generator=torch.Generator().manual_seed(9))
            g1 = torch.autograd.grad(o, [fe, fl], go.to(dev))
            g0 = torch.autograd.grad(r, [fe0, fl0], go)
            assert frac_bad(g1[0], g0[0], (1e-5 if atol <= 1e-5 else 8 * atol)) < 2e-3 and frac_bad(g1[1], g0[1], (1e-4 if atol <= 1e-5 else 8 * atol)) < 2e-3


def check_ssim(dev, cases=((2, 40, 70, 0), (1, 64, 96, 3), (2, 8, 26, 0), (1, 33, 31, 1))):
    """a11: per-pixel 2e-4 abs (fp32 cancellation noise of the reference itself), mean 1e-5 rel."""
    for (B, H, W, smooth) in cases:
        fr = syn.frames(B, H, W, seed=3, n_frames=2, smooth=smooth)
        x, y = leaf(fr[0], dev), leaf(fr[1], dev)
        x0, y0 = leaf(fr[0], "cpu"), leaf(fr[1], "cpu")
        o, r = SS.ssim(x, y), L.ssim(x0, y0)
        assert float((o.detach().cpu() - r.detach()).abs().max()) < 2e-4
        assert abs(float(o.mean()) - float(r.mean())) < 1e-5 * abs(float(r.mean())) + 1e-7
        go = torch.randn(r.shape, generator=torch.Generator().manual_seed(4))
        g1 = torch.autograd.grad(o, [x, y], go.to(dev))
        g0 = torch.autograd.grad(r, [x0, y0], go)
        assert rel(g1[0], g0[0]) < 5e-5 and rel(g1[1], g0[1]) < 5e-5
    xx = torch.rand(1, 3, 20, 30)
    assert float((SS.ssim(xx.to(dev), xx.to(dev)).cpu() - 1).abs().max()) < 1e-5     # ssim(x, x) == 1
    # the reference's ssim() takes any channel count and window size (ssim.py:42-76; the module form defaults to 11 taps): other
    # channel counts run on the same 13-tap kernels, other windows on the generic depth-wise path
    gen = torch.Generator().manual_seed(6)
    for C, win in ((1, 13), (4, 13), (2, 11), (3, 7)):
        a0, b0 = torch.rand(2, C, 24, 36, generator=gen), torch.rand(2, C, 24, 36, generator=gen)
        a1, b1 = leaf(a0, dev), leaf(b0, dev)
        a0, b0 = leaf(a0, "cpu"), leaf(b0, "cpu")
        o = SS.ssim(a1, b1, win) if win != 11 else SS.SSIM()(a1, b1)
        r = SS._ssim_generic(a0, b0, win)                                         # the reference's formula on the CPU
        assert o.shape == r.shape and float((o.detach().cpu() - r.detach()).abs().max()) < 2e-4, (C, win)
        go = torch.randn(r.shape, generator=gen)
        g1 = torch.autograd.grad(o, [a1, b1], go.to(dev))
        g0 = torch.autograd.grad(r, [a0, b0], go)
        assert rel(g1[0], g0[0]) < 5e-5 and rel(g1[1], g0[1]) < 5e-5, (C, win)


def _pyr(dev, B, H, W, smooth=0):
    tgt, refs, K, Kinv = syn.sample(B, H, W, seed=1, smooth=smooth)
    pyr = pyramid_inputs(B, H, W)
    pose = syn.kernel_inputs(B, 8, 8, seed=2)["pose"] * 3.0

    def mk(d):
        return dict(depth=[leaf(p["depth"], d) for p in pyr], mask=[leaf(p["mask"], d) for p in pyr],
                    ff=[leaf(p["flow_fwd"], d) for p in pyr], fb=[leaf(p["flow_bwd"], d) for p in pyr],
                    pose=leaf(pose, d), tgt=tgt.to(d), refs=[r.to(d) for r in refs], K=K.to(d), Kinv=Kinv.to(d))
    return mk(dev), mk("cpu")


def _cmp(name, l1, l0, w1, w0, ltol=1e-4, gtol=1e-4):
    assert abs(float(l1) - float(l0)) <= ltol * abs(float(l0)), (name, float(l1), float(l0))
    g1 = torch.autograd.grad(l1, w1, allow_unused=True)
    g0 = torch.autograd.grad(l0, w0, allow_unused=True)
    for a, b in zip(g1, g0):
        assert (a is None) == (b is None), name
        if b is not None:
            # max-norm where the two sides are arithmetic-identical, flip-tolerant otherwise (see frac_bad)
            # pose gradients are pixel sums over every (possibly flipped) tap: small tensors, 1e-2 of their max
            ok = rel(a, b) < gtol or (b.numel() < 1000 and _note("cmp:small_grad_rel", rel(a, b)) < 5e-3) or (
                frac_bad(a, b, gtol * float(b.abs().max()), 1e-3) < 2e-3 and
                float((a.detach().cpu() - b.detach()).norm() / (b.detach().norm() + 1e-30)) < 5e-2)
            assert ok, (name, rel(a, b))


def check_losses(dev, B=2, H=64, W=96, smooth=0):
    """a12-a17 with gradients (north-star bar: losses within 1e-4 rel of the reference CPU path)."""
    for ac in (False, True):
        a, o = _pyr(dev, B, H, W, smooth)
        _cmp("photometric_reconstruction_loss",
             LF.photometric_reconstruction_loss(a["tgt"], a["refs"], a["K"], a["Kinv"], a["depth"], a["mask"], a["pose"],
                                                wssim=0.997, qch=0.5, align_corners=ac),
             L.photometric_reconstruction_loss(o["tgt"], o["refs"], o["K"], o["Kinv"], o["depth"], o["mask"], o["pose"],
                                               wssim=0.997, qch=0.5, align_corners=ac),
             a["depth"] + a["mask"] + [a["pose"]], o["depth"] + o["mask"] + [o["pose"]])
        a, o = _pyr(dev, B, H, W, smooth)
        _cmp("photometric_reconstruction_loss(no mask, lambda_oob, qch=0.4)",
             LF.photometric_reconstruction_loss(a["tgt"], a["refs"], a["K"], a["Kinv"], a["depth"], [None] * 6, a["pose"],
                                                wssim=0.5, qch=0.4, lambda_oob=0.3, align_corners=ac),
             L.photometric_reconstruction_loss(o["tgt"], o["refs"], o["K"], o["Kinv"], o["depth"], [None] * 6, o["pose"],
                                               wssim=0.5, qch=0.4, lambda_oob=0.3, align_corners=ac),
             a["depth"] + [a["pose"]], o["depth"] + [o["pose"]])
        a, o = _pyr(dev, B, H, W, smooth)
        _cmp("photometric_flow_loss",
             LF.photometric_flow_loss(a["tgt"], a["refs"][1:3], [a["fb"], a["ff"]], [1 - m[:, 1:3] for m in a["mask"]],
                                      wssim=0.997, align_corners=ac),
             L.photometric_flow_loss(o["tgt"], o["refs"][1:3], [o["fb"], o["ff"]], [1 - m[:, 1:3] for m in o["mask"]],
                                     wssim=0.997, align_corners=ac),
             a["ff"] + a["fb"] + a["mask"], o["ff"] + o["fb"] + o["mask"])
    a, o = _pyr(dev, B, H, W, smooth)
    _cmp("explainability_loss", LF.explainability_loss(a["mask"]), L.explainability_loss(o["mask"]), a["mask"], o["mask"])
    for nm in ("depth", "ff", "mask"):
        a, o = _pyr(dev, B, H, W, smooth)
        # 5 scales: the 6th (2x3) makes the reference's smooth_loss NaN (mean over an empty tensor)
        _cmp("smooth_loss " + nm, LF.smooth_loss(a[nm][:5]), L.smooth_loss(o[nm][:5]), a[nm][:5], o[nm][:5])
        assert torch.isnan(LF.smooth_loss(a[nm])) and torch.isnan(L.smooth_loss(o[nm]))
        _cmp("edge_aware_smoothness_loss " + nm, LF.edge_aware_smoothness_loss(a["tgt"], a[nm]),
             L.edge_aware_smoothness_loss(o["tgt"], o[nm]), a[nm], o[nm])
    # the four smoothness terms of the training step as ONE job table (engine extension, cc_amd.trainer.cc_forward)
    a, o = _pyr(dev, B, H, W, smooth)
    _cmp("edge_aware_smoothness_sum", LF.edge_aware_smoothness_sum(a["tgt"], [a["depth"], a["ff"], a["fb"], a["mask"]]),
         sum(L.edge_aware_smoothness_loss(o["tgt"], o[nm]) for nm in ("depth", "ff", "fb", "mask")),
         a["depth"] + a["ff"] + a["fb"] + a["mask"], o["depth"] + o["ff"] + o["fb"] + o["mask"])
    a, o = _pyr(dev, B, H, W, smooth)
    with torch.no_grad():
        cf1 = [IW.pose2flow(d[:, 0], a["pose"][:, 2], a["K"], a["Kinv"]) for d in a["depth"]]
        cb1 = [IW.pose2flow(d[:, 0], a["pose"][:, 1], a["K"], a["Kinv"]) for d in a["depth"]]
        cf0 = [G.pose2flow(d[:, 0], o["pose"][:, 2], o["K"], o["Kinv"]) for d in o["depth"]]
        cb0 = [G.pose2flow(d[:, 0], o["pose"][:, 1], o["K"], o["Kinv"]) for d in o["depth"]]
        t1 = LF.consensus_exp_masks(cf1, cb1, a["ff"], a["fb"], a["tgt"], a["refs"][2], a["refs"][1], wssim=0.997, wrig=1.0)
        t0 = L.consensus_exp_masks(cf0, cb0, o["ff"], o["fb"], o["tgt"], o["refs"][2], o["refs"][1], wssim=0.997, wrig=1.0)
        for x, y in zip(t1, t0):
            assert float((x.cpu() != y).float().mean()) <= 2e-3      # discontinuous output: flip rate
        assert 0.2 < float(t0[0].mean()) < 0.95                        # both classes present at the fine level
        occ1 = LF.depth_occlusion_masks(a["depth"][0], a["pose"], a["K"], a["Kinv"])
        occ0 = L.depth_occlusion_masks(o["depth"][0], o["pose"], o["K"], o["Kinv"])
        assert float((occ1.cpu() != occ0).float().mean()) <= 1e-5
        big_b, big_f = o["fb"][0] * 3 + 4, o["ff"][0] * 3 + 4         # force occlusions (signed-sum rule, Q5)
        ob1, of1 = LF.occlusion_masks(big_b.to(dev), big_f.to(dev))
        ob0, of0 = L.occlusion_masks(big_b, big_f)
        assert torch.equal(ob1.cpu(), ob0) and torch.equal(of1.cpu(), of0) and 0.05 < float(ob0.mean()) < 0.95
        rf = [(x - y).abs() for x, y in zip(cf0, o["ff"])]
        rb = [(x - y).abs() for x, y in zip(cb0, o["fb"])]
    _cmp("consensus_depth_flow_mask",
         LF.consensus_depth_flow_mask(a["mask"], to(dev, rb), to(dev, rf), to(dev, t0), to(dev, t0), THRESH=0.5, wbce=0.5),
         L.consensus_depth_flow_mask(o["mask"], rb, rf, t0, t0, THRESH=0.5, wbce=0.5), a["mask"], o["mask"])


def check_occluded_photo_loss(dev, B=2, H=32, W=48):
    """photometric_flow_loss with flows large enough that the occlusion factor (1 - occ) is exercised."""
    tgt, refs, K, Kinv = syn.sample(B, H, W, seed=4)
    ki = syn.kernel_inputs(B, H, W, seed=9)
    fb, ff = ki["flow_bwd"] * 0.5 + 0.3, ki["flow_fwd"] * 0.5 + 0.3
    assert 0.05 < float(L.occlusion_masks(fb, ff)[0].mean()) < 0.95
    m = ki["mask"][:, :2].contiguous()
    a = dict(fb=leaf(fb, dev), ff=leaf(ff, dev), m=leaf(m, dev))
    o = dict(fb=leaf(fb, "cpu"), ff=leaf(ff, "cpu"), m=leaf(m, "cpu"))
    _cmp("photometric_flow_loss (occluded, single scale)",
         LF.photometric_flow_loss(tgt.to(dev), to(dev, refs[1:3]), [a["fb"], a["ff"]], a["m"], wssim=0.5),
         L.photometric_flow_loss(tgt, refs[1:3], [o["fb"], o["ff"]], o["m"], wssim=0.5),
         [a["fb"], a["ff"], a["m"]], [o["fb"], o["ff"], o["m"]])


def check_pyramid(dev):
    from cc_amd._lib import engine, STREAM
    x = syn.frames(2, 64, 96, seed=11, n_frames=1)[0]
    xd = x.to(dev)
    for (h, w) in ((32, 48), (16, 24), (2, 3), (20, 31)):
        ref = torch.nn.functional.adaptive_avg_pool2d(x, (h, w))
        out = torch.empty(2, 3, h, w, device=dev)
        engine().call("cc_adaptive_avg_pool", xd, out, 6, 64, 96, h, w, STREAM)
        assert float((out.cpu() - ref).abs().max()) < 1e-6
    packed = torch.empty(6 * (32 * 48 + 16 * 24 + 8 * 12 + 4 * 6 + 2 * 3), device=dev)
    engine().call("cc_pyramid_build", xd, packed, 6, 6, 64, 96, STREAM)      # one launch: all levels of each 32x32 tile
    off = 0
    for l in (1, 2, 3, 4, 5):
        h, w = 64 >> l, 96 >> l
        lv = packed[off:off + 6 * h * w].view(2, 3, h, w).cpu()
        assert float((lv - torch.nn.functional.adaptive_avg_pool2d(x, (h, w))).abs().max()) < 1e-6
        off += 6 * h * w
    # the frames of a step in ONE launch (loss_functions.pyramid_cache.prefetch -> cc_pyramid_build_multi): bit-identical to the
    # per-image launch, and what get() then returns
    from cc_amd import loss_functions as LF
    frames = [f.to(dev) for f in syn.frames(2, 64, 96, seed=12, n_frames=3)]
    LF.pyramid_cache.clear()
    LF.pyramid_cache.prefetch(frames)
    for f in frames:
        single = torch.empty_like(packed)
        engine().call("cc_pyramid_build", f, single, 6, 6, 64, 96, STREAM)
        off = 0
        for l in (1, 2, 3, 4, 5):
            h, w = 64 >> l, 96 >> l
            assert (h, w) in LF.pyramid_cache.items[id(f)][2], "prefetch did not fill the cache"
            assert torch.equal(LF.pyramid_cache.get(f, h, w), single[off:off + 6 * h * w].view(2, 3, h, w))
            off += 6 * h * w
    LF.pyramid_cache.clear()


def check_warps_bit_exact_vs_golden(dev, golden_dir):
    """Index arithmetic + blend of the warp kernels reproduce the REFERENCE's outputs bit-for-bit (fixtures written by
    the unmodified reference in the build container, oracle/make_golden.py) when given the reference's P = K.[R|t]
    (the kernel boundary).  Host-CPU independent: nothing is recomputed on this box's CPU."""
    import os
    from oracle.make_golden import FB, FH, FW
    tgt, refs, K, Kinv = syn.sample(FB, FH, FW, seed=1)
    pyr = pyramid_inputs(FB, FH, FW)
    d0 = pyr[0]["depth"][:, 0].to(dev)
    out = {}
    for tag, ac in (("acF", 0), ("acT", 1)):
        g = dict(np.load(os.path.join(golden_dir, "functions_%s.npz" % tag)))
        P = torch.from_numpy(g["P"]).reshape(-1, 12).to(dev)
        w = IW._InverseWarpFn.apply(refs[0].to(dev), d0, P, Kinv.to(dev), 0, ac).cpu().numpy()
        f = IW._Pose2FlowFn.apply(d0, P, Kinv.to(dev), 0).cpu().numpy()
        fw = IW.flow_warp(refs[1].to(dev), pyr[0]["flow_fwd"].to(dev), align_corners=bool(ac)).cpu().numpy()
        feat = syn.frames(FB, 16, 24, seed=7, n_frames=1)[0]
        feat = torch.cat([feat, feat.flip(1), feat * 0.5], 1)[:, :8].contiguous()
        flo = syn.kernel_inputs(FB, 16, 24, seed=8)["flow_fwd"]
        ft = IW.feature_warp(feat.to(dev), flo.to(dev), align_corners=bool(ac)).cpu().numpy()
        # the stand-alone halves (inverse_warp.py:31-79): pixel2cam -> cam2pixel with the reference's P reproduce its grid
        P34 = torch.from_numpy(g["P"]).to(dev)
        grid = IW.cam2pixel(IW.pixel2cam(d0, Kinv.to(dev)), P34[:, :, :3].contiguous(), P34[:, :, 3:].contiguous(), "zeros")
        # tap-flip rate: how many pixels sample a different 2x2 neighbourhood when P comes from THIS device's sin/cos
        # (cc_pose_proj_fwd) instead of the reference's CPU matrices -- the number behind the flip-tolerant gradient bars
        pose = syn.kernel_inputs(FB, 8, 8, seed=2)["pose"] * 3.0
        P_dev = IW.projection_matrix(pose[:, 0].to(dev), K.to(dev)).reshape(-1, 3, 4)
        grid_dev = IW.cam2pixel(IW.pixel2cam(d0, Kinv.to(dev)), P_dev[:, :, :3].contiguous(), P_dev[:, :, 3:].contiguous(), "zeros")

        def taps(gr):
            x = torch.floor((gr[..., 0] + 1) * (FW / 2.0) - 0.5) if not ac else torch.floor((gr[..., 0] + 1) / 2 * (FW - 1))
            y = torch.floor((gr[..., 1] + 1) * (FH / 2.0) - 0.5) if not ac else torch.floor((gr[..., 1] + 1) / 2 * (FH - 1))
            return x, y
        (xa, ya), (xb, yb) = taps(grid), taps(grid_dev)
        out[tag] = dict(inverse_warp=float(np.mean(w == g["inverse_warp"])), pose2flow=float(np.mean(f == g["pose2flow"])),
                        flow_warp=float(np.mean(fw == g["flow_warp"])), feature_warp=float(np.mean(ft == g["feature_warp"])),
                        grid=float(np.mean(grid.cpu().numpy() == g["grid_zeros"])),
                        tap_flip_rate_device_P=float(((xa != xb) | (ya != yb)).float().mean()),
                        P_max_rel_diff=float((P_dev - P34).abs().max() / P34.abs().max()),
                        maxabs=float(np.abs(w - g["inverse_warp"]).max()))
    return out


MEASURED = {}      # worst value seen per bar (printed by the tests: the bars below are held at ~2x what is measured)


def _note(key, v):
    MEASURED[key] = max(MEASURED.get(key, 0.0), float(v))
    return v


def _grad_ok(gr, ref, tight):
    """tight: max-norm 2e-4 of the gradient's max.  Otherwise (white-noise frames, where the last-ulp difference
    between this device's P = K.[R|t] and the reference's can move a bilinear tap across a pixel boundary):
    all but 5e-4 of the elements within 1e-4 of the max, and 5e-3 in L2 (the elements that ARE off are pinned to tap
    boundaries by _flip_pinned)."""
    mx = float(ref.abs().max())
    l2 = float((gr.detach().cpu() - ref).norm() / (ref.norm() + 1e-30))
    if tight:
        # low-pass frames: a flipped tap changes the interpolation SLOPE at that pixel only slightly -> 5e-3 in L2
        return rel(gr, ref) < 2e-4 or (ref.numel() < 1000 and rel(gr, ref) < 2e-3) or \
            (frac_bad(gr, ref, 1e-4 * mx, 1e-3) < 2e-3 and l2 < 5e-3)
    if ref.numel() < 1000:          # pose gradient: a pixel sum over every (possibly flipped) tap (measured worst 2.5e-3)
        return _note("noise:pose_grad_rel", rel(gr, ref)) < 5e-3
    if rel(gr, ref) < 2e-4:
        return True
    # measured (emulator and MI355X): 1.6e-4 of the elements off the tight bar, 1.5e-3 in L2 -> bars at ~3x that
    return _note("noise:frac_bad", frac_bad(gr, ref, 1e-4 * mx, 1e-3)) < 5e-4 and _note("noise:l2", l2) < 5e-3


def _tap_boundary_distance(a, i, ac):
    """Distance (pixels) of every rigid sampling coordinate of pyramid level i to the nearest bilinear tap boundary (an
    integer coordinate), minimum over the four reference frames -- from the ORACLE's P on the host (loss_functions.py:91-92
    intrinsics scaling, inverse_warp.py:66-72 normalisation, ATen's grid_sampler unnormalise).  -> [B,H,W]"""
    d = a["depth"][i].detach().cpu()[:, 0]
    B, h, w = d.shape
    down = a["tgt"].shape[2] / h
    K, Kinv = a["K"].detach().cpu(), a["Kinv"].detach().cpu()
    K_s = torch.cat((K[:, 0:2] / down, K[:, 2:]), dim=1)
    Kinv_s = torch.cat((Kinv[:, :, 0:2] * down, Kinv[:, :, 2:]), dim=2)
    best = torch.full((B, h, w), 1.0)
    for r in range(a["pose"].shape[1]):
        grid = G.warp_grid(d, a["pose"].detach().cpu()[:, r], K_s, Kinv_s, padding_mode=None)
        for c, n in ((0, w), (1, h)):
            pix = (grid[..., c] + 1) / 2 * (n - 1) if ac else ((grid[..., c] + 1) * n - 1) / 2
            best = torch.minimum(best, (pix - pix.round()).abs())
    return best


def _flip_pinned(name, a, wrt_keys, grads, g, ac):
    """White-noise frames, rigid warp: the product computes P = K.[R|t] with its own sin/cos (cc_pose_proj_fwd), one ulp away
    from the reference's; where a sampling coordinate sits within ~1e-4 px of an integer the bilinear taps -- and, on noise,
    that pixel's gradient -- change.  This pins that mechanism: EVERY depth-gradient element outside 1e-4 of the maximum lies
    at a pixel whose coordinate is within 2e-3 px of a tap boundary, there are only a handful of them, and everything else
    meets the tight bar.  -> number of such elements."""
    nflip = 0
    for k, gr in zip(wrt_keys, grads):
        if gr is None or not k.startswith("depth"):
            continue
        ref = torch.from_numpy(g[name + ".grad." + k])
        bad = (gr.detach().cpu() - ref).abs() > 1e-4 * float(ref.abs().max())
        if bool(bad.any()):
            near = _tap_boundary_distance(a, int(k[5:]), ac) < 2e-3
            assert bool((near.unsqueeze(1) | ~bad).all()), (name, k, "a gradient element off the tight bar is NOT at a tap boundary")
            nflip += int(bad.sum())
            assert int(bad.sum()) <= max(4, ref.numel() // 2000), (name, k, int(bad.sum()))
    return nflip


def check_losses_vs_golden(dev, golden_dir):
    """a12-a17 (+ gradients) against the fixtures the UNMODIFIED reference wrote (oracle/make_golden.py): the
    host-CPU-independent parity gate.  Losses within 1e-4 rel (north star) on white-noise AND low-pass frames;
    gradients: tight on the low-pass frames, flip-tolerant on white noise (see _grad_ok)."""
    import os
    from oracle.make_golden import FB, FH, FW
    res = {}
    for tag, ac in (("acF", False), ("acT", True)):
        for smooth, suffix in ((0, ""), (3, "_smooth")):
            g = dict(np.load(os.path.join(golden_dir, "functions_%s%s.npz" % (tag, suffix))))

            def chk(name, loss, wrt, tight):
                if name not in g:
                    return
                assert abs(float(loss) - float(g[name])) <= 1e-4 * abs(float(g[name])), (tag, suffix, name, float(loss), float(g[name]))
                grads = torch.autograd.grad(loss, list(wrt.values()), allow_unused=True)
                worst = 0.0
                for (k, _), gr in zip(wrt.items(), grads):
                    key = name + ".grad." + k
                    if gr is None:
                        assert key not in g
                        continue
                    ref = torch.from_numpy(g[key])
                    worst = max(worst, rel(gr, ref))
                    assert _grad_ok(gr, ref, tight), (tag, suffix, name, k, rel(gr, ref))
                res[tag + suffix + ":" + name] = worst
                if not tight and "pose" in wrt:
                    res[tag + suffix + ":" + name + ":elements_at_tap_boundaries"] = _flip_pinned(name, a, list(wrt.keys()), grads, g, ac)

            tight_warp = smooth > 0
            a, _ = _pyr(dev, FB, FH, FW, smooth)
            wrt = {("depth%d" % i): d for i, d in enumerate(a["depth"])}
            wrt.update({("mask%d" % i): m for i, m in enumerate(a["mask"])})
            wrt["pose"] = a["pose"]
            chk("photometric_reconstruction_loss",
                LF.photometric_reconstruction_loss(a["tgt"], a["refs"], a["K"], a["Kinv"], a["depth"], a["mask"], a["pose"],
                                                   wssim=0.997, qch=0.5, align_corners=ac), wrt, tight_warp)
            a, _ = _pyr(dev, FB, FH, FW, smooth)
            wrt = {("depth%d" % i): d for i, d in enumerate(a["depth"])}
            wrt["pose"] = a["pose"]
            chk("photometric_reconstruction_loss_nomask",
                LF.photometric_reconstruction_loss(a["tgt"], a["refs"], a["K"], a["Kinv"], a["depth"], [None] * 6, a["pose"],
                                                   wssim=0.5, qch=0.5, lambda_oob=0.3, align_corners=ac), wrt, tight_warp)
            a, _ = _pyr(dev, FB, FH, FW, smooth)
            wrt = {("flow_fwd%d" % i): f for i, f in enumerate(a["ff"])}
            wrt.update({("flow_bwd%d" % i): f for i, f in enumerate(a["fb"])})
            wrt.update({("mask%d" % i): m for i, m in enumerate(a["mask"])})
            # flow warps take no P: coordinates are bit-identical to the reference's -> tight on noise too
            chk("photometric_flow_loss",
                LF.photometric_flow_loss(a["tgt"], a["refs"][1:3], [a["fb"], a["ff"]], [1 - m[:, 1:3] for m in a["mask"]],
                                         wssim=0.997, qch=0.5, align_corners=ac), wrt, True)
            if smooth:
                continue
            mk = {("mask%d" % i): m for i, m in enumerate(a["mask"])}
            chk("explainability_loss", LF.explainability_loss(a["mask"]), mk, True)
            for nm, key in (("depth", "depth"), ("flow_fwd", "ff"), ("mask", "mask")):
                chk("edge_aware_smoothness_loss." + nm, LF.edge_aware_smoothness_loss(a["tgt"], a[key]),
                    {("%s%d" % (nm, i)): t for i, t in enumerate(a[key])}, True)
            with torch.no_grad():
                p = a["pose"].detach()
                cf = [IW.pose2flow(d[:, 0], p[:, 2], a["K"], a["Kinv"]) for d in a["depth"]]
                cb = [IW.pose2flow(d[:, 0], p[:, 1], a["K"], a["Kinv"]) for d in a["depth"]]
                tg = LF.consensus_exp_masks(cf, cb, a["ff"], a["fb"], a["tgt"], a["refs"][2], a["refs"][1], wssim=0.997,
                                            wrig=1.0, align_corners=ac)
                for i, t in enumerate(tg):
                    assert float(np.mean(t.cpu().numpy().astype(np.uint8) != g["consensus_exp_masks.%d" % i])) <= 5e-3
                occ = LF.depth_occlusion_masks(a["depth"][0], p, a["K"], a["Kinv"])
                assert float(np.mean(occ.cpu().numpy().astype(np.uint8) != g["depth_occlusion_masks.0"])) <= 1e-4
                rf = [torch.from_numpy(g["pose2flow_fullK.%d" % i]).to(dev) - y for i, y in enumerate(a["ff"])]
                rf = [x.abs() for x in rf]
                rb = [(x - y).abs() for x, y in zip(cb, a["fb"])]
                tgt_ref = [torch.from_numpy(g["consensus_exp_masks.%d" % i].astype(np.float32)).to(dev) for i in range(6)]
            # THRESH-ed census masks are discontinuous in the rigid flow: compare the loss only loosely on this term's
            # inputs that came through this device's P (rb); exact reference values are used for rf
            l5 = LF.consensus_depth_flow_mask(a["mask"], rb, rf, tgt_ref, tgt_ref, THRESH=0.5, wbce=0.5)
            assert abs(float(l5) - float(g["consensus_depth_flow_mask"])) <= 1e-3 * abs(float(g["consensus_depth_flow_mask"]))
    return res


# ----------------------------------------------------------------------------------------------------------------------
# convolutions (the networks' hot kernels): engine vs torch fp32 on the CPU (the reference's nn.Conv2d / ConvTranspose2d)
CONV_CASES = [  # B, Cin, H, W, Cout, k, stride, pad, act, bias, residual
    (2, 3, 16, 20, 32, 7, 2, 3, "relu", True, False),
    (2, 16, 9, 13, 40, 3, 1, 1, "lrelu", True, False),
    (1, 20, 8, 10, 2, 3, 1, 1, None, True, False),
    (2, 33, 7, 9, 70, 3, 2, 1, "relu", False, False),
    (2, 24, 6, 7, 24, 3, 1, 1, "relu", False, True),
    (1, 130, 5, 6, 150, 1, 1, 0, None, True, False),
    (2, 8, 10, 12, 1, 3, 1, 1, "sigmoid", True, False),
    (2, 5, 12, 9, 16, 5, 2, 2, "relu", True, False),
    (2, 12, 6, 8, 12, 1, 2, 0, None, False, False),
    (2, 10, 11, 70, 20, 3, 1, 1, "relu", True, False),
    (1, 6, 21, 75, 8, 3, 2, 1, "relu", True, False),
    (2, 64, 5, 40, 64, 3, 1, 1, "relu", False, True),
    (1, 40, 6, 32, 96, 3, 1, 1, "lrelu", True, False),
    # few-channel 3x3 layers on larger maps (prediction heads: <= 4 outputs; their data-gradients: <= 4 reduction channels; a
    # 3-channel input layer)
    (1, 20, 32, 72, 4, 3, 1, 1, "sigmoid", True, False),
    (2, 9, 33, 70, 1, 3, 1, 1, "sigmoid", True, False),
    (1, 16, 40, 64, 2, 3, 1, 1, None, True, False),
    (1, 3, 32, 72, 16, 3, 1, 1, "relu", True, False),
    (2, 2, 35, 66, 5, 3, 1, 1, "lrelu", False, False),
    (2, 7, 9, 13, 3, 3, 1, 1, "sigmoid", True, False),          # odd width: one pixel per work-item in conv_heads.hip
    # maps of <= 4 lattice rows: the pixel tile stacks the rows of 2 / 4 / 8 consecutive images (conv.hip CP::ipt) -- every channel
    # tile (16 / 32 / 64 / 128), a batch that does not fill the last tile, stride 2 onto such a map (and its data-gradient's parity
    # classes), rows that are / are not 16-byte aligned, split-K (deep) and fused epilogues
    (4, 40, 2, 7, 70, 3, 1, 1, "relu", True, True),
    (4, 72, 4, 13, 130, 3, 1, 1, "lrelu", True, False),
    (5, 24, 3, 5, 20, 3, 1, 1, None, True, False),
    (8, 16, 1, 4, 10, 3, 1, 1, "sigmoid", True, False),
    (3, 33, 8, 14, 40, 3, 2, 1, "relu", True, False),
    (4, 20, 4, 8, 24, 3, 2, 1, "relu", False, False),
    (4, 64, 2, 8, 64, 3, 1, 1, "relu", True, False),
    (3, 12, 4, 12, 12, 1, 1, 0, None, True, False),
    (4, 260, 2, 7, 48, 3, 1, 1, "relu", True, False),
]
# 3x3 / stride 1 / pad 1 layers on the Winograd F(2x2, 3x3) kernel (wino.hip; forward and data-gradient): odd map sizes (tiles that
# hang over the border), channel counts that are not multiples of 4 / 8 / 64, residual + every activation, tile blocks that span
# images, split-K over channel chunks (deep layer, few tiles)
CONV_CASES_WINO = [      # (the kernel takes maps whose width is a multiple of 4 with >= 16 tile columns, or exactly 8)
    (2, 64, 5, 40, 64, 3, 1, 1, "relu", False, True),
    (1, 40, 6, 32, 96, 3, 1, 1, "lrelu", True, False),
    (2, 33, 7, 36, 70, 3, 1, 1, "relu", True, False),
    (3, 26, 9, 44, 40, 3, 1, 1, None, False, False),
    (2, 129, 8, 32, 65, 3, 1, 1, "sigmoid", True, False),
    (1, 196, 6, 16, 128, 3, 1, 1, "lrelu", True, False),
    (2, 48, 16, 52, 48, 3, 1, 1, "relu", True, True),
    (4, 256, 4, 16, 130, 3, 1, 1, "relu", True, False),
    (1, 24, 5, 104, 40, 3, 1, 1, "lrelu", True, False),
]
# ... at sizes whose launches are large enough to keep the whole reduction in one workgroup (fused bias / residual / activation
# epilogue in the Winograd kernel): GPU only (the reference is torch on the CPU: a few GFLOP each)
# (smooth epilogues only: among millions of outputs a few pre-activations lie within the summation-order noise of zero, and a
# ReLU / LeakyReLU derivative that flips there moves single gradient elements by O(1) -- in ANY two implementations)
CONV_CASES_WINO_LARGE = [
    (4, 32, 64, 208, 128, 3, 1, 1, None, True, False),
    (4, 64, 64, 208, 64, 3, 1, 1, None, False, True),
    (2, 40, 63, 204, 130, 3, 1, 1, "sigmoid", True, False),
    (4, 128, 32, 104, 128, 3, 1, 1, None, True, True),
    (4, 129, 33, 100, 65, 3, 1, 1, "sigmoid", True, False),
]
# few channels x many pixels (wgrad_thin.hip; the pixel threshold is lowered for the small test maps)
# 3x3 / stride-1 layers on maps whose width is NOT a multiple of 4 (8x26, 4x13 in the step): their weight gradients run on the Winograd
# kernel over zero-padded copies of dY and the input (conv.hip k_pad_rows); smooth epilogues (see CONV_CASES_WINO_LARGE)
CONV_CASES_WINO_PADW = [          # product thresholds (>= 96 channels on both sides, >= 64 tiles)
    (4, 96, 8, 26, 128, 3, 1, 1, None, True, False),
    (4, 100, 4, 13, 96, 3, 1, 1, "sigmoid", True, False),
    (3, 128, 5, 10, 97, 3, 1, 1, None, False, False),
]
# ... and their forward / data-gradient arithmetic on the Winograd kernel over a zero-padded copy of the INPUT (conv.hip plan_conv
# ConvPlan::wpad: always split-K, the epilogue kernel writes the real output): product thresholds (>= 256 channels on both sides)
CONV_CASES_WINO_PADIN = [
    (4, 256, 8, 26, 256, 3, 1, 1, None, True, True),
    (2, 256, 9, 30, 260, 3, 1, 1, "sigmoid", True, False),
]
CONV_CASES_WINO_PADIN_SMALL = [   # emulator sizes (thresholds lowered through the tools switches); 13 -> 16 columns (8 tile columns)
    (2, 20, 6, 26, 24, 3, 1, 1, None, True, False),
    (1, 16, 5, 13, 24, 3, 1, 1, "sigmoid", True, True),
    (3, 24, 4, 30, 40, 3, 1, 1, "relu", True, False),
    (2, 9, 3, 26, 33, 3, 1, 1, "lrelu", False, False),
]
CONV_CASES_WINO_PADW_SMALL = [    # emulator sizes (thresholds lowered through the tools switches)
    (2, 20, 4, 10, 24, 3, 1, 1, None, True, False),
    (2, 33, 5, 13, 40, 3, 1, 1, "sigmoid", True, True),
    (1, 16, 3, 7, 24, 3, 1, 1, None, True, False),
]
# prediction heads (<= 4 output channels from <= 64 inputs) on the VALU kernel of conv_heads.hip (k_conv_thinm; the pixel threshold is
# lowered for the small test maps): 1 / 2 / 4 / 8 channel shares per workgroup, a channel count that does not divide by the shares
CONV_CASES_HEADS = [
    (1, 16, 40, 64, 2, 3, 1, 1, None, True, False),
    (2, 8, 10, 12, 1, 3, 1, 1, "sigmoid", True, False),
    (1, 16, 7, 36, 4, 3, 1, 1, "sigmoid", True, False),
    (2, 40, 8, 16, 4, 3, 1, 1, "relu", True, False),
    (2, 64, 6, 8, 3, 3, 1, 1, "lrelu", False, False),
    (1, 5, 9, 20, 1, 3, 1, 1, None, False, False),
    (1, 37, 5, 24, 2, 3, 1, 1, "sigmoid", True, False),
]
CONV_CASES_THIN = [
    (2, 16, 9, 16, 16, 3, 1, 1, "relu", True, False),
    (2, 17, 6, 20, 16, 3, 1, 1, "lrelu", True, False),
    (1, 16, 7, 36, 4, 3, 1, 1, "sigmoid", True, False),
    (2, 3, 12, 24, 16, 3, 2, 1, "relu", True, False),
    (2, 15, 16, 40, 16, 7, 2, 3, "relu", True, False),
    (1, 32, 6, 16, 32, 7, 1, 3, "relu", True, False),
    (2, 17, 5, 12, 16, 1, 1, 0, None, False, False),
    (2, 16, 10, 24, 32, 5, 2, 2, "relu", True, False),
    (2, 16, 5, 8, 1, 3, 1, 1, "sigmoid", True, False),
    (2, 20, 6, 44, 30, 3, 2, 1, None, True, False),
    (2, 16, 32, 128, 16, 3, 1, 1, "relu", True, False),          # 64 unit ranges: the XCD-swizzled assignment
    (2, 24, 10, 24, 40, 1, 2, 0, None, False, False),            # 1x1 stride-2 shortcut convs of the ResNet blocks
]
CONVT_CASES = [  # B, Cin, H, W, Cout, k, stride, pad, output_padding, act
    (2, 16, 5, 7, 24, 3, 2, 1, 1, "relu"),
    (2, 20, 4, 6, 12, 4, 2, 1, 0, "relu"),
    (1, 40, 3, 5, 40, 3, 2, 1, 1, None),
    (2, 48, 6, 8, 16, 4, 2, 1, 0, "relu"),
    (2, 32, 5, 8, 16, 3, 2, 1, 1, None),
    (4, 24, 2, 7, 16, 3, 2, 1, 1, "relu"),       # stacked tiny maps (see CONV_CASES)
    (3, 40, 1, 4, 40, 4, 2, 1, 0, None),
]


def _torch_act(y, act, a, b):
    import torch.nn.functional as F
    if act == "relu":
        return F.relu(y)
    if act == "lrelu":
        return F.leaky_relu(y, b if b != 0 else 0.2)
    if act == "sigmoid":
        return a * torch.sigmoid(y) + b
    return y


def check_convs(dev, cases=CONV_CASES, tcases=CONVT_CASES, tol=2e-5, seed=0, prepack=False):
    """conv2d / conv_transpose2d forward (+ fused bias / residual / activation epilogue) and all gradients, max-abs
    error relative to the largest reference magnitude <= tol (fp32 MFMA accumulation order differs from the CPU's).
    prepack=True runs every case a second time through the trainer's per-step weight images (cc_repack_table) -- the
    path on which the parity classes of stride-2 data-gradients / transposed convs share one launch."""
    import torch.nn.functional as F
    from cc_amd import ops
    g = torch.Generator().manual_seed(seed)
    passes = 2 if prepack else 1
    ops.packs.reset()

    def rn(*s):
        return torch.randn(*s, generator=g)
    for (B, Cin, H, W, Cout, k, st, pad, act, hb, hr) in cases:
        x0, w0 = rn(B, Cin, H, W), rn(Cout, Cin, k, k) * 0.2
        b0 = rn(Cout) if hb else None
        OH, OW = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
        r0 = rn(B, Cout, OH, OW) if hr else None
        aa, ab = (10.0, 0.01) if act == "sigmoid" else (1.0, 0.0)
        ins_d = [leaf(t, dev) if t is not None else None for t in (x0, w0, b0, r0)]
        ins_c = [leaf(t, "cpu") if t is not None else None for t in (x0, w0, b0, r0)]
        r = F.conv2d(ins_c[0], ins_c[1], ins_c[2], st, pad)
        if hr:
            r = r + ins_c[3]
        r = _torch_act(r, act, aa, ab)
        go = rn(*r.shape)
        g0 = torch.autograd.grad(r, [t for t in ins_c if t is not None], go)
        for ps in range(passes):
            if ps == 1:
                ops.packs.prepack_all()
            y = ops.conv2d(ins_d[0], ins_d[1], ins_d[2], st, pad, act, ins_d[3], aa, ab)
            g1 = torch.autograd.grad(y, [t for t in ins_d if t is not None], go.to(dev))
            errs = [rel(y, r)] + [rel(a, b) for a, b in zip(g1, g0)]
            assert max(errs) < tol, ((B, Cin, H, W, Cout, k, st, pad, act), ps, errs)
            ops.packs.invalidate()
    for (B, Cin, H, W, Cout, k, st, pad, op, act) in tcases:
        x0, w0, b0 = rn(B, Cin, H, W), rn(Cin, Cout, k, k) * 0.2, rn(Cout)
        ins_d = [leaf(t, dev) for t in (x0, w0, b0)]
        ins_c = [leaf(t, "cpu") for t in (x0, w0, b0)]
        r = _torch_act(F.conv_transpose2d(ins_c[0], ins_c[1], ins_c[2], st, pad, op), act, 1.0, 0.0)
        go = rn(*r.shape)
        g0 = torch.autograd.grad(r, ins_c, go)
        for ps in range(passes):
            if ps == 1:
                ops.packs.prepack_all()
            y = ops.conv_transpose2d(ins_d[0], ins_d[1], ins_d[2], st, pad, op, act)
            g1 = torch.autograd.grad(y, ins_d, go.to(dev))
            errs = [rel(y, r)] + [rel(a, b) for a, b in zip(g1, g0)]
            assert max(errs) < tol, ((B, Cin, H, W, Cout, k, st, pad, op, act), ps, errs)
            ops.packs.invalidate()
    ops.packs.reset()


# Winograd launches that hold the WHOLE reduction (fused epilogue, no split-K) with the non-smooth epilogues the step uses: ReLU /
# LeakyReLU (+ residual) in the forward kernel (EPI_LIN) and `(sum) * act'(y1)` in the data-gradient of the consumer (EPI_GRAD: the
# producer's activation backward deferred into it).  (B, Cin, H, W, C1, C2, act, residual)
CONV_CASES_WINO_ACT = [
    (4, 64, 64, 208, 64, 64, "relu", True),        # 64 x 64 blocks (208 of them) + the 32 x 32 instances of the second layer's gradient
    (4, 128, 32, 104, 128, 96, "lrelu", False),    # 32 x 32 blocks, four waves
    (4, 256, 16, 52, 256, 128, "relu", True),      # 32 x 32 blocks, eight waves (the reduction halved inside the workgroup)
    (2, 32, 64, 208, 32, 48, "lrelu", True),       # 32 output channels: one 32-row block
]
CONV_CASES_WINO_ACT_SMALL = [                      # emulator sizes
    (2, 48, 8, 32, 40, 48, "relu", True),
    (1, 64, 6, 32, 64, 32, "lrelu", False),
]


def check_convs_act_pinned(dev, cases=CONV_CASES_WINO_ACT, tol=2e-5):
    """x -> y1 = act(conv1(x) + b1 [+ res]) -> y2 = conv2(y1) + b2, every gradient, with the activation derivative PINNED: among
    millions of pre-activations a few lie within the summation-order noise of zero, and a ReLU / LeakyReLU derivative that falls the
    other way there moves single gradient elements by O(1) in ANY two implementations.  The reference therefore takes its derivative
    mask from the device's own y1 (> 0); the test asserts that the masks differ only where the reference's pre-activation is within
    1e-5 of zero (relative to the largest), counts those elements, and holds every gradient to the tight bar.  Both forms of the
    activation backward are run: as a pass of its own (the layer's output has a non-conv consumer) and deferred into the consumer's
    data-gradient epilogue (defer / pre_act: the form the tape uses inside the networks)."""
    import torch.nn.functional as F
    from cc_amd import ops
    g = torch.Generator().manual_seed(33)

    def rn(*s):
        return torch.randn(*s, generator=g)
    report = []
    for (B, Cin, H, W, C1, C2, act, hr) in cases:
        slope = 0.0 if act == "relu" else 0.2
        x0, w1, b1 = rn(B, Cin, H, W), rn(C1, Cin, 3, 3) * (1.5 / (3 * Cin ** 0.5)), rn(C1) * 0.3
        w2, b2 = rn(C2, C1, 3, 3) * (1.5 / (3 * C1 ** 0.5)), rn(C2) * 0.3
        r0 = rn(B, C1, H, W) if hr else None
        go = rn(B, C2, H, W)
        with torch.no_grad():
            pre = F.conv2d(x0, w1, b1, 1, 1)
            if hr:
                pre = pre + r0
        for deferred in ((False, True) if not hr else (False,)):      # (the deferred form has no residual operand: _Conv2dFn)
            ops.packs.reset()
            td = [leaf(t, dev) if t is not None else None for t in (x0, w1, b1, r0, w2, b2)]
            if deferred:
                y1 = ops.conv2d(td[0], td[1], td[2], 1, 1, act, None, 1.0, slope, defer=True)
                y2 = ops.conv2d(y1, td[4], td[5], 1, 1, None, pre_act=act, pre_slope=slope)
            else:
                y1 = ops.conv2d(td[0], td[1], td[2], 1, 1, act, td[3], 1.0, slope)
                y2 = ops.conv2d(y1, td[4], td[5], 1, 1, None)
            g1 = torch.autograd.grad(y2, [t for t in td if t is not None], go.to(dev))
            mask = (y1.detach().cpu() > 0)
            flips = mask != (pre > 0)
            nflip = int(flips.sum())
            if nflip:
                assert float(pre[flips].abs().max()) <= 1e-5 * float(pre.abs().max()), ("flip away from zero", float(pre[flips].abs().max()))
            assert nflip <= 1e-4 * pre.numel(), (nflip, pre.numel())
            tc = [leaf(t, "cpu") if t is not None else None for t in (x0, w1, b1, r0, w2, b2)]
            p1 = F.conv2d(tc[0], tc[1], tc[2], 1, 1)
            if hr:
                p1 = p1 + tc[3]
            y1c = torch.where(mask, p1, slope * p1)                  # the activation with the device's derivative mask
            y2c = F.conv2d(y1c, tc[4], tc[5], 1, 1)
            g0 = torch.autograd.grad(y2c, [t for t in tc if t is not None], go)
            errs = [rel(y1, y1c), rel(y2, y2c)] + [rel(a, b) for a, b in zip(g1, g0)]
            assert max(errs) < tol, ((B, Cin, H, W, C1, C2, act, hr), deferred, errs)
            report.append(((B, Cin, H, W, C1, C2, act, hr, deferred), nflip, max(errs)))
    ops.packs.reset()
    return report


def check_conv_groups(dev, tol=2e-5, prepack=True, cases=((2, 6, 9, 14, 20, 12, 1), (1, 3, 12, 20, 16, 24, 2), (2, 40, 5, 8, 136, 16, 1), (1, 32, 4, 16, 72, 8, 1))):
    """Grouped convolution chains (cc_conv2d_*_group: the parallel decoder / feature branches of Back2Future as one launch
    per pass) vs per-branch torch convs: conv(stride s, LeakyReLU, deferred activation backward) -> conv(LeakyReLU) -> conv,
    G = 3 branches with their own inputs and parameters, one branch's output left out of the loss (no gradient: it must drop
    out of the backward launches), all input / weight / bias gradients."""
    import torch.nn.functional as F
    from cc_amd import ops
    g = torch.Generator().manual_seed(21)
    G = 3

    def rn(*s):
        return torch.randn(*s, generator=g)
    for (B, Cin, H, W, C1, C2, st) in cases:
        x0 = [rn(B, Cin, H, W) for _ in range(G)]
        w1 = [rn(C1, Cin, 3, 3) * 0.2 for _ in range(G)]
        b1 = [rn(C1) * 0.3 for _ in range(G)]
        w2 = [rn(C2, C1, 3, 3) * 0.1 for _ in range(G)]
        b2 = [rn(C2) * 0.3 for _ in range(G)]
        w3 = [rn(2, C2, 3, 3) * 0.1 for _ in range(G)]
        b3 = [rn(2) for _ in range(G)]
        flat = x0 + w1 + b1 + w2 + b2 + w3 + b3
        td = [leaf(t, dev) for t in flat]
        tc = [leaf(t, "cpu") for t in flat]

        def split(ts):
            return [ts[i * G:(i + 1) * G] for i in range(7)]
        # reference
        xc, w1c, b1c, w2c, b2c, w3c, b3c = split(tc)
        outs_c = []
        for k in range(G):
            y = F.leaky_relu(F.conv2d(xc[k], w1c[k], b1c[k], st, 1), 0.2)
            y = F.leaky_relu(F.conv2d(y, w2c[k], b2c[k], 1, 1), 0.2)
            outs_c.append(F.conv2d(y, w3c[k], b3c[k], 1, 1))
        go = [rn(*o.shape) for o in outs_c]
        used = [0, 2]                                     # branch 1 gets no gradient
        loss_c = sum((outs_c[k] * go[k]).sum() for k in used)
        wrt_c = [t for i, t in enumerate(tc) if (i % G) in used]
        g0 = torch.autograd.grad(loss_c, wrt_c)
        for ps in range(2 if prepack else 1):
            ops.packs.reset()
            if ps == 1:
                ops.packs.prepack_all()                   # opens registration; second prepack below builds the images
            xd, w1d, b1d, w2d, b2d, w3d, b3d = split(td)

            def run():
                y = ops.conv2d_group(xd, w1d, b1d, st, 1, "lrelu", defer=True)
                y = ops.conv2d_group(y, w2d, b2d, 1, 1, "lrelu", pre_act="lrelu", defer=True)
                return ops.conv2d_group(y, w3d, b3d, 1, 1, None, pre_act="lrelu")
            if ps == 1:
                with torch.no_grad():
                    run()                                 # registers the layers
                ops.packs.prepack_all()
            outs_d = run()
            for k in range(G):
                assert rel(outs_d[k], outs_c[k]) < tol, ("group fwd", (B, Cin, H, W, C1, C2, st), ps, k, rel(outs_d[k], outs_c[k]))
            loss_d = sum((outs_d[k] * go[k].to(dev)).sum() for k in used)
            wrt_d = [t for i, t in enumerate(td) if (i % G) in used]
            g1 = torch.autograd.grad(loss_d, wrt_d)
            errs = [rel(a, b) for a, b in zip(g1, g0)]
            assert max(errs) < tol, ("group grads", (B, Cin, H, W, C1, C2, st), ps, errs)
            ops.packs.invalidate()
    ops.packs.reset()


def check_conv_list(dev, tol=2e-5):
    """cc_conv2d_list: four problems of different shapes and kinds in one call -- a residual + ReLU forward (130 channels),
    a plain forward on another map size, a stride-2 data-gradient with the fused (sum + add) * relu'(mul) epilogue and a
    stride-1 data-gradient accumulating in place under LeakyReLU' -- against torch's convolutions."""
    import torch.nn.functional as F
    from cc_amd import ops
    from tools import launchlist as LL
    g = torch.Generator().manual_seed(5)

    def rn(*s):
        return torch.randn(*s, generator=g)
    ops.packs.reset()
    xa, wa, ba, ra = rn(2, 24, 5, 8), rn(130, 24, 3, 3) * 0.2, rn(130), rn(2, 130, 5, 8)
    xb, wb = rn(1, 70, 9, 14), rn(128, 70, 3, 3) * 0.1
    wc, gyc, mulc, addc = rn(136, 40, 3, 3) * 0.1, rn(2, 136, 5, 6), torch.relu(rn(2, 40, 10, 12)), rn(2, 40, 10, 12)
    wd, gyd, muld, gxd0 = rn(72, 33, 3, 3) * 0.1, rn(2, 72, 6, 7), rn(2, 33, 6, 7), rn(2, 33, 6, 7)
    # a 160-channel wide buffer whose channels 20..59 receive problem C's result (batch-strided output)
    wide = torch.zeros(2, 160, 10, 12)
    d = [t.clone().to(dev) for t in (xa, wa, ba, ra, xb, wb, wc, gyc, mulc, addc, wd, gyd, muld, gxd0, wide)]
    xa_, wa_, ba_, ra_, xb_, wb_, wc_, gyc_, mulc_, addc_, wd_, gyd_, muld_, gxd_, wide_ = d
    ya_ = torch.empty(2, 130, 5, 8, device=dev)
    yb_ = torch.empty(1, 128, 9, 14, device=dev)
    gxc_ = wide_[:, 20:60]
    recs = [LL.conv_record(xa_, wa_, ba_, ra_, ya_, 1, 1, 1),
            LL.conv_record(xb_, wb_, None, None, yb_, 1, 1, 0),
            LL.tconv_record(gyc_, wc_, None, gxc_, 2, 1, 40 * 9, 9, 3, 3, act=1, mul=mulc_, add=addc_),
            LL.tconv_record(gyd_, wd_, None, gxd_, 1, 1, 33 * 9, 9, 3, 3, act=2, mul=muld_, add=gxd_)]
    ll = LL.LaunchList(recs, dev)
    ops.packs.prepack_all()
    ll.run()
    ops.packs.invalidate()
    ref = [F.relu(F.conv2d(xa, wa, ba, 1, 1) + ra), F.conv2d(xb, wb, None, 1, 1),
           (F.conv_transpose2d(gyc, wc, None, 2, 1, 1) + addc) * (mulc > 0).float()]
    rd = F.conv_transpose2d(gyd, wd, None, 1, 1) + gxd0
    ref.append(torch.where(muld > 0, rd, 0.2 * rd))
    errs = [rel(a, b) for a, b in zip((ya_, yb_, gxc_, gxd_), ref)]
    assert max(errs) < tol, errs
    assert float(wide_[:, :20].abs().max()) == 0 and float(wide_[:, 60:].abs().max()) == 0      # the slice's neighbours untouched
    ops.packs.reset()


def check_feature_warp_deterministic(dev, cases=((2, 32, 16, 24), (1, 96, 12, 20), (2, 128, 8, 13))):
    """cc_feature_warp_bwd_det (config.deterministic: the feature-gradient scatter as 64-bit fixed-point integer atomics) against
    the float-atomic kernel: same gradient to 1e-6 of its magnitude, accumulation onto an existing gradient, and two runs of
    the deterministic form bit for bit (flows of several pixels, many targets landing on the same source pixel: border
    clamping)."""
    from cc_amd._lib import engine, STREAM
    E = engine()
    g = torch.Generator().manual_seed(17)
    for (B, C, H, W) in cases:
        feat = torch.randn(B, C, H, W, generator=g).to(dev)
        flow = (torch.randn(B, 2, H, W, generator=g) * 6.0).to(dev)          # many samples beyond the border (clamped)
        gout = (torch.randn(B, C, H, W, generator=g) * 1e-4).to(dev)
        base = torch.randn(B, C, H, W, generator=g).to(dev) * 1e-4
        ref = torch.zeros_like(feat)
        gflow0 = torch.empty_like(flow)
        E.call("cc_feature_warp_bwd", gout, feat, flow, gflow0, ref, B, C, H, W, 0, 0.625, STREAM)
        outs = []
        for rep in range(2):
            ws = torch.empty(int(E.call("cc_feature_warp_bwd_det_ws_bytes", B, C, H, W)), dtype=torch.uint8, device=dev)
            got, gflow1 = torch.empty_like(feat), torch.empty_like(flow)
            E.call("cc_feature_warp_bwd_det", gout, feat, flow, gflow1, got, ws, B, C, H, W, 0, 0.625, 0, STREAM)
            outs.append(got)
            assert torch.equal(gflow0, gflow1)
        assert torch.equal(outs[0], outs[1])
        assert rel(outs[0], ref) < 1e-6, ((B, C, H, W), rel(outs[0], ref))
        acc = base.clone()
        ws = torch.empty(int(E.call("cc_feature_warp_bwd_det_ws_bytes", B, C, H, W)), dtype=torch.uint8, device=dev)
        E.call("cc_feature_warp_bwd_det", gout, feat, flow, None, acc, ws, B, C, H, W, 0, 0.625, 1, STREAM)
        assert rel(acc, base + ref) < 1e-6


def check_pixel2cam_cam2pixel_grads(dev, B=2, H=20, W=28):
    """The stand-alone pixel2cam / cam2pixel (inverse_warp.py:31-79) under autograd: outputs and ALL gradients (depth, K^-1,
    camera points, rotation, translation; 'zeros' rewrite and the no-rotation / no-translation switches) of the HIP kernels
    against the oracle's restatement of the reference formulas."""
    g = torch.Generator().manual_seed(4)
    depth = torch.rand(B, H, W, generator=g) * 5 + 0.5
    _, _, K, Kinv = syn.sample(B, H, W, seed=1)
    pose = torch.randn(B, 6, generator=g) * 0.05
    P = G.projection(pose, K)
    for mode, use_rot, use_tr in (("zeros", True, True), ("border", True, True), ("zeros", False, True), ("zeros", True, False)):
        outs = []
        for side in ("dev", "cpu"):
            d_ = depth.clone().to(dev if side == "dev" else "cpu").requires_grad_(True)
            ki = Kinv.clone().to(d_.device).requires_grad_(True)
            rot = P[:, :, :3].clone().contiguous().to(d_.device).requires_grad_(True) if use_rot else None
            tr = P[:, :, 3:].clone().contiguous().to(d_.device).requires_grad_(True) if use_tr else None
            if side == "dev":
                cam = IW.pixel2cam(d_, ki)
                grid = IW.cam2pixel(cam, rot, tr, mode)
            else:
                cam = G.pixel2cam(d_, ki)
                grid = G.cam2pixel(cam, rot, tr, mode)
            wgt = torch.randn(grid.shape, generator=torch.Generator().manual_seed(9)).to(grid.device)
            wrt = [t for t in (d_, ki, rot, tr) if t is not None]
            gs = torch.autograd.grad((grid * wgt).sum() + (cam * cam).sum() * 1e-3, wrt)
            outs.append((cam.detach().cpu(), grid.detach().cpu(), [x.cpu() for x in gs]))
        (c1, g1, gr1), (c0, g0, gr0) = outs
        assert rel(c1, c0) < 1e-6 and frac_bad(g1, g0, 1e-5, 1e-5) < 1e-3
        for a, b in zip(gr1, gr0):
            assert rel(a, b) < 2e-4, (mode, use_rot, use_tr, rel(a, b))


def check_corr(dev, cases=((2, 8, 6, 12), (1, 5, 7, 13), (2, 12, 5, 8), (1, 33, 9, 20))):
    """9x9 cost volume (Back2Future): plain `correlate` and the fused pair with the idx_fwd / idx_bwd channel
    permutations, forward and all gradients, vs the oracle; W % 4 == 0 runs the register-blocked kernels, other widths
    the scalar ones."""
    from cc_amd import ops
    from oracle.corr import correlate9
    idx = [k for n in range(80, 71, -1) for k in range(n, -1, -9)]          # models/back2future.py:56-57
    idx_b = list(reversed(idx))
    g = torch.Generator().manual_seed(3)
    for (B, C, H, W) in cases:
        a0, b0, c0 = (torch.randn(B, C, H, W, generator=g) for _ in range(3))
        ad, bd, cd = leaf(a0, dev), leaf(b0, dev), leaf(c0, dev)
        ac, bc, cc_ = leaf(a0, "cpu"), leaf(b0, "cpu"), leaf(c0, "cpu")
        o = ops.correlate(ad, bd)
        r = correlate9(ac, bc)
        assert rel(o, r) < 2e-6, ("correlate", (B, C, H, W), rel(o, r))
        go = torch.randn(r.shape, generator=g)
        g1 = torch.autograd.grad(o, [ad, bd], go.to(dev))
        g0 = torch.autograd.grad(r, [ac, bc], go)
        assert max(rel(x, y) for x, y in zip(g1, g0)) < 5e-6, ("correlate grads", (B, C, H, W))
        o = ops.correlation_pair(ad, bd, cd, idx, idx_b)
        r = torch.cat((correlate9(ac, bc).index_select(1, torch.tensor(idx)),
                       correlate9(ac, cc_).index_select(1, torch.tensor(idx_b))), 1)
        assert rel(o, r) < 2e-6, ("pair", (B, C, H, W), rel(o, r))
        go = torch.randn(r.shape, generator=g)
        g1 = torch.autograd.grad(o, [ad, bd, cd], go.to(dev))
        g0 = torch.autograd.grad(r, [ac, bc, cc_], go)
        assert max(rel(x, y) for x, y in zip(g1, g0)) < 5e-6, ("pair grads", (B, C, H, W), [rel(x, y) for x, y in zip(g1, g0)])


def check_upsample2x(dev, cases=((2, 1, 5, 8, 1.0), (1, 2, 3, 6, -0.625), (2, 2, 16, 26, 20.0), (1, 1, 1, 2, 1.0), (2, 2, 4, 13, 0.625), (1, 1, 3, 1, 1.0))):
    """scale * F.interpolate(x, scale_factor=2, 'bilinear', align_corners=False) (csrc/resize.hip) and its adjoint vs ATen."""
    import torch.nn.functional as F
    from cc_amd import ops
    g = torch.Generator().manual_seed(9)
    for (B, C, H, W, sc) in cases:
        x0 = torch.randn(B, C, H, W, generator=g)
        xd, xc = leaf(x0, dev), leaf(x0, "cpu")
        y = ops.upsample_bilinear2x(xd, sc)
        r = sc * F.interpolate(xc, scale_factor=2, mode="bilinear", align_corners=False)
        assert rel(y, r) < 1e-6, ("up2x", (B, C, H, W), rel(y, r))
        go = torch.randn(r.shape, generator=g)
        g1 = torch.autograd.grad(y, [xd], go.to(dev))[0]
        g0 = torch.autograd.grad(r, [xc], go)[0]
        assert rel(g1, g0) < 2e-6, ("up2x grad", (B, C, H, W), rel(g1, g0))


def check_batch_norm(dev, cases=((2, 5, 7, 12), (3, 16, 9, 13), (2, 8, 32, 64), (2, 3, 96, 128), (4, 2, 80, 104))):
    """Training-mode BatchNorm2d (csrc/bnorm.hip) vs ATen on the CPU: output, running statistics and all gradients;
    cases on both sides of the 16 k values-per-channel switch between the one-launch and the three-launch kernels."""
    import torch.nn.functional as F
    from cc_amd import ops
    g = torch.Generator().manual_seed(5)
    for (B, C, H, W) in cases:
        x0 = torch.randn(B, C, H, W, generator=g) * 2.0 + 0.7
        w0, b0 = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
        rm0, rv0 = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
        xd, wd, bd = leaf(x0, dev), leaf(w0, dev), leaf(b0, dev)
        xc, wc, bc = leaf(x0, "cpu"), leaf(w0, "cpu"), leaf(b0, "cpu")
        rmd, rvd, rmc, rvc = rm0.clone().to(dev), rv0.clone().to(dev), rm0.clone(), rv0.clone()
        y = ops._BNTrainFn.apply(xd, wd, bd, rmd, rvd, 0.1, 1e-5)
        r = F.batch_norm(xc, rmc, rvc, wc, bc, True, 0.1, 1e-5)
        assert rel(y, r) < 2e-6, ("bn out", (B, C, H, W), rel(y, r))
        assert rel(rmd, rmc) < 2e-6 and rel(rvd, rvc) < 2e-6, ("bn running stats", rel(rmd, rmc), rel(rvd, rvc))
        go = torch.randn(r.shape, generator=g)
        g1 = torch.autograd.grad(y, [xd, wd, bd], go.to(dev))
        g0 = torch.autograd.grad(r, [xc, wc, bc], go)
        errs = [rel(a, b) for a, b in zip(g1, g0)]
        assert max(errs) < 1e-5, ("bn grads", (B, C, H, W), errs)
        # eval mode (running statistics; the validation loops): output and input gradient, statistics untouched
        rm1, rv1 = rmd.clone(), rvd.clone()
        ye = ops.batch_norm(xd, wd.detach(), bd.detach(), rmd, rvd, None, False, 0.1, 1e-5)
        re = F.batch_norm(xc, rmc, rvc, wc.detach(), bc.detach(), False, 0.1, 1e-5)
        assert rel(ye, re) < 2e-6, ("bn eval out", (B, C, H, W), rel(ye, re))
        assert torch.equal(rm1, rmd) and torch.equal(rv1, rvd)
        ge1, = torch.autograd.grad(ye, [xd], go.to(dev))
        ge0, = torch.autograd.grad(re, [xc], go)
        assert rel(ge1, ge0) < 2e-6, ("bn eval grad", (B, C, H, W), rel(ge1, ge0))
        # ... and with affine parameters that still train in eval mode (nn.BatchNorm2d allows it): all three gradients
        ye2 = ops.batch_norm(xd, wd, bd, rmd, rvd, None, False, 0.1, 1e-5)
        re2 = F.batch_norm(xc, rmc, rvc, wc, bc, False, 0.1, 1e-5)
        ga1 = torch.autograd.grad(ye2, [xd, wd, bd], go.to(dev))
        ga0 = torch.autograd.grad(re2, [xc, wc, bc], go)
        errs = [rel(a, b) for a, b in zip(ga1, ga0)]
        assert max(errs) < 1e-5, ("bn eval affine grads", (B, C, H, W), errs)
    with pytest.raises(ValueError):          # torch's own check: one value per channel cannot be normalised
        ops.batch_norm(torch.zeros(1, 3, 1, 1, device=dev), None, None, None, None, None, True, 0.1, 1e-5)


def check_corr_patch(dev, cases=((2, 6, 9, 14, 5, 1), (1, 4, 12, 10, 7, 2), (1, 3, 6, 7, 21, 2))):
    """General P x P / dilation-D cost volume (FlowNetC6) vs the oracle's restatement of spatial_correlation_sample."""
    from cc_amd import ops
    from oracle.corr import correlation_volume
    g = torch.Generator().manual_seed(11)
    for (B, C, H, W, P, D) in cases:
        a0, b0 = torch.randn(B, C, H, W, generator=g), torch.randn(B, C, H, W, generator=g)
        ad, bd, ac, bc = leaf(a0, dev), leaf(b0, dev), leaf(a0, "cpu"), leaf(b0, "cpu")
        o = ops.correlate_patch(ad, bd, P, D)
        r = correlation_volume(ac, bc, P, D).reshape(B, P * P, H, W) / C
        assert rel(o, r) < 2e-6, ("corr patch", (B, C, H, W, P, D), rel(o, r))
        go = torch.randn(r.shape, generator=g)
        g1 = torch.autograd.grad(o, [ad, bd], go.to(dev))
        g0 = torch.autograd.grad(r, [ac, bc], go)
        assert max(rel(x, y) for x, y in zip(g1, g0)) < 5e-6, ("corr patch grads", (B, C, H, W, P, D))


def check_sum_strided(dev, cases=None):
    """cc_sum_strided: out (=, +=) sum of n <= 8 [B, C, H, W] tensors that are channel slices of wider buffers (own batch strides),
    16-byte and scalar paths -- against the same sum in torch, bit for bit (same left-to-right order)."""
    import ctypes
    from cc_amd import ops
    E = ops.engine()
    g = torch.Generator().manual_seed(13)
    cases = cases or ((2, 4, 6, 8, 2, 0), (2, 3, 5, 7, 3, 1), (1, 2, 8, 12, 8, 0), (3, 2, 4, 4, 5, 1), (2, 6, 9, 16, 1, 1))
    worst = 0.0
    for (B, C, H, W, n, accumulate) in cases:
        parts, keep = [], []
        for k in range(n):
            extra = (k % 3)
            full = torch.randn(B, C + extra, H, W, generator=g).to(dev)
            keep.append(full)
            parts.append(full[:, extra:] if k % 2 == 0 else full[:, :C])
        wide = torch.randn(B, C + 2, H, W, generator=g).to(dev)
        out = wide[:, 1:1 + C]
        want = out.clone() if accumulate else None
        acc = parts[0].clone()
        for p in parts[1:]:
            acc = acc + p
        want = (want + acc) if accumulate else acc
        src = (ctypes.c_long * n)(*[p.data_ptr() for p in parts])
        sbs = (ctypes.c_long * n)(*[p.stride(0) for p in parts])
        guard = wide.clone()
        E.call("cc_sum_strided", n, ctypes.addressof(src), ctypes.addressof(sbs), out, out.stride(0), B, C * H * W, accumulate, ops.STREAM)
        assert torch.equal(out.cpu(), want.cpu()), (B, C, H, W, n, accumulate, float((out - want).abs().max()))
        assert torch.equal(wide[:, 0].cpu(), guard[:, 0].cpu()) and torch.equal(wide[:, 1 + C:].cpu(), guard[:, 1 + C:].cpu())
        worst = max(worst, float((out - want).abs().max()))
    return {"sum_strided": worst}


def check_bias_grad_table(dev, cases=None, tol=2e-5):
    """cc_bias_grad_defer + cc_bias_grad_table (+ the parked second stage, cc_wgrad_reduce_table kind 4): the bias gradients of
    many layers in one launch = the (n, h, w) sums of the pre-activation gradients accumulated onto the existing bias gradients --
    small maps (written directly), chunked planes, planes that are not a multiple of four, channel slices of a larger buffer
    (batch-strided), and more than 32 jobs (two launches)."""
    import ctypes
    from cc_amd import ops
    E = ops.engine()
    g = torch.Generator().manual_seed(11)
    cases = cases or ((2, 16, 9, 13, 0), (2, 8, 64, 160, 0), (1, 130, 5, 6, 0), (2, 4, 33, 70, 3), (3, 24, 12, 40, 8), (2, 1, 128, 130, 0),
                      (4, 3, 8, 8, 0))
    cases = tuple(cases) * 6 if len(cases) * 6 > 32 else tuple(cases)
    jobs, reds, keep, want, gbs = [], [], [], [], []
    for (B, C, H, W, extra) in cases:
        full = torch.randn(B, C + extra, H, W, generator=g).to(dev)
        gy = full[:, extra:]                                   # a channel slice: batch stride (C + extra) * H * W
        gb = torch.randn(C, generator=g).to(dev)
        want.append(gb.double().cpu() + gy.double().sum((0, 2, 3)).cpu())
        ws = torch.empty(max(E.call("cc_act_bwd_ws_bytes", C) // 4, 4), device=dev, dtype=torch.float32)
        job, red, nred = (ctypes.c_long * 12)(), (ctypes.c_long * 16)(), ctypes.c_int(0)
        E.call("cc_bias_grad_defer", gy, gb, ws, B, C, H, W, (C + extra) * H * W, 1, ctypes.addressof(job), ctypes.addressof(red),
               ctypes.addressof(nred))
        jobs.extend(job[:])
        if nred.value:
            reds.extend(red[:])
        keep.append((full, ws))
        gbs.append(gb)
    assert len(jobs) // 12 > 32 and reds, "the cases must cover two table launches and the chunked second stage"
    arr = (ctypes.c_long * len(jobs))(*jobs)
    E.call("cc_bias_grad_table", ctypes.addressof(arr), len(jobs) // 12, ops.STREAM)
    arr2 = (ctypes.c_long * len(reds))(*reds)
    E.call("cc_wgrad_reduce_table", ctypes.addressof(arr2), len(reds) // 16, ops.STREAM)
    worst = 0.0
    for gb, w in zip(gbs, want):
        err = float((gb.double().cpu() - w).abs().max() / (w.abs().max() + 1e-30))
        worst = max(worst, err)
        assert err <= tol, err
    return {"bias_grad_table": worst}


def check_concat_gradient_slices(dev, tol=2e-5):
    """The producers of a torch.cat receive narrow() views of the concat gradient: conv / transposed conv (with activation) and
    the x2 up-sampling read them in place through the kernels' batch-stride arguments (ops._slice_or_c) -- same gradients as
    stock torch, and no contiguous copy of the slice is made."""
    import torch.nn.functional as F
    from cc_amd import ops
    g = torch.Generator().manual_seed(5)
    B, H, W = 2, 8, 12
    x1 = torch.randn(B, 8, H, W, generator=g).to(dev).requires_grad_(True)
    x2 = torch.randn(B, 12, H // 2, W // 2, generator=g).to(dev).requires_grad_(True)
    x3 = torch.randn(B, 4, H // 2, W // 2, generator=g).to(dev).requires_grad_(True)
    w1 = (torch.randn(8, 8, 3, 3, generator=g) * 0.2).to(dev).requires_grad_(True)
    b1 = torch.randn(8, generator=g).to(dev).requires_grad_(True)
    w2 = (torch.randn(12, 8, 3, 3, generator=g) * 0.2).to(dev).requires_grad_(True)
    wo = (torch.randn(5, 20, 3, 3, generator=g) * 0.2).to(dev).requires_grad_(True)
    seen = []
    orig = ops._slice_or_c

    def spy(t):
        r = orig(t)
        seen.append(not r[0].is_contiguous())
        return r
    ops._slice_or_c = spy
    try:
        a = ops.conv2d(x1, w1, b1, 1, 1, act="relu")
        b = ops.conv_transpose2d(x2, w2, None, 2, 1, 1, act="relu")
        c = ops.upsample_bilinear2x(x3, 2.0)
        y = ops.conv2d(torch.cat((a, b, c), 1), wo, None, 1, 1)
        (y * y).sum().backward()
    finally:
        ops._slice_or_c = orig
    got = [t.grad.clone() for t in (x1, x2, x3, w1, b1, w2, wo)]
    for t in (x1, x2, x3, w1, b1, w2, wo):
        t.grad = None
    a = F.relu(F.conv2d(x1, w1, b1, 1, 1))
    b = F.relu(F.conv_transpose2d(x2, w2, None, 2, 1, 1))
    c = 2.0 * F.interpolate(x3, scale_factor=2, mode="bilinear", align_corners=False)
    y = F.conv2d(torch.cat((a, b, c), 1), wo, None, 1, 1)
    (y * y).sum().backward()
    for n, u, t in zip(("x1", "x2", "x3", "w1", "b1", "w2", "wo"), got, (x1, x2, x3, w1, b1, w2, wo)):
        err = float((u - t.grad).abs().max() / (t.grad.abs().max() + 1e-30))
        assert err < tol, (n, err)
    assert len(seen) == 3
    if torch.device(dev).type == "cuda":
        assert all(seen), seen          # every producer read its slice in place (CPU slices may miss the 16-byte alignment)


def check_adam(dev, n=10007, steps=3):
    """cc_adam_step against torch.optim.Adam (betas (0.9, 0.999), eps 1e-8, no weight decay: train.py:307-310) over several steps
    with a gradient scale, and cc_adam_step_segment: the update of [0, cut) with the tick followed by [cut, n) without it is bit
    for bit the one-launch update (the data-parallel step updates the big gradient segment while the small one is still being
    exchanged)."""
    from cc_amd._lib import engine, STREAM
    E = engine()
    g = torch.Generator().manual_seed(9)
    p0 = torch.randn(n, generator=g)
    ref_p = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref_p], lr=2e-4, betas=(0.9, 0.999), eps=1e-8)
    mk = lambda: [t.to(dev) for t in (p0.clone(), torch.zeros(n), torch.zeros(n), torch.zeros(1))]
    pa, ma, va, sa = mk()       # one launch
    pb, mb, vb, sb = mk()       # two segments
    pc, mc, vc, sc = mk()       # round 6: cc_adam_tick first, then three segments without a tick, last one first (per-network pipeline)
    cut = (n // 3) // 4 * 4
    scale = 0.5
    for _ in range(steps):
        grad = torch.randn(n, generator=g)
        ref_p.grad = (grad * scale).clone()
        opt.step()
        gd = grad.to(dev)
        E.call("cc_adam_step", pa, gd, ma, va, sa, n, 2e-4, 0.9, 0.999, 1e-8, scale, STREAM)
        E.call("cc_adam_step_segment", pb[:cut], gd[:cut], mb[:cut], vb[:cut], sb, cut, 2e-4, 0.9, 0.999, 1e-8, scale, 1, STREAM)
        E.call("cc_adam_step_segment", pb[cut:], gd[cut:], mb[cut:], vb[cut:], sb, n - cut, 2e-4, 0.9, 0.999, 1e-8, scale, 0, STREAM)
        E.call("cc_adam_tick", sc, STREAM)
        cut2 = cut + (n - cut) // 2 // 4 * 4
        for lo, hi in ((cut2, n), (0, cut), (cut, cut2)):
            E.call("cc_adam_step_segment", pc[lo:hi], gd[lo:hi], mc[lo:hi], vc[lo:hi], sc, hi - lo, 2e-4, 0.9, 0.999, 1e-8, scale, 0, STREAM)
    assert float(sa) == steps and float(sb) == steps and float(sc) == steps
    assert torch.equal(pa.cpu(), pb.cpu()) and torch.equal(ma.cpu(), mb.cpu()) and torch.equal(va.cpu(), vb.cpu())
    assert torch.equal(pa.cpu(), pc.cpu()) and torch.equal(ma.cpu(), mc.cpu()) and torch.equal(va.cpu(), vc.cpu())
    err = float((pa.cpu() - ref_p.detach()).abs().max())
    assert err <= 1e-6, err          # a few ulp (different but equivalent operation order: lr / bc1 folded into the step size)


WGRAD_LIST_SHAPES = [(2, 12, 9, 14, 20, 3, 2, 1), (2, 24, 6, 10, 150, 1, 1, 0), (1, 40, 5, 7, 33, 3, 1, 1), (2, 8, 12, 16, 16, 5, 2, 2),
                     (2, 130, 2, 3, 140, 3, 1, 1), (2, 16, 8, 8, 24, 4, 2, 1), (2, 48, 6, 16, 64, 3, 1, 1)]
# 3x3 / stride-1 layers that take the Winograd weight-gradient kernel (thresholds lowered by the caller): their launches are parked
# too and share multi-geometry launches (wino_wgrad.hip k_wino_wgrad_multi) -- different channel counts / map sizes / split counts,
# a G = 2 group (index 1), a width that is not a multiple of 4 (zero-padded copies first), next to a generic-kernel problem
WGRAD_LIST_SHAPES_WINO = [(2, 48, 6, 16, 64, 3, 1, 1), (1, 20, 8, 24, 40, 3, 1, 1), (2, 33, 5, 13, 40, 3, 1, 1), (2, 16, 4, 8, 24, 3, 1, 1),
                          (2, 12, 9, 14, 20, 3, 2, 1), (1, 70, 9, 32, 130, 3, 1, 1)]


# thin layers (<= 32 channels a side, >= 8192 output pixels, width a multiple of 4: wgrad_thin.hip) inside a list: problems of one kernel
# instance share launches (k_wgrad_thin_multi) -- three 3x3 / stride-1 problems of different channel counts plus a G = 3 group, two
# 3x3 / stride-2, two 1x1, a 7x7 / stride-2 on its own (the plain kernel), next to a generic-kernel problem
WGRAD_LIST_SHAPES_THIN = [(2, 16, 64, 128, 16, 3, 1, 1), (2, 32, 64, 128, 16, 3, 1, 1), (2, 12, 9, 14, 20, 3, 2, 1), (2, 16, 64, 128, 8, 3, 1, 1),
                          (2, 16, 128, 128, 32, 3, 2, 1), (2, 8, 128, 128, 16, 3, 2, 1), (2, 32, 64, 128, 16, 1, 1, 0), (2, 16, 64, 128, 24, 1, 1, 0),
                          (2, 8, 128, 128, 16, 7, 2, 3), (1, 17, 128, 128, 16, 3, 1, 1)]


def check_wgrad_list(dev, tol=2e-5, shapes=None, groups=None):
    """cc_conv2d_wgrad_list (ops._wgrad_list: what a backward stage's weight-gradient queue flushes at its end): groups of different
    shapes in one call -- stride-2 / 1x1 / small-map layers on the generic kernel (k_wgrad_multi: several per launch), direct-mode
    problems (no split) next to split ones, a G = 2 group, a weight that occurs twice (its two accumulations must not share a
    launch), and 3x3 / stride-1 layers that take other kernels inside the same list -- against torch's convolution weight
    gradients, accumulated into pre-filled buffers."""
    import torch.nn.functional as F
    from cc_amd import ops
    g = torch.Generator().manual_seed(33)

    def rn(*s):
        return torch.randn(*s, generator=g)

    def ref_grad(x, gy, shape, k, st, pad):
        w = torch.zeros(*shape, requires_grad=True)
        F.conv2d(x, w, None, st, pad).backward(gy)
        return w.grad
    # (B, Cin, H, W, Cout, k, stride, pad)
    shapes = list(shapes or WGRAD_LIST_SHAPES)
    items, want, bufs = [], [], []
    for si_, (B, Cin, H, W, Cout, k, st, pad) in enumerate(shapes):
        G = (groups or {1: 2}).get(si_, 1)          # problems per group (same-shaped layers of one launch): default a G = 2 group at index 1
        OH, OW = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
        a_l, x_l, gw_l = [], [], []
        for _ in range(G):
            x, gy, init = rn(B, Cin, H, W), rn(B, Cout, OH, OW), rn(Cout, Cin, k, k) * 0.1
            want.append(init + ref_grad(x, gy, (Cout, Cin, k, k), k, st, pad))
            bufs.append(init.clone().to(dev))
            a_l.append(gy.to(dev).contiguous())
            x_l.append(x.to(dev).contiguous())
            gw_l.append(bufs[-1])
        items.append((a_l, x_l, gw_l, (B, Cout, OH, OW, Cout * OH * OW, Cin, H, W, Cin * H * W, k, k, st, pad, Cin * k * k, k * k)))
    # problem 0 once more, with other operands, INTO THE SAME buffer (a weight used twice in one stage)
    B, Cin, H, W, Cout, k, st, pad = shapes[0]
    x2, gy2 = rn(B, Cin, H, W), rn(*items[0][0][0].shape)
    want[0] = want[0] + ref_grad(x2, gy2, (Cout, Cin, k, k), k, st, pad)
    items.append(([gy2.to(dev).contiguous()], [x2.to(dev).contiguous()], [bufs[0]], items[0][3]))
    ops.wgrad_queue.enabled = True
    try:
        ops._wgrad_list(items)
        ops.wgrad_reduces.flush()
    finally:
        ops.wgrad_queue.enabled = False
    if dev != "cpu":
        torch.cuda.synchronize()
    for i, (b, w_) in enumerate(zip(bufs, want)):
        err = float((b.cpu() - w_).abs().max()) / max(float(w_.abs().max()), 1e-30)
        assert err <= tol, ("wgrad_list problem %d" % i, err)
