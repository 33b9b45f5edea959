"""TEST INFRASTRUCTURE ONLY -- writes tests/golden/*.npz from the UNMODIFIED
reference (``/root/reference``) run on CPU under the two shims of
``oracle/ref_import.py``.  Run in the build container only:

    python -m oracle.make_golden

Inputs are re-creatable from seeds (``cc_amd/synthetic.py``, numpy streams), so
the fixtures hold reference OUTPUTS only (fp32).  Two flavours of every
grid_sample-dependent result are stored: ``acF`` = what the reference executes
under this torch (align_corners=False) and ``acT`` = the authors'
torch-1.0 semantics (grid_sample patched to align_corners=True).
"""
import os
import sys
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_import, step as S          # noqa: E402
from cc_amd import synthetic as syn               # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

# function-level case: a 6-level pyramid on a 64x96 base, B=2
FB, FH, FW = 2, 64, 96
# net/step-level case
SB, SH, SW = 2, 128, 192


def npy(t):
    return t.detach().cpu().numpy().astype(np.float32) if torch.is_tensor(t) else np.float32(t)


def pyramid_inputs(B, H, W, levels=6, seed=3):
    """Per-level depth / flows / masks for function-level loss goldens (numpy-seeded)."""
    out = []
    for l in range(levels):
        h, w = H >> l, W >> l
        ki = syn.kernel_inputs(B, h, w, seed=seed + l)
        s = 1.0 / (1 << l)
        out.append(dict(depth=ki["depth"], flow_fwd=ki["flow_fwd"] * s, flow_bwd=ki["flow_bwd"] * s, mask=ki["mask"]))
    return out


def function_level(ac, smooth=0):
    """smooth=0: white-noise frames (worst case for tap-index exactness); smooth>0: low-pass frames, where the
    losses and their gradients are insensitive to a sampling coordinate landing on the other side of an integer
    (used for gradient parity on hosts/devices whose P = K.[R|t] differs from the reference's in the last ulp)."""
    ref = ref_import.load(ac)
    iw, lf, ss = ref.inverse_warp, ref.loss_functions, ref.ssim
    tgt, refs, K, Kinv = syn.sample(FB, FH, FW, seed=1, smooth=smooth)
    pyr = pyramid_inputs(FB, FH, FW)
    pose = syn.kernel_inputs(FB, 8, 8, seed=2)["pose"] * 3.0
    g = {}
    d0 = pyr[0]["depth"][:, 0]
    g["P"] = npy(K.bmm(iw.pose_vec2mat(pose[:, 0])))
    g["pose_mat_euler"] = npy(iw.pose_vec2mat(pose[:, 0], "euler"))
    g["pose_mat_quat"] = npy(iw.pose_vec2mat(pose[:, 0] * 10, "quat"))
    cam = iw.pixel2cam(d0, Kinv)
    P = K.bmm(iw.pose_vec2mat(pose[:, 0]))
    g["grid_zeros"] = npy(iw.cam2pixel(cam, P[:, :, :3], P[:, :, -1:], "zeros"))
    g["inverse_warp"] = npy(iw.inverse_warp(refs[0], d0, pose[:, 0], K, Kinv))
    g["inverse_warp_quat"] = npy(iw.inverse_warp(refs[0], d0, pose[:, 0], K, Kinv, "quat"))
    g["pose2flow"] = npy(iw.pose2flow(d0, pose[:, 0], K, Kinv))
    g["flow_warp"] = npy(iw.flow_warp(refs[1], pyr[0]["flow_fwd"]))
    g["flow2oob"] = iw.flow2oob(pyr[0]["flow_fwd"] * 4).numpy().astype(np.uint8)
    g["ssim"] = npy(ss.ssim(tgt, refs[1]))
    g["ssim_self"] = npy(ss.ssim(tgt, tgt))
    bf = ref.back2future.Model(6)
    feat = syn.frames(FB, 16, 24, seed=7, n_frames=1)[0]
    feat = torch.cat([feat, feat.flip(1), feat * 0.5], 1)[:, :8].contiguous()
    flo = syn.kernel_inputs(FB, 16, 24, seed=8)["flow_fwd"]
    g["feature_warp"] = npy(bf.warp(feat, flo))
    g["corr9"] = npy(ref.back2future.correlate(feat, feat.flip(3)))

    # losses with gradients w.r.t. their differentiable inputs
    depth = [p["depth"].clone().requires_grad_(True) for p in pyr]
    mask = [p["mask"].clone().requires_grad_(True) for p in pyr]
    ffw = [p["flow_fwd"].clone().requires_grad_(True) for p in pyr]
    fbw = [p["flow_bwd"].clone().requires_grad_(True) for p in pyr]
    posev = pose.clone().requires_grad_(True)

    def record(name, loss, wrt):
        g[name] = npy(loss)
        grads = torch.autograd.grad(loss, [w for w in wrt.values()], allow_unused=True)
        for (k, _), gr in zip(wrt.items(), grads):
            if gr is not None:
                g[name + ".grad." + k] = npy(gr)

    wrt = {("depth%d" % i): d for i, d in enumerate(depth)}
    wrt.update({("mask%d" % i): m for i, m in enumerate(mask)})
    wrt["pose"] = posev
    record("photometric_reconstruction_loss",
           lf.photometric_reconstruction_loss(tgt, refs, K, Kinv, depth, mask, posev, wssim=0.997, qch=0.5), wrt)
    wrt = {("depth%d" % i): d for i, d in enumerate(depth)}
    wrt["pose"] = posev
    record("photometric_reconstruction_loss_nomask",
           lf.photometric_reconstruction_loss(tgt, refs, K, Kinv, depth, [None] * 6, posev, wssim=0.5, qch=0.5,
                                              lambda_oob=0.3), wrt)
    fmask = [1 - m[:, 1:3] for m in mask]
    wrt = {("flow_fwd%d" % i): f for i, f in enumerate(ffw)}
    wrt.update({("flow_bwd%d" % i): f for i, f in enumerate(fbw)})
    wrt.update({("mask%d" % i): m for i, m in enumerate(mask)})
    record("photometric_flow_loss",
           lf.photometric_flow_loss(tgt, refs[1:3], [fbw, ffw], fmask, wssim=0.997, qch=0.5), wrt)
    record("explainability_loss", lf.explainability_loss(mask), {("mask%d" % i): m for i, m in enumerate(mask)})
    record("gaussian_explainability_loss", lf.gaussian_explainability_loss(mask),
           {("mask%d" % i): m for i, m in enumerate(mask)})
    for nm, lst in (("depth", depth), ("flow_fwd", ffw), ("mask", mask)):
        record("smooth_loss." + nm, lf.smooth_loss(lst), {("%s%d" % (nm, i)): t for i, t in enumerate(lst)})
        record("edge_aware_smoothness_loss." + nm, lf.edge_aware_smoothness_loss(tgt, lst),
               {("%s%d" % (nm, i)): t for i, t in enumerate(lst)})
    with torch.no_grad():
        cam_f = [iw.pose2flow(d[:, 0], pose[:, 2], K, Kinv) for d in depth]
        cam_b = [iw.pose2flow(d[:, 0], pose[:, 1], K, Kinv) for d in depth]
        target = lf.consensus_exp_masks(cam_f, cam_b, ffw, fbw, tgt, refs[2], refs[1], wssim=0.997, wrig=1.0, ws=0.1)
        for i, t in enumerate(target):
            g["consensus_exp_masks.%d" % i] = t.numpy().astype(np.uint8)
        for i in range(6):
            g["pose2flow_fullK.%d" % i] = npy(cam_f[i])
        occ = lf.depth_occlusion_masks(depth[0], pose, K, Kinv)
        g["depth_occlusion_masks.0"] = occ.numpy().astype(np.uint8)
        rig_f = [(a - b).abs() for a, b in zip(cam_f, ffw)]
        rig_b = [(a - b).abs() for a, b in zip(cam_b, fbw)]
    record("consensus_depth_flow_mask",
           lf.consensus_depth_flow_mask(mask, rig_b, rig_f, target, target, THRESH=0.5, wbce=0.5),
           {("mask%d" % i): m for i, m in enumerate(mask)})
    return g


def _coarse(name, tensors, g, limit=4096):
    """store small tensors fully, large ones as (sum, abs-sum, strided sample)."""
    for i, t in enumerate(tensors):
        if t is None:
            continue
        k = "%s.%d" % (name, i)
        if t.numel() <= limit:
            g[k] = npy(t)
        else:
            flat = t.detach().reshape(-1)
            g[k + ".stats"] = np.array([float(flat.double().sum()), float(flat.double().abs().sum())], dtype=np.float64)
            g[k + ".sample"] = npy(flat[:: max(1, flat.numel() // 2048)])


def net_and_step_level(ac):
    ref = ref_import.load(ac)
    g = {}
    batch = syn.sample(SB, SH, SW, seed=1)
    tgt, refs, K, Kinv = batch
    nets = S.build_nets("ref", ref)
    for n in nets:
        n.load_state_dict(syn.seeded_state_dict(n, 0))
        n.train()
    cfg = S.StepConfig()
    out = S.cc_forward(nets, batch, cfg, impl=ref, keep=True)
    for k in ("loss", "loss_1", "loss_2", "loss_3", "loss_4", "loss_5"):
        g[k] = npy(out[k])
    _coarse("disparities", out["disparities"], g)
    g["pose"] = npy(out["pose"])
    _coarse("exp_mask", out["exp_mask"], g)
    _coarse("flow_fwd", out["flow_fwd"], g)
    _coarse("flow_bwd", out["flow_bwd"], g)
    _coarse("cam_fwd", out["cam_fwd"], g)
    for i, t in enumerate(out["target"]):
        g["target.%d.mean" % i] = np.float32(t.mean())
    out["loss"].backward()
    for name, n in zip(("disp", "pose", "mask", "flow"), nets):
        sq = 0.0
        for pn, p in n.named_parameters():
            if p.grad is None:
                continue
            sq += float(p.grad.double().pow(2).sum())
            if p.numel() <= 64:
                g["grad.%s.%s" % (name, pn)] = npy(p.grad)
        g["gradnorm." + name] = np.float64(sq ** 0.5)
    # one Adam step, then the loss again: pins train.py:566-568
    opt = S.make_optimizer(nets, cfg)
    opt.step()
    out2 = S.cc_forward(nets, batch, cfg, impl=ref)
    g["loss_after_adam"] = npy(out2["loss"])

    # BASELINE config 2 (DispResNet6 + PoseNetB6 only, mask None)
    nets2 = S.build_nets("ref", ref, flow=False, mask=False)
    for n in nets2:
        if n is not None:
            n.load_state_dict(syn.seeded_state_dict(n, 0))
            n.train()
    o2 = S.cc_forward(nets2, batch, cfg, impl=ref)
    g["c2.loss"], g["c2.loss_1"], g["c2.loss_3"] = npy(o2["loss"]), npy(o2["loss_1"]), npy(o2["loss_3"])

    # BASELINE config 1 (DispNetS + PoseExpNet, 1 scale, wssim=0, photometric + smooth)
    dn, pn = ref.DispNetS.DispNetS(), ref.PoseExpNet.PoseExpNet(nb_ref_imgs=4, output_exp=False)
    for n in (dn, pn):
        n.load_state_dict(syn.seeded_state_dict(n, 0))
        n.train()
    disp = dn(tgt)
    _, pose = pn(tgt, refs)
    depth = 1 / disp[0]
    # depth_occlusion_masks squeezes the depth itself (loss_functions.py:133)
    l1 = ref.loss_functions.photometric_reconstruction_loss(tgt, refs, K, Kinv, depth, None, pose, wssim=0)
    l3 = ref.loss_functions.smooth_loss(depth)
    g["c1.loss_1"], g["c1.loss_3"] = npy(l1), npy(l3)
    _coarse("c1.disp", disp, g)
    g["c1.pose"] = npy(pose)
    return g


AB, AH, AW = 2, 64, 128           # alternative-architecture case (kept small: the CPU suite runs it under the HIP emulator)
ALT_NETS = (("DispNetS6", {}), ("DispResNetS6", {}), ("PoseNet6", dict(nb_ref_imgs=4)),
            ("MaskResNet6", dict(nb_ref_imgs=4, output_exp=True)), ("FlowNetC6", dict(nlevels=6)))


def alt_net_args(name, tgt, refs):
    """Call signature per architecture: disparity nets (tgt), pose / mask nets (tgt, refs), FlowNetC6 (tgt, ref+)."""
    if name.startswith("Disp"):
        return (tgt,)
    if name == "FlowNetC6":
        return (tgt, refs[2])
    return (tgt, refs)


def alt_nets_level():
    """The alternative architectures of train.py:84-91 from the unmodified reference modules:
    train-mode outputs and parameter-gradient norms on the seeded sample with seeded weights."""
    import importlib.util
    ref_import.load(None)
    g = {}
    tgt, refs, K, Kinv = syn.sample(AB, AH, AW, seed=1)
    root = os.path.join(ref_import.REF_ROOT, "models")        # as a package: FlowNetC6 imports .submodules
    spec = importlib.util.spec_from_file_location("ccref_models_pkg", os.path.join(root, "__init__.py"),
                                                  submodule_search_locations=[root])
    pkg = importlib.util.module_from_spec(spec)
    sys.modules["ccref_models_pkg"] = pkg
    spec.loader.exec_module(pkg)
    for name, kw in ALT_NETS:
        net = getattr(pkg, name)(**kw)
        net.load_state_dict(syn.seeded_state_dict(net, 0))
        net.train()
        out = net(*alt_net_args(name, tgt, refs))
        outs = list(out) if isinstance(out, (tuple, list)) else [out]
        _coarse(name, outs, g)
        loss = sum((o * o).mean() for o in outs)
        loss.backward()
        g[name + ".gradnorm"] = np.float64(sum(float(p.grad.double().pow(2).sum()) for p in net.parameters() if p.grad is not None) ** 0.5)
    return g


def transform_inputs(seed=31, n=3, H=40, W=56):
    """Seeded 'decoded JPEG' frames as datasets/sequence_folders.py:27-28 hands them to the transforms (float32 HWC 0..255)
    and a KITTI-like intrinsics matrix."""
    r = np.random.RandomState(seed)
    frames = [np.floor(r.rand(H, W, 3) * 256).clip(0, 255).astype(np.float32) for _ in range(n)]
    K = np.array([[0.58 * W, 0, 0.49 * W], [0, 1.92 * H, 0.5 * H], [0, 0, 1]], dtype=np.float32)
    return frames, K


def load_reference_transforms():
    """The unmodified custom_transforms.py.  Its `from scipy.misc import imresize, imrotate` (gone from SciPy 1.3) is served by
    oracle/pilutil.py -- a numpy restatement of SciPy 1.1's imresize over Pillow's 8-bit resampler, pinned to Pillow itself and
    independent of cc_amd; everything else in the fixture is the reference's own arithmetic.  `imrotate` (RandomRotate, the first
    transform of train.py:178-184's pipeline) is served the same way: oracle/pilutil.imrotate restates SciPy 1.1's imrotate over
    Pillow's affine-bilinear transform and is pinned bit for bit against Pillow (tests/test_transforms.py)."""
    import importlib.util
    import types
    from oracle import pilutil

    shim = types.ModuleType("scipy.misc")
    shim.imresize, shim.imrotate = pilutil.imresize, pilutil.imrotate
    import scipy
    sys.modules["scipy.misc"] = shim
    scipy.misc = shim
    spec = importlib.util.spec_from_file_location("ccref_custom_transforms", os.path.join(ref_import.REF_ROOT, "custom_transforms.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def run_train_transform(ct, seed):
    """train.py:166-177 pipeline with Normalize(0.5, 0.5) on the seeded frames; both RNGs seeded like SequenceFolder does."""
    import random
    frames, K = transform_inputs()
    random.seed(seed)
    np.random.seed(seed)
    t = ct.Compose([ct.RandomHorizontalFlip(), ct.RandomScaleCrop(), ct.ArrayToTensor(),
                    ct.Normalize(mean=[0.5, 0.5, 0.5], std=[0.5, 0.5, 0.5])])
    imgs, Kout = t([f.copy() for f in frames], np.copy(K))
    return imgs, Kout


def run_train_transform_rotate(ct, seed):
    """train.py:178-184: the pipeline with RandomRotate first (the flow network is trained), seeded like run_train_transform."""
    import random
    frames, K = transform_inputs()
    random.seed(seed)
    np.random.seed(seed)
    t = ct.Compose([ct.RandomRotate(), ct.RandomHorizontalFlip(), ct.RandomScaleCrop(), ct.ArrayToTensor(),
                    ct.Normalize(mean=[0.5, 0.5, 0.5], std=[0.5, 0.5, 0.5])])
    imgs, Kout = t([f.copy() for f in frames], np.copy(K))
    return imgs, Kout


ROT_SEEDS = (0, 1, 2, 3, 4, 5)


def transforms_level():
    ct = load_reference_transforms()
    g = {}
    for seed in ROT_SEEDS:
        imgs, Kout = run_train_transform_rotate(ct, seed)
        g["rot.seed%d.K" % seed] = np.asarray(Kout, dtype=np.float32)
        for i, im in enumerate(imgs):
            g["rot.seed%d.img%d" % (seed, i)] = npy(im)
    frames, _ = transform_inputs()
    g["rot.direct"] = np.asarray(ct.imrotate(frames[0], 7.25))                   # the reference module's own imported name
    for seed in (0, 1, 2, 3):
        imgs, Kout = run_train_transform(ct, seed)
        g["seed%d.K" % seed] = np.asarray(Kout, dtype=np.float32)
        for i, im in enumerate(imgs):
            g["seed%d.img%d" % (seed, i)] = npy(im)
    frames, K = transform_inputs()
    sc, Ks = ct.Compose([ct.Scale(h=24, w=32), ct.ArrayToTensor()])([f.copy() for f in frames], np.copy(K))
    g["scale.K"] = np.asarray(Ks, dtype=np.float32)
    g["scale.img0"] = npy(sc[0])
    return g


def metric_inputs(seed=21, B=2, H=48, W=64, h=24, w=32):
    """Seeded inputs of the validation metrics: KITTI-like flow ground truth (u, v, valid), two predictions at half
    resolution, a soft rigidity mask, and depth ground truth / prediction with invalid (0 / > 80 m) pixels."""
    r = np.random.RandomState(seed)
    gt = np.concatenate([r.randn(B, 2, H, W).astype(np.float32) * 6.0, (r.rand(B, 1, H, W) > 0.3).astype(np.float32)], 1)
    rigid = r.randn(B, 2, h, w).astype(np.float32) * 3.0
    nonrigid = r.randn(B, 2, h, w).astype(np.float32) * 3.0
    mask = r.rand(B, 1, h, w).astype(np.float32)
    dgt = (r.rand(B, H, W).astype(np.float32) * 100.0 - 10.0)          # some <= 0 and some >= 80
    dpred = (r.rand(B, H, W).astype(np.float32) * 60.0 + 0.5)
    t = torch.from_numpy
    return dict(gt=t(gt), rigid=t(rigid), nonrigid=t(nonrigid), mask=t(mask), dgt=t(dgt), dpred=t(dpred))


def metrics_level():
    """Validation metrics (loss_functions.py:355-467) from the unmodified reference (no grid_sample inside: one flavour)."""
    ref = ref_import.load(None)
    lf = ref.loss_functions
    m = metric_inputs()
    g = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        g["flow_diff"] = npy(lf.flow_diff(m["gt"], m["rigid"]))
        g["epe3"] = np.float32(lf.compute_epe(m["gt"], m["rigid"]))
        g["epe2"] = np.float32(lf.compute_epe(m["gt"][:, :2].contiguous(), m["nonrigid"]))
        g["outlier"] = np.float32(lf.outlier_err(m["gt"], m["rigid"]))
        g["all_epes"] = np.asarray(lf.compute_all_epes(m["gt"], m["rigid"], m["nonrigid"], m["mask"]), dtype=np.float32)
        g["errors_crop"] = np.asarray([float(v) for v in lf.compute_errors(m["dgt"], m["dpred"])], dtype=np.float32)
        g["errors_nocrop"] = np.asarray([float(v) for v in lf.compute_errors(m["dgt"], m["dpred"], crop=False)], dtype=np.float32)
    return g


VB, VH, VW, VGH, VGW = 1, 64, 192, 90, 260      # validation case: batch 1 (train.py:236), ground truth larger than the net input
RIGIDITY_NAMES = ("oob_rigid", "oob_non_rigid", "rigidity_mask", "rigidity_mask_census_soft", "rigidity_mask_census_u",
                  "rigidity_mask_census_v", "rigidity_mask_census", "rigidity_mask_combined", "flow_fwd_non_rigid",
                  "flow_fwd_rigid", "total_flow")


def validate_args():
    """The fields of train.py's global `args` that validate_* read (defaults of train.py:40-120 for the CC configuration)."""
    import types
    return types.SimpleNamespace(spatial_normalize=False, flownet="Back2Future", THRESH=0.01, DEBUG=False, log_terminal=False,
                                 print_freq=10, sequence_length=5, rotation_mode="euler", padding_mode="zeros")


def validate_inputs(n=2, seed=41):
    """-> (val_flow_loader items, val_loader items): what datasets/validation_flow.py and validation_folders.py hand to
    validate_flow_with_gt / validate_depth_with_gt (train.py:650, :600) -- seeded frames, KITTI-like sparse flow ground truth
    (u, v, valid) and object map at their own resolution, depth ground truth with invalid pixels."""
    flow_items, depth_items = [], []
    r = np.random.RandomState(seed)
    for i in range(n):
        tgt, refs, K, Kinv = syn.sample(VB, VH, VW, seed=seed + i)
        gt = np.concatenate([r.randn(VB, 2, VGH, VGW).astype(np.float32) * 4.0,
                             (r.rand(VB, 1, VGH, VGW) > 0.4).astype(np.float32)], 1)
        obj = (r.rand(VB, VGH, VGW) > 0.7).astype(np.float32)
        flow_items.append((tgt, refs, K, Kinv, torch.from_numpy(gt), torch.from_numpy(obj)))
        depth = (r.rand(VB, VH, VW).astype(np.float32) * 100.0 - 10.0)
        depth_items.append((tgt, torch.from_numpy(depth)))
    return flow_items, depth_items


def validate_net_tweak(mask_net):
    """The seeded MaskNet6 answers ~0.5 everywhere, i.e. 'rigid' for every pixel; shift its finest head so that the
    composition of train.py:676 lands on both sides of 0.5 and the non-rigid branch of the loop carries pixels."""
    with torch.no_grad():
        mask_net.state_dict()["pred_mask1.bias"].fill_(-1.0)


def rigidity_inputs(seed=43, B=1, H=40, W=56, thresh=0.5):
    """Inputs of the rigidity-mask composition (train.py:673-687) that reach every branch: masks around 0.5, a third of the
    pixels with |flow_cam - flow_fwd| below the threshold on both channels, flows large enough to leave the image."""
    r = np.random.RandomState(seed)
    mask = r.rand(B, 4, H, W).astype(np.float32)
    cam = (r.randn(B, 2, H, W) * 12.0).astype(np.float32)
    near = (r.rand(B, 1, H, W) < 0.5).astype(np.float32)
    fwd = (cam + near * r.randn(B, 2, H, W) * thresh * 0.7 + (1 - near) * r.randn(B, 2, H, W) * 5.0).astype(np.float32)
    t = torch.from_numpy
    return dict(explainability_mask=t(mask), flow_cam=t(cam), flow_fwd=t(fwd), THRESH=thresh)


def reference_train_functions():
    """validate_depth_with_gt / validate_flow_with_gt / AverageMeter compiled from the UNMODIFIED source text of the reference's
    train.py and logger.py (their module-level imports -- tensorboardX, blessings, progressbar, path -- do not exist here, so
    the function nodes are lifted out of the parsed files and executed in a namespace holding the reference's own helper
    functions), plus the rigidity-composition statements of train.py:673-687 as a standalone code object."""
    import ast
    import time
    ref = ref_import.load(None)
    ns = dict(torch=torch, np=np, time=time, Variable=torch.autograd.Variable,
              spatial_normalize=ref.loss_functions.spatial_normalize, compute_errors=ref.loss_functions.compute_errors,
              compute_all_epes=ref.loss_functions.compute_all_epes, flow_diff=ref.loss_functions.flow_diff,
              pose2flow=ref.inverse_warp.pose2flow, flow2oob=ref.inverse_warp.flow2oob,
              inverse_warp=ref.inverse_warp.inverse_warp, flow_warp=ref.inverse_warp.flow_warp)
    tree = ast.parse(open(os.path.join(ref_import.REF_ROOT, "logger.py")).read())
    keep = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "AverageMeter"]
    exec(compile(ast.Module(body=keep, type_ignores=[]), "reference/logger.py", "exec"), ns)
    tree = ast.parse(open(os.path.join(ref_import.REF_ROOT, "train.py")).read())
    fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("validate_depth_with_gt", "validate_flow_with_gt")]
    assert len(fns) == 2
    exec(compile(ast.Module(body=fns, type_ignores=[]), "reference/train.py", "exec"), ns)
    vf = [n for n in fns if n.name == "validate_flow_with_gt"][0]
    loop = [n for n in vf.body if isinstance(n, ast.For)][0]
    stmts = [n for n in loop.body if isinstance(n, ast.Assign) and isinstance(n.targets[0], ast.Name)
             and n.targets[0].id in RIGIDITY_NAMES]
    assert [n.targets[0].id for n in stmts] == list(RIGIDITY_NAMES), [n.targets[0].id for n in stmts]
    ns["_rigidity_code"] = compile(ast.Module(body=stmts, type_ignores=[]), "reference/train.py:673-687", "exec")
    return ns


def validate_level():
    """validate_flow_with_gt / validate_depth_with_gt (train.py:588-777) and the rigidity composition, from the reference."""
    import types
    ns = reference_train_functions()
    g = {}
    ri = rigidity_inputs()
    loc = dict(explainability_mask=ri["explainability_mask"], flow_cam=ri["flow_cam"], flow_fwd=ri["flow_fwd"],
               args=types.SimpleNamespace(THRESH=ri["THRESH"]), flow2oob=ns["flow2oob"])
    exec(ns["_rigidity_code"], loc)
    for k in RIGIDITY_NAMES:
        g["rigidity." + k] = npy(loc[k].float())
    ref = ref_import.load(None)
    nets = S.build_nets("ref", ref)
    for n in nets:
        n.load_state_dict(syn.seeded_state_dict(n, 0))
    validate_net_tweak(nets[2])
    flow_items, depth_items = validate_inputs()
    ns["args"] = validate_args()
    with torch.no_grad():
        err, names = ns["validate_flow_with_gt"](flow_items, nets[0], nets[1], nets[2], nets[3], 0, None)
        g["flow.errors"] = np.asarray([float(e) for e in err], dtype=np.float64)
        g["flow.names"] = np.asarray(names)
        err, names = ns["validate_depth_with_gt"](depth_items, nets[0], 0, None)
        g["depth.errors"] = np.asarray([float(e) for e in err], dtype=np.float64)
        g["depth.names"] = np.asarray(names)
        ns["args"].spatial_normalize = True
        err, _ = ns["validate_depth_with_gt"](depth_items, nets[0], 0, None)
        g["depth.errors_spatial_normalize"] = np.asarray([float(e) for e in err], dtype=np.float64)
    return g


HEADLINE = (("c3_b4", True, 4, 256, 832),      # BASELINE.json configs[2]: the configuration the metric is quoted on
            ("c2_b4", False, 4, 256, 832),     # configs[1]
            ("c5_b2", True, 2, 512, 1664))     # configs[4] per-GPU shape


def headline_level(only=None):
    """Step-level goldens AT THE BENCHMARKED SIZES from the unmodified reference (as-run align_corners): the six loss
    scalars, per-net gradient norms after loss.backward(), and the loss after one Adam step (train.py:454-509,566-568).
    Inputs = what bench.py feeds (syn.sample(B, H, W, seed=1, smooth=3)), weights = syn.seeded_state_dict(net, 0).
    Scalars only: the fixture stays tiny although the run is full-size (minutes of CPU)."""
    import time
    ref = ref_import.load(None)
    g = {}
    for tag, full, B, H, W in HEADLINE:
        if only and tag not in only:
            continue
        t0 = time.time()
        batch = syn.sample(B, H, W, seed=1, smooth=3)
        nets = S.build_nets("ref", ref, flow=full, mask=full)
        for n in nets:
            if n is not None:
                n.load_state_dict(syn.seeded_state_dict(n, 0))
                n.train()
        cfg = S.StepConfig()
        out = S.cc_forward(nets, batch, cfg, impl=ref)
        for k, v in out.items():
            if k.startswith("loss"):
                g["%s.%s" % (tag, k)] = np.float64(float(v))
        out["loss"].backward()
        for name, n in zip(("disp", "pose", "mask", "flow"), nets):
            if n is None:
                continue
            sq = sum(float(p.grad.double().pow(2).sum()) for p in n.parameters() if p.grad is not None)
            g["%s.gradnorm.%s" % (tag, name)] = np.float64(sq ** 0.5)
        opt = S.make_optimizer(nets, cfg)
        opt.step()
        with torch.no_grad():
            out2 = S.cc_forward(nets, batch, cfg, impl=ref)
        g["%s.loss_after_adam" % tag] = np.float64(float(out2["loss"]))
        print("headline %s: loss %.8f  (%.0f s)" % (tag, float(out["loss"]), time.time() - t0), flush=True)
    return g


def main():
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "headline":
        torch.set_num_threads(os.cpu_count() or 1)      # scalars at 1e-4: thread-count rounding is irrelevant here
        path = os.path.join(OUT, "headline.npz")
        g = dict(np.load(path)) if os.path.exists(path) and len(sys.argv) > 2 else {}
        g.update(headline_level(sys.argv[2:] or None))
        np.savez_compressed(path, **g)
        print("wrote headline")
        return
    torch.set_num_threads(1)   # run-to-run bit reproducibility of the fixtures
    if len(sys.argv) > 1 and sys.argv[1] == "validate":
        np.savez_compressed(os.path.join(OUT, "validate.npz"), **validate_level())
        print("wrote validate")
        return
    np.savez_compressed(os.path.join(OUT, "metrics.npz"), **metrics_level())
    print("wrote metrics")
    if len(sys.argv) > 1 and sys.argv[1] == "metrics":
        return
    if len(sys.argv) > 1 and sys.argv[1] == "transforms":
        np.savez_compressed(os.path.join(OUT, "transforms.npz"), **transforms_level())
        print("wrote transforms")
        return
    np.savez_compressed(os.path.join(OUT, "transforms.npz"), **transforms_level())
    np.savez_compressed(os.path.join(OUT, "altnets.npz"), **alt_nets_level())
    print("wrote altnets")
    if len(sys.argv) > 1 and sys.argv[1] == "altnets":
        return
    for tag, ac in (("acF", None), ("acT", True)):
        np.savez_compressed(os.path.join(OUT, "functions_%s.npz" % tag), **function_level(ac))
        sm = function_level(ac, smooth=3)
        keep = {k: v for k, v in sm.items() if k.split(".")[0] in (
            "photometric_reconstruction_loss", "photometric_reconstruction_loss_nomask", "photometric_flow_loss")}
        np.savez_compressed(os.path.join(OUT, "functions_%s_smooth.npz" % tag), **keep)
        np.savez_compressed(os.path.join(OUT, "step_%s.npz" % tag), **net_and_step_level(ac))
        print("wrote", tag)
    ref_import.set_align_corners(None)


if __name__ == "__main__":
    main()
