"""Validation loops of the reference's train.py: validate_depth_with_gt (train.py:588-635) and
validate_flow_with_gt (train.py:637-774) on the engine's networks and kernels, same names, argument order and return values
(`errors.avg, error_names`).

What differs from the reference, deliberately:
  * train.py reads a module-global `args`; here the same namespace is passed as `args=` (fields: spatial_normalize, flownet,
    THRESH, and -- only for the optional per-sample hook -- rotation_mode / padding_mode).
  * the ~25 stock-torch launches of the rigidity-mask composition (train.py:673-687) are ONE HIP launch (`cc_rigidity_compose`,
    cc_amd/csrc/validate.hip) -> `rigidity_composition`.
  * the metrics stay on the device (`sync=False`) and are read back once at the end of the loop instead of eight `.item()` host
    syncs per sample.
  * TensorBoard image / histogram logging and the terminal progress bar (train.py:615-633,700-741,758-768) are UI and out of
    scope; `on_sample(i, dict_of_intermediates)` is the hook a caller's own logging attaches to -- it receives every tensor the
    reference logs.
The reference composes the masks with batch size 1 only (train.py:236: KITTI images differ in size); its broadcasting at
train.py:676 mixes samples for B > 1, so per-sample semantics are what is implemented (identical for B = 1).
"""
import types

import torch

from . import loss_functions as LF
from ._lib import engine, STREAM
from .inverse_warp import pose2flow
from .logger import AverageMeter

RIGIDITY_FIELDS = ("rigidity_mask", "rigidity_mask_census", "rigidity_mask_combined", "flow_fwd_non_rigid", "flow_fwd_rigid",
                   "total_flow", "oob_rigid", "oob_non_rigid")
DEPTH_ERROR_NAMES = ['abs_diff', 'abs_rel', 'sq_rel', 'a1', 'a2', 'a3']
FLOW_ERROR_NAMES = ['epe_total', 'epe_rigid', 'epe_non_rigid', 'outliers', 'epe_total_with_gt_mask', 'epe_rigid_with_gt_mask',
                    'epe_non_rigid_with_gt_mask', 'outliers_gt_mask']


def rigidity_composition(explainability_mask, flow_cam, flow_fwd, THRESH, want=RIGIDITY_FIELDS):
    """train.py:673-687 in one launch.  explainability_mask [B,MC>=3,H,W], flow_cam / flow_fwd [B,2,H,W] -> namespace with
    rigidity_mask [B,1,H,W], rigidity_mask_census [B,H,W], rigidity_mask_combined [B,1,H,W], flow_fwd_non_rigid, flow_fwd_rigid,
    total_flow [B,2,H,W] (fp32, masks as 0/1) and oob_rigid / oob_non_rigid [B,H,W] (bool, inverse_warp.py:222-238).
    `want`: the fields to produce (the others are None and cost no HBM traffic)."""
    B, MC, H, W = explainability_mask.shape
    assert flow_cam.shape == (B, 2, H, W) and flow_fwd.shape == (B, 2, H, W), "rigidity_composition: flows must be [B,2,H,W] of the mask"
    m = explainability_mask.detach().float().contiguous()
    fc, ff = flow_cam.detach().float().contiguous(), flow_fwd.detach().float().contiguous()
    shapes = dict(rigidity_mask=(B, 1, H, W), rigidity_mask_census=(B, H, W), rigidity_mask_combined=(B, 1, H, W),
                  flow_fwd_non_rigid=(B, 2, H, W), flow_fwd_rigid=(B, 2, H, W), total_flow=(B, 2, H, W),
                  oob_rigid=(B, H, W), oob_non_rigid=(B, H, W))
    unknown = set(want) - set(shapes)
    assert not unknown, "rigidity_composition: unknown fields %s" % sorted(unknown)
    o = {k: (torch.empty(shapes[k], dtype=torch.float32, device=m.device) if k in want else None) for k in RIGIDITY_FIELDS}
    engine().call("cc_rigidity_compose", m, MC, fc, ff, o["rigidity_mask"], o["rigidity_mask_census"], o["rigidity_mask_combined"],
                  o["flow_fwd_non_rigid"], o["flow_fwd_rigid"], o["total_flow"], o["oob_rigid"], o["oob_non_rigid"],
                  float(THRESH), B, H, W, STREAM)
    for k in ("oob_rigid", "oob_non_rigid"):
        if o[k] is not None:
            o[k] = o[k] > 0.5
    return types.SimpleNamespace(**o)


def _device_of(net):
    return next(net.parameters()).device


def _finish(errors):
    """the loop's only host read-back"""
    avg = [a.detach().double() if torch.is_tensor(a) else torch.tensor(float(a), dtype=torch.float64) for a in errors.avg]
    dev = next((a.device for a in avg if a.is_cuda), torch.device("cpu"))
    return torch.stack([a.to(dev) for a in avg]).tolist()


def validate_depth_with_gt(val_loader, disp_net, epoch=0, logger=None, output_writers=(), args=None, on_sample=None):
    """train.py:588-635: Eigen depth errors of disp_net over (tgt_img, depth_gt) batches."""
    error_names = list(DEPTH_ERROR_NAMES)
    errors = AverageMeter(i=len(error_names))
    disp_net.eval()                                                            # :596
    dev = _device_of(disp_net)
    with torch.no_grad():
        for i, (tgt_img, depth) in enumerate(val_loader):
            tgt_img = tgt_img.to(dev)
            output_disp = disp_net(tgt_img)                                    # :602
            if args is not None and getattr(args, "spatial_normalize", False):
                output_disp = LF.spatial_normalize(output_disp)                # :603-604
            output_depth = 1 / output_disp                                     # :606
            depth = depth.to(dev)
            errors.update(LF.compute_errors(depth, output_depth.squeeze(1)))   # :624 (0-dim device tensors, no sync)
            if on_sample is not None:
                on_sample(i, dict(tgt_img=tgt_img, depth=depth, output_disp=output_disp, output_depth=output_depth))
    return _finish(errors), error_names


def validate_flow_with_gt(val_loader, disp_net, pose_net, mask_net, flow_net, epoch=0, logger=None, output_writers=(),
                          args=None, on_sample=None):
    """train.py:637-774: end-point errors of the composed (rigid + non-rigid) flow against KITTI flow ground truth, once with
    the predicted rigidity mask and once with the ground-truth object map."""
    assert args is not None, "validate_flow_with_gt: pass the training arguments namespace as args= (THRESH, flownet, spatial_normalize)"
    error_names = list(FLOW_ERROR_NAMES)
    errors = AverageMeter(i=len(error_names))
    for net in (disp_net, pose_net, mask_net, flow_net):                      # :644-648
        net.eval()
    dev = _device_of(disp_net)
    nan_seen = None
    with torch.no_grad():
        for i, (tgt_img, ref_imgs, intrinsics, intrinsics_inv, flow_gt, obj_map_gt) in enumerate(val_loader):
            tgt_img = tgt_img.to(dev)
            ref_imgs = [img.to(dev) for img in ref_imgs]
            intrinsics, intrinsics_inv = intrinsics.to(dev), intrinsics_inv.to(dev)
            flow_gt, obj_map_gt = flow_gt.to(dev), obj_map_gt.to(dev)
            disp = disp_net(tgt_img)                                           # :659
            if getattr(args, "spatial_normalize", False):
                disp = LF.spatial_normalize(disp)
            depth = 1 / disp
            pose = pose_net(tgt_img, ref_imgs)
            explainability_mask = mask_net(tgt_img, ref_imgs)
            if getattr(args, "flownet", "Back2Future") == 'Back2Future':       # :666-670
                flow_fwd, flow_bwd, _ = flow_net(tgt_img, ref_imgs[1:3])
            else:
                flow_fwd = flow_net(tgt_img, ref_imgs[2])
                flow_bwd = flow_net(tgt_img, ref_imgs[1])
            flow_cam = pose2flow(depth.squeeze(1), pose[:, 2], intrinsics, intrinsics_inv)       # :672
            r = rigidity_composition(explainability_mask, flow_cam, flow_fwd, args.THRESH,
                                     want=RIGIDITY_FIELDS if on_sample is not None else ("rigidity_mask_combined", "total_flow"))
            obj_map_gt_expanded = obj_map_gt.unsqueeze(1).type_as(flow_fwd)    # :689
            bad = torch.isnan(flow_gt.sum()) | torch.isnan(r.total_flow.sum())  # :746 (device flag; reported after the loop)
            nan_seen = bad if nan_seen is None else (nan_seen | bad)
            _epe_errors = LF.compute_all_epes(flow_gt, flow_cam, flow_fwd, r.rigidity_mask_combined, sync=False) + \
                LF.compute_all_epes(flow_gt, flow_cam, flow_fwd, (1 - obj_map_gt_expanded), sync=False)   # :748
            errors.update(_epe_errors)
            if on_sample is not None:
                on_sample(i, dict(tgt_img=tgt_img, ref_imgs=ref_imgs, flow_gt=flow_gt, depth=depth, pose=pose,
                                  explainability_mask=explainability_mask, flow_fwd=flow_fwd, flow_bwd=flow_bwd, flow_cam=flow_cam,
                                  epe_errors=_epe_errors, **vars(r)))
    if nan_seen is not None and bool(nan_seen):
        print('NaN encountered')                                               # :747
    return _finish(errors), error_names
