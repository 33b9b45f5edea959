"""TEST INFRASTRUCTURE ONLY (oracle) -- CPU/torch-fp32 restatement of the
reference's loss layer (``/root/reference/loss_functions.py`` and ``ssim.py``).

See the header of ``oracle/geometry.py`` for who may import this and how it is
pinned to the reference.  Quirks Q1-Q8 of SURVEY.md section 8 are the spec and
are reproduced on purpose (mean over masked-out elements, whole-batch OOB
normaliser, full-resolution K in the occlusion masks, gradient_x along H, the
unused smoothness weight, the zero-padded 13x13 SSIM window ...).
"""
from math import exp

import torch
import torch.nn.functional as F

from .geometry import inverse_warp, flow_warp, pose2flow

EPS = 1e-8  # loss_functions.py:11


# ----------------------------------------------------------------------------- ssim.py
def gaussian_1d(window_size=13, sigma=1.5):
    """ssim.py:9-11 (python-float exp, fp32 tensor, normalised in fp32)."""
    g = torch.Tensor([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    return g / g.sum()


def ssim_window(window_size, channel):
    """ssim.py:13-17: outer product of the 1-D Gaussian, one copy per channel."""
    g = gaussian_1d(window_size).unsqueeze(1)
    w2 = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0)
    return w2.expand(channel, 1, window_size, window_size).contiguous()


def ssim(img1, img2, window_size=13):
    """ssim.py:19-36,68-76: per-pixel, per-channel SSIM map (NOT averaged)."""
    ch = img1.size(1)
    win = ssim_window(window_size, ch).type_as(img1)
    pad = window_size // 2
    mu1 = F.conv2d(img1, win, padding=pad, groups=ch)
    mu2 = F.conv2d(img2, win, padding=pad, groups=ch)
    mu1_sq, mu2_sq, mu12 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    s1 = F.conv2d(img1 * img1, win, padding=pad, groups=ch) - mu1_sq
    s2 = F.conv2d(img2 * img2, win, padding=pad, groups=ch) - mu2_sq
    s12 = F.conv2d(img1 * img2, win, padding=pad, groups=ch) - mu12
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return ((2 * mu12 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))


# ----------------------------------------------------------------------------- small pieces
def spatial_normalize(disp):
    """loss_functions.py:13-16."""
    m = disp.mean(dim=1, keepdim=True).mean(dim=2, keepdim=True).mean(dim=3, keepdim=True)
    return disp / m


def robust_l1_per_pix(x, q=0.5, eps=1e-2):
    """loss_functions.py:23-25."""
    return torch.pow(x.pow(2) + eps, q)


def robust_l1(x, q=0.5, eps=1e-2):
    """loss_functions.py:18-21 (mean over ALL elements, Q1)."""
    return robust_l1_per_pix(x, q, eps).mean()


def logical_or(a, b):
    """loss_functions.py:157-158."""
    return 1 - (1 - a) * (1 - b)


def occlusion_masks(flow_bw, flow_fw):
    """loss_functions.py:343-352 (signed sum, occ_fw == occ_bw: Q5)."""
    mag_sq = flow_fw.pow(2).sum(dim=1) + flow_bw.pow(2).sum(dim=1)
    thresh = 0.08 * mag_sq + 1.0
    occ_fw = (flow_fw + flow_bw).sum(dim=1) > thresh
    occ_bw = (flow_bw + flow_fw).sum(dim=1) > thresh
    return occ_bw.type_as(flow_bw), occ_fw.type_as(flow_fw)


def depth_occlusion_masks(depth, pose, intrinsics, intrinsics_inv):
    """loss_functions.py:132-137: needs 4 refs; pairs (1,2) and (0,3); full-res K (Q4)."""
    fc = [pose2flow(depth.squeeze(), pose[:, i], intrinsics, intrinsics_inv) for i in range(pose.size(1))]
    m1, m2 = occlusion_masks(fc[1], fc[2])
    m0, m3 = occlusion_masks(fc[0], fc[3])
    return torch.stack((m0, m1, m2, m3), dim=1)


def _valid(warped):
    """loss_functions.py:45,100: 1 unless all channels are EXACTLY zero (Q3)."""
    return 1 - (warped == 0).prod(1, keepdim=True).type_as(warped)


def _photo_term(tgt_s, warped, mask, wssim, qch, lambda_oob, align_corners=None):
    """The body shared by loss_functions.py:41-60 and :96-116."""
    valid = _valid(warped)
    diff = (tgt_s - warped) * valid
    ssim_loss = 1 - ssim(tgt_s, warped) * valid
    oob = valid.nelement() / valid.sum()
    assert (oob == oob).item() == 1
    for m in mask:
        diff = diff * m.expand_as(diff)
        ssim_loss = ssim_loss * m.expand_as(ssim_loss)
    return (1 - wssim) * oob * (robust_l1(diff, q=qch) + wssim * ssim_loss.mean()) \
        + lambda_oob * robust_l1(1 - valid, q=qch)


# ----------------------------------------------------------------------------- losses
def photometric_reconstruction_loss(tgt_img, ref_imgs, intrinsics, intrinsics_inv, depth,
                                    explainability_mask, pose, rotation_mode="euler",
                                    padding_mode="zeros", lambda_oob=0, qch=0.5, wssim=0.5,
                                    align_corners=False):
    """loss_functions.py:80-128."""
    def one_scale(d, emask, occ):
        assert emask is None or d.size()[2:] == emask.size()[2:]
        assert pose.size(1) == len(ref_imgs)
        b, _, h, w = d.size()
        downscale = tgt_img.size(2) / h
        tgt_s = F.adaptive_avg_pool2d(tgt_img, (h, w))
        refs_s = [F.adaptive_avg_pool2d(r, (h, w)) for r in ref_imgs]
        K_s = torch.cat((intrinsics[:, 0:2] / downscale, intrinsics[:, 2:]), dim=1)
        Kinv_s = torch.cat((intrinsics_inv[:, :, 0:2] * downscale, intrinsics_inv[:, :, 2:]), dim=2)
        total = 0
        for i, ref in enumerate(refs_s):
            warped = inverse_warp(ref, d[:, 0], pose[:, i], K_s, Kinv_s, rotation_mode, padding_mode,
                                  align_corners=align_corners)
            masks = [1 - occ[:, i:i + 1]]
            if emask is not None:
                masks.append(emask[:, i:i + 1])
            total = total + _photo_term(tgt_s, warped, masks, wssim, qch, lambda_oob)
            assert (total == total).item() == 1
        return total

    if type(explainability_mask) not in [tuple, list]:
        explainability_mask = [explainability_mask]
    if type(depth) not in [list, tuple]:
        depth = [depth]
    loss = 0
    for d, m in zip(depth, explainability_mask):
        occ = depth_occlusion_masks(d, pose, intrinsics, intrinsics_inv)
        loss = loss + one_scale(d, m, occ)
    return loss


def photometric_flow_loss(tgt_img, ref_imgs, flows, explainability_mask, lambda_oob=0, qch=0.5,
                          wssim=0.5, align_corners=False):
    """loss_functions.py:27-77."""
    def one_scale(emask, occ, fl):
        assert emask is None or fl[0].size()[2:] == emask.size()[2:]
        assert len(fl) == len(ref_imgs)
        b, _, h, w = fl[0].size()
        tgt_s = F.adaptive_avg_pool2d(tgt_img, (h, w))
        refs_s = [F.adaptive_avg_pool2d(r, (h, w)) for r in ref_imgs]
        total = 0
        for i, ref in enumerate(refs_s):
            warped = flow_warp(ref, fl[i], align_corners=align_corners)
            masks = []
            if emask is not None:
                masks.append(emask[:, i:i + 1])
            if occ is not None:
                masks.append(1 - occ[:, i:i + 1])
            total = total + _photo_term(tgt_s, warped, masks, wssim, qch, lambda_oob)
            assert (total == total).item() == 1
        return total

    if type(flows[0]) not in [tuple, list]:
        if explainability_mask is not None:
            explainability_mask = [explainability_mask]
        flows = [[uv] for uv in flows]
    loss = 0
    for i in range(len(flows[0])):
        fl = [uv[i] for uv in flows]
        occ_bw, occ_fw = occlusion_masks(fl[0], fl[1])
        occ = torch.stack((occ_bw, occ_fw), dim=1)
        loss = loss + one_scale(explainability_mask[i], occ, fl)
    return loss


def gaussian_explainability_loss(mask):
    """loss_functions.py:139-145."""
    if type(mask) not in [tuple, list]:
        mask = [mask]
    loss = 0
    for m in mask:
        loss = loss + torch.exp(-torch.mean((m - 0.5).pow(2)) / 0.15)
    return loss


def explainability_loss(mask):
    """loss_functions.py:148-155: sum over scales of BCE(mask, 1)."""
    if type(mask) not in [tuple, list]:
        mask = [mask]
    loss = 0
    for m in mask:
        loss = loss + F.binary_cross_entropy(m, torch.ones(1).expand_as(m).type_as(m))
    return loss


def consensus_exp_masks(cam_flows_fwd, cam_flows_bwd, flows_fwd, flows_bwd, tgt_img, ref_img_fwd,
                        ref_img_bwd, wssim, wrig, ws=0.1, align_corners=False):
    """loss_functions.py:160-202: per-pixel {0,1} target, non-differentiable."""
    def err(tgt_s, warped):
        return (1 - wssim) * robust_l1_per_pix(tgt_s - warped).mean(1, keepdim=True) \
            + wssim * (1 - ssim(tgt_s, warped)).mean(1, keepdim=True)

    out = []
    for i in range(len(cam_flows_fwd)):
        b, _, h, w = cam_flows_fwd[i].size()
        tgt_s = F.adaptive_avg_pool2d(tgt_img, (h, w))
        rf = F.adaptive_avg_pool2d(ref_img_fwd, (h, w))
        rb = F.adaptive_avg_pool2d(ref_img_bwd, (h, w))
        cam_f = flow_warp(rf, cam_flows_fwd[i], align_corners=align_corners)
        cam_b = flow_warp(rb, cam_flows_bwd[i], align_corners=align_corners)
        flo_f = flow_warp(rf, flows_fwd[i], align_corners=align_corners)
        valid_cam = logical_or(_valid(cam_f), _valid(cam_b))
        cam_err = torch.min(err(tgt_s, cam_f), err(tgt_s, cam_b)) * valid_cam
        flow_err = err(tgt_s, flo_f)
        out.append((wrig * cam_err <= (flow_err + EPS)).type_as(cam_err))
    return out


def compute_joint_mask_for_depth(explainability_mask, rigidity_mask_bwd, rigidity_mask_fwd, THRESH):
    """loss_functions.py:204-219 (signature kept; the train.py flag path is broken, H10)."""
    out = []
    for i in range(len(explainability_mask)):
        e = explainability_mask[i]
        rf = (rigidity_mask_fwd[i] > THRESH).type_as(e)
        rb = (rigidity_mask_bwd[i] > THRESH).type_as(e)
        ej = 1 - (1 - e[:, 1]) * (1 - e[:, 2]).unsqueeze(1) > 0.5
        jf = logical_or(rf.type_as(e), ej.type_as(e)).detach()
        jb = logical_or(rb.type_as(e), ej.type_as(e)).detach()
        out.append(torch.cat((jb, jb, jf, jf), dim=1))
    return out


def weighted_binary_cross_entropy(output, target, weights=None):
    """loss_functions.py:252-261."""
    if weights is not None:
        assert len(weights) == 2
        loss = weights[1] * (target * torch.log(output + EPS)) + \
            weights[0] * ((1 - target) * torch.log(1 - output + EPS))
    else:
        loss = target * torch.log(output + EPS) + (1 - target) * torch.log(1 - output + EPS)
    return torch.neg(torch.mean(loss))


def consensus_depth_flow_mask(explainability_mask, census_mask_bwd, census_mask_fwd,
                              exp_masks_bwd_target, exp_masks_fwd_target, THRESH, wbce):
    """loss_functions.py:221-250."""
    assert len(explainability_mask) == len(census_mask_bwd)
    assert len(explainability_mask) == len(census_mask_fwd)
    loss = 0.
    for i in range(len(explainability_mask)):
        e = explainability_mask[i]
        cf = (census_mask_fwd[i] < THRESH).type_as(e).prod(dim=1, keepdim=True)
        cb = (census_mask_bwd[i] < THRESH).type_as(e).prod(dim=1, keepdim=True)
        cf = logical_or(cf, exp_masks_fwd_target[i]).detach()
        cb = logical_or(cb, exp_masks_bwd_target[i]).detach()
        target = torch.cat((cb, cb, cf, cf), dim=1)
        loss = loss + weighted_binary_cross_entropy(e, target.type_as(e), [wbce, 1 - wbce])
    return loss


def _grad_h(t):
    return t[:, :, :-1, :] - t[:, :, 1:, :]   # loss_functions.py:288-290 "gradient_x" (along H, Q7)


def _grad_w(t):
    return t[:, :, :, :-1] - t[:, :, :, 1:]   # loss_functions.py:292-294 "gradient_y" (along W)


def edge_aware_smoothness_loss(img, pred_disp):
    """loss_functions.py:287-319 (the /2.3 weight is computed but never applied, Q7)."""
    loss = 0
    for p in pred_disp:
        b, _, h, w = p.size()
        im = F.adaptive_avg_pool2d(img, (h, w))
        wx = torch.exp(-torch.mean(torch.abs(_grad_h(im)), 1, keepdim=True))
        wy = torch.exp(-torch.mean(torch.abs(_grad_w(im)), 1, keepdim=True))
        loss = loss + torch.mean(torch.abs(_grad_h(p)) * wx) + torch.mean(torch.abs(_grad_w(p)) * wy)
    return loss


def smooth_loss(pred_disp):
    """loss_functions.py:323-341: second-order differences, weight / 2.3 per scale."""
    def grad(p):
        return p[:, :, :, 1:] - p[:, :, :, :-1], p[:, :, 1:] - p[:, :, :-1]   # (dx, dy)

    if type(pred_disp) not in [tuple, list]:
        pred_disp = [pred_disp]
    loss, weight = 0, 1.
    for p in pred_disp:
        dx, dy = grad(p)
        dx2, dxdy = grad(dx)
        dydx, dy2 = grad(dy)
        loss = loss + (dx2.abs().mean() + dxdy.abs().mean() + dydx.abs().mean() + dy2.abs().mean()) * weight
        weight /= 2.3
    return loss
