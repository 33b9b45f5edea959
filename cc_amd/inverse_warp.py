"""Drop-in for the reference's ``inverse_warp`` module (inverse_warp.py), backed by the fused
gfx950 kernels of cc_amd/csrc/warp.hip through the C ABI (include/ccengine.h).

Same names, argument order, defaults and error messages as the reference
(``from inverse_warp import inverse_warp, pose2flow, flow2oob, flow_warp`` -- train.py:22;
``pose_vec2mat`` -- test_pose.py:11).  Differences, all explicit:

* ``align_corners`` is a keyword (default: ``cc_amd.config.align_corners`` = False, i.e. what the
  unmodified reference executes under a current torch; True = the authors' torch-1.0 semantics);
* no module-level pixel-grid cache (inverse_warp.py:10-20), nothing calls ``.cuda()``;
* tensors must live on a HIP device: there is no CPU path.

The kernel boundary sits at P = K.[R|t] (12 floats per sample): the tiny
pose -> matrix algebra below runs as stock torch ops (differentiable), everything per-pixel runs in HIP.
"""
import torch

from . import config
from ._lib import engine, STREAM


def check_sizes(input, input_name, expected):
    """inverse_warp.py:23-28 (same assertion text)."""
    condition = [input.ndimension() == len(expected)]
    for i, size in enumerate(expected):
        if size.isdigit():
            condition.append(input.size(i) == int(size))
    assert all(condition), "wrong size for {}, expected {}, got  {}".format(
        input_name, 'x'.join(expected), list(input.size()))


# ------------------------------------------------------------------ pixel2cam / cam2pixel (stand-alone halves of the fused warp)
class _Pixel2CamFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, Kinv):
        d, Ki = _f32c(depth), _f32c(Kinv)
        B, H, W = d.shape
        cam = torch.empty(B, 3, H, W, device=d.device, dtype=torch.float32)
        engine().call("cc_pixel2cam", d, Ki, cam, B, H, W, STREAM)
        ctx.save_for_backward(d, Ki)
        return cam

    @staticmethod
    def backward(ctx, g):
        d, Ki = ctx.saved_tensors
        B, H, W = d.shape
        E = engine()
        gd = torch.empty_like(d) if ctx.needs_input_grad[0] else None
        gk = torch.empty(B, 12, device=d.device, dtype=torch.float32)
        ws = torch.empty(int(E.call("cc_warp_partials_bytes", B, H, W)) // 4, device=d.device, dtype=torch.float32)
        E.call("cc_pixel2cam_bwd", _f32c(g), d, Ki, gd, gk, ws, B, H, W, STREAM)
        return gd, (gk[:, :9].reshape(B, 3, 3) if ctx.needs_input_grad[1] else None)


class _Cam2PixelFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cam, rot, tr, mode):
        c = _f32c(cam)
        B, _, H, W = c.shape
        P = torch.zeros(B, 3, 4, device=c.device, dtype=torch.float32)
        if rot is not None:
            P[:, :, :3] = rot
        if tr is not None:
            P[:, :, 3:] = tr
        P = P.reshape(B, 12)
        grid = torch.empty(B, H, W, 2, device=c.device, dtype=torch.float32)
        engine().call("cc_cam2pixel", c, P, grid, B, H, W, int(rot is not None), int(tr is not None),
                      1 if mode == 'zeros' else 0, STREAM)
        ctx.save_for_backward(c, P)
        ctx.cfg = (rot is not None, tr is not None, mode)
        return grid

    @staticmethod
    def backward(ctx, g):
        c, P = ctx.saved_tensors
        has_rot, has_tr, mode = ctx.cfg
        B, _, H, W = c.shape
        E = engine()
        gcam = torch.empty_like(c) if ctx.needs_input_grad[0] else None
        gP = torch.empty(B, 12, device=c.device, dtype=torch.float32)
        ws = torch.empty(int(E.call("cc_warp_partials_bytes", B, H, W)) // 4, device=c.device, dtype=torch.float32)
        E.call("cc_cam2pixel_bwd", _f32c(g), c, P, gcam, gP, ws, B, H, W, int(has_rot), int(has_tr), 1 if mode == 'zeros' else 0,
               STREAM)
        gP = gP.reshape(B, 3, 4)
        grot = gP[:, :, :3].contiguous() if (has_rot and ctx.needs_input_grad[1]) else None
        gtr = gP[:, :, 3:].contiguous() if (has_tr and ctx.needs_input_grad[2]) else None
        return gcam, grot, gtr, None


def pixel2cam(depth, intrinsics_inv):
    """inverse_warp.py:31-45: depth [B,H,W], intrinsics_inv [B,3,3] -> camera-frame points [B,3,H,W]."""
    check_sizes(depth, 'depth', 'BHW')
    check_sizes(intrinsics_inv, 'intrinsics', 'B33')
    return _Pixel2CamFn.apply(depth, intrinsics_inv)


def cam2pixel(cam_coords, proj_c2p_rot, proj_c2p_tr, padding_mode):
    """inverse_warp.py:48-79: camera-frame points [B,3,H,W] -> normalised sampling grid [B,H,W,2] (padding_mode 'zeros':
    out-of-range coordinates rewritten to 2)."""
    return _Cam2PixelFn.apply(cam_coords, proj_c2p_rot, proj_c2p_tr, padding_mode)


# ------------------------------------------------------------------ pose algebra (tiny, torch)
def euler2mat(angle):
    """inverse_warp.py:82-119: R = Rx.Ry.Rz, [B,3] -> [B,3,3]."""
    B = angle.size(0)
    x, y, z = angle[:, 0], angle[:, 1], angle[:, 2]
    zero = z.detach() * 0
    one = zero.detach() + 1
    cz, sz = torch.cos(z), torch.sin(z)
    cy, sy = torch.cos(y), torch.sin(y)
    cx, sx = torch.cos(x), torch.sin(x)
    zmat = torch.stack([cz, -sz, zero, sz, cz, zero, zero, zero, one], dim=1).view(B, 3, 3)
    ymat = torch.stack([cy, zero, sy, zero, one, zero, -sy, zero, cy], dim=1).view(B, 3, 3)
    xmat = torch.stack([one, zero, zero, zero, cx, -sx, zero, sx, cx], dim=1).view(B, 3, 3)
    return xmat.bmm(ymat).bmm(zmat)


def quat2mat(quat):
    """inverse_warp.py:122-143."""
    nq = torch.cat([quat[:, :1].detach() * 0 + 1, quat], dim=1)
    nq = nq / nq.norm(p=2, dim=1, keepdim=True)
    w, x, y, z = nq[:, 0], nq[:, 1], nq[:, 2], nq[:, 3]
    B = quat.size(0)
    w2, x2, y2, z2 = w.pow(2), x.pow(2), y.pow(2), z.pow(2)
    wx, wy, wz, xy, xz, yz = w * x, w * y, w * z, x * y, x * z, y * z
    return torch.stack([w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
                        2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
                        2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2], dim=1).view(B, 3, 3)


def pose_vec2mat(vec, rotation_mode='euler'):
    """inverse_warp.py:146-162: [B,6] (tx,ty,tz,rx,ry,rz) -> [B,3,4]."""
    translation = vec[:, :3].unsqueeze(-1)
    rot = vec[:, 3:]
    if rotation_mode == 'euler':
        rot_mat = euler2mat(rot)
    elif rotation_mode == 'quat':
        rot_mat = quat2mat(rot)
    return torch.cat([rot_mat, translation], dim=2)


class _PoseProjFn(torch.autograd.Function):
    """pose [B,6] (may be a strided slice of [B,R,6]) -> P = (K rows 0,1 / k_div) . [Rx.Ry.Rz | t], flat [B,12]."""

    @staticmethod
    def forward(ctx, pose, K, k_div):
        pose = pose.float()
        if pose.stride(1) != 1:
            pose = pose.contiguous()
        Kc = K.detach().contiguous().float()
        B = pose.shape[0]
        P = torch.empty(B, 12, device=pose.device, dtype=torch.float32)
        engine().call("cc_pose_proj_fwd", pose.data_ptr(), pose.stride(0), Kc, P, B, float(k_div), STREAM)
        ctx.save_for_backward(pose, Kc)
        ctx.k_div = float(k_div)
        return P

    @staticmethod
    def backward(ctx, gP):
        pose, Kc = ctx.saved_tensors
        B = pose.shape[0]
        g = torch.empty(B, 6, device=pose.device, dtype=torch.float32)
        engine().call("cc_pose_proj_bwd", gP.contiguous().float(), pose.data_ptr(), pose.stride(0), Kc, g, 6, B, ctx.k_div, 0,
                      STREAM)
        return g, None, None


def projection_matrix(pose, intrinsics, rotation_mode='euler', k_div=1.0):
    """inverse_warp.py:214 / :278: P = K.[R|t], returned flat [B,12] (the kernels' input).  k_div divides rows 0,1
    of K (the per-scale intrinsics of loss_functions.py:91)."""
    if rotation_mode == 'euler':
        return _PoseProjFn.apply(pose, intrinsics, k_div)
    K = intrinsics if k_div == 1.0 else torch.cat((intrinsics[:, 0:2] / k_div, intrinsics[:, 2:]), dim=1)
    return K.bmm(pose_vec2mat(pose, rotation_mode)).reshape(-1, 12)


def _ac(align_corners):
    return int(config.align_corners if align_corners is None else bool(align_corners))


def _f32c(t):
    return t.contiguous().float()


# ------------------------------------------------------------------ autograd bindings
class _InverseWarpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, depth, P, Kinv, border, ac):
        img, depth, P, Kinv = _f32c(img), _f32c(depth), _f32c(P), _f32c(Kinv)
        B, C, H, W = img.shape
        out = torch.empty_like(img)
        engine().call("cc_inverse_warp_fwd", img, depth, P, Kinv, out, B, C, H, W, border, ac, STREAM)
        ctx.save_for_backward(img, depth, P, Kinv)
        ctx.cfg = (border, ac)
        return out

    @staticmethod
    def backward(ctx, gout):
        img, depth, P, Kinv = ctx.saved_tensors
        border, ac = ctx.cfg
        B, C, H, W = img.shape
        E = engine()
        gout = _f32c(gout)
        gdepth = torch.empty_like(depth)
        gP = torch.empty_like(P)
        ws = torch.empty(E.call("cc_warp_partials_bytes", B, H, W) // 4, device=img.device, dtype=torch.float32)
        gimg = torch.zeros_like(img) if ctx.needs_input_grad[0] else None
        E.call("cc_inverse_warp_bwd", gout, img, depth, P, Kinv, gdepth, gP, gimg, ws, B, C, H, W, border, ac, STREAM)
        return gimg, gdepth, gP, None, None, None


class _Pose2FlowFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, P, Kinv, rewrite):
        depth, P, Kinv = _f32c(depth), _f32c(P), _f32c(Kinv)
        B, H, W = depth.shape
        flow = torch.empty(B, 2, H, W, device=depth.device, dtype=torch.float32)
        engine().call("cc_pose2flow_fwd", depth, P, Kinv, flow, B, H, W, rewrite, STREAM)
        ctx.save_for_backward(depth, P, Kinv)
        ctx.rewrite = rewrite
        ctx.set_materialize_grads(False)       # a consumer that returns no gradient (thresholds) must not cost a backward pass
        return flow

    @staticmethod
    def backward(ctx, gflow):
        if gflow is None:
            return None, None, None, None
        depth, P, Kinv = ctx.saved_tensors
        B, H, W = depth.shape
        E = engine()
        gdepth = torch.empty_like(depth)
        gP = torch.empty_like(P)
        ws = torch.empty(E.call("cc_warp_partials_bytes", B, H, W) // 4, device=depth.device, dtype=torch.float32)
        E.call("cc_pose2flow_bwd", _f32c(gflow), depth, P, Kinv, gdepth, gP, ws, B, H, W, ctx.rewrite, STREAM)
        return gdepth, gP, None, None


class _FlowWarpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, flow, border, ac, feature, fscale=1.0):
        img, flow = _f32c(img), _f32c(flow)
        ctx.fscale = fscale
        B, C, H, W = img.shape
        out = torch.empty_like(img)
        if feature:
            engine().call("cc_feature_warp_fwd", img, flow, out, B, C, H, W, ac, float(fscale), STREAM)
        else:
            engine().call("cc_flow_warp_fwd", img, flow, out, B, C, H, W, border, ac, STREAM)
        ctx.save_for_backward(img, flow)
        ctx.cfg = (border, ac, feature)
        return out

    @staticmethod
    def backward(ctx, gout):
        img, flow = ctx.saved_tensors
        border, ac, feature = ctx.cfg
        B, C, H, W = img.shape
        gflow = torch.empty_like(flow) if ctx.needs_input_grad[1] else None
        gimg = torch.zeros_like(img) if ctx.needs_input_grad[0] else None
        if feature:
            engine().call("cc_feature_warp_bwd", _f32c(gout), img, flow, gflow, gimg, B, C, H, W, ac, float(ctx.fscale), STREAM)
        else:
            engine().call("cc_flow_warp_bwd", _f32c(gout), img, flow, gflow, gimg, B, C, H, W, border, ac, STREAM)
        return gimg, gflow, None, None, None, None


def _border(padding_mode):
    if padding_mode == 'zeros':
        return 0
    if padding_mode == 'border':
        return 1
    raise ValueError("padding_mode must be 'zeros' or 'border', got %r" % (padding_mode,))


# ------------------------------------------------------------------ public API (reference signatures)
def inverse_warp(img, depth, pose, intrinsics, intrinsics_inv, rotation_mode='euler', padding_mode='zeros',
                 align_corners=None):
    """inverse_warp.py:250-283: warp a source image to the target image plane."""
    check_sizes(img, 'img', 'B3HW')
    check_sizes(depth, 'depth', 'BHW')
    check_sizes(pose, 'pose', 'B6')
    check_sizes(intrinsics, 'intrinsics', 'B33')
    check_sizes(intrinsics_inv, 'intrinsics', 'B33')
    assert intrinsics_inv.size() == intrinsics.size()
    P = projection_matrix(pose, intrinsics, rotation_mode)
    return _InverseWarpFn.apply(img, depth, P, intrinsics_inv, _border(padding_mode), _ac(align_corners))


def pose2flow(depth, pose, intrinsics, intrinsics_inv, rotation_mode='euler', padding_mode=None):
    """inverse_warp.py:195-220: pose parameters -> rigid optical flow [B,2,H,W]."""
    check_sizes(depth, 'depth', 'BHW')
    check_sizes(pose, 'pose', 'B6')
    check_sizes(intrinsics, 'intrinsics', 'B33')
    check_sizes(intrinsics_inv, 'intrinsics', 'B33')
    assert intrinsics_inv.size() == intrinsics.size()
    P = projection_matrix(pose, intrinsics, rotation_mode)
    return _Pose2FlowFn.apply(depth, P, intrinsics_inv, 1 if padding_mode == 'zeros' else 0)


def flow_warp(img, flow, padding_mode='zeros', align_corners=None):
    """inverse_warp.py:164-192."""
    check_sizes(img, 'img', 'BCHW')
    check_sizes(flow, 'flow', 'B2HW')
    return _FlowWarpFn.apply(img, flow, _border(padding_mode), _ac(align_corners), False)


def feature_warp(x, flo, align_corners=None, flow_scale=1.0):
    """models/back2future.py:287-321 Model.warp (border padding) of x by flo * flow_scale."""
    return _FlowWarpFn.apply(x, flo, 1, _ac(align_corners), True, float(flow_scale))


def flow2oob(flow):
    """inverse_warp.py:222-238 (validation-side helper; elementwise, stock torch)."""
    check_sizes(flow, 'flow', 'B2HW')
    bs, _, h, w = flow.size()
    u, v = flow[:, 0], flow[:, 1]
    gx = torch.arange(0, w, device=flow.device).view(1, 1, w).expand(1, h, w).type_as(u).expand_as(u)
    gy = torch.arange(0, h, device=flow.device).view(1, h, 1).expand(1, h, w).type_as(v).expand_as(v)
    X = 2 * ((gx + u) / (w - 1.0) - 0.5)
    Y = 2 * ((gy + v) / (h - 1.0) - 0.5)
    return (X.abs() > 1).add(Y.abs() > 1) > 0
