"""TEST INFRASTRUCTURE ONLY (oracle) -- plain torch (CPU, fp32, stock ATen ops)
restatement of the six reference networks BASELINE.json's configs use.

``state_dict()`` keys/shapes equal the reference's so a
weight set can be loaded into the reference, this oracle and the HIP engine
alike; that is also how the restatement is pinned (tests/golden/nets_*.npz hold
reference outputs for numpy-seeded weights).  Wiring per SURVEY.md appendix E.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .corr import correlate9
from .geometry import feature_warp


def _xavier_zero_bias(net):
    """DispResNet6.py:138-143 / PoseNetB6.py:43-48 / MaskNet6.py:54-59."""
    for m in net.modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            nn.init.xavier_uniform_(m.weight.data)
            if m.bias is not None:
                m.bias.data.zero_()


def _crop(t, ref):
    assert t.size(2) >= ref.size(2) and t.size(3) >= ref.size(3)
    return t[:, :, :ref.size(2), :ref.size(3)]


def _up2(t):
    return F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=False)


# ----------------------------------------------------------------------------- DispResNet6
class ResBlock(nn.Module):
    """models/DispResNet6.py:14-43 BasicBlock (no BN on the main path, bias-free 3x3s)."""

    def __init__(self, cin, cout, stride, shortcut):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.downsample = shortcut

    def forward(self, x):
        y = self.conv2(F.relu(self.conv1(x)))
        r = x if self.downsample is None else self.downsample(x)
        return F.relu(y + r)


def _res_stage(cin, cout, blocks, stride):
    """models/DispResNet6.py:45-60 make_layer: 1x1(stride)+BN shortcut on the first block when shapes change."""
    sc = None
    if stride != 1 or cin != cout:
        sc = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))
    seq = [ResBlock(cin, cout, stride, sc)] + [ResBlock(cout, cout, 1, None) for _ in range(1, blocks)]
    return nn.Sequential(*seq)


def _stem(cin, cout, k):
    p = (k - 1) // 2
    return nn.Sequential(nn.Conv2d(cin, cout, k, 2, p), nn.ReLU(inplace=True),
                         nn.Conv2d(cout, cout, k, 1, p), nn.ReLU(inplace=True))


def _up3(cin, cout):
    return nn.Sequential(nn.ConvTranspose2d(cin, cout, 3, 2, 1, output_padding=1), nn.ReLU(inplace=True))


def _disp_head(cin):
    return nn.Sequential(nn.Conv2d(cin, 1, 3, padding=1), nn.Sigmoid())


class DispResNet6(nn.Module):
    """models/DispResNet6.py:97-194."""

    def __init__(self, alpha=10, beta=0.01):
        super().__init__()
        self.alpha, self.beta = alpha, beta
        enc = [32, 64, 128, 256, 512, 512, 512]
        dec = [512, 512, 256, 128, 64, 32, 16]
        self.conv1 = _stem(3, enc[0], 7)
        for i in range(1, 7):
            setattr(self, "conv%d" % (i + 1), _res_stage(enc[i - 1], enc[i], 2, 2))
        ups_in = [enc[6]] + dec[:6]
        for lvl in range(7, 0, -1):
            setattr(self, "upconv%d" % lvl, _up3(ups_in[7 - lvl], dec[7 - lvl]))
        skips = {7: enc[5], 6: enc[4], 5: enc[3], 4: enc[2], 3: 1 + enc[1], 2: 1 + enc[0], 1: 1}
        for lvl in range(7, 0, -1):
            c = dec[7 - lvl]
            setattr(self, "iconv%d" % lvl, _res_stage(c + skips[lvl], c, 1, 1))
        for lvl in range(6, 0, -1):
            setattr(self, "predict_disp%d" % lvl, _disp_head(dec[7 - lvl]))

    def init_weights(self):
        _xavier_zero_bias(self)

    def forward(self, x):
        c = [x]
        for i in range(1, 8):
            c.append(getattr(self, "conv%d" % i)(c[-1]))
        out = c[7]
        disps = {}
        prev_disp = None
        for lvl in range(7, 0, -1):
            skip = c[lvl - 1]
            up = _crop(getattr(self, "upconv%d" % lvl)(out), skip)
            parts = [up] if lvl == 1 else [up, skip]
            if lvl <= 3:
                parts.append(_crop(_up2(prev_disp), skip))
            out = getattr(self, "iconv%d" % lvl)(torch.cat(parts, 1))
            if lvl <= 6:
                prev_disp = self.alpha * getattr(self, "predict_disp%d" % lvl)(out) + self.beta
                disps[lvl] = prev_disp
        if self.training:
            return tuple(disps[l] for l in range(1, 7))
        return disps[1]


# ----------------------------------------------------------------------------- DispNetS (config 1)
def _conv_relu(cin, cout):
    return nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1), nn.ReLU(inplace=True))


class DispNetS(nn.Module):
    """models/DispNetS.py:39-133 (4 scales)."""

    def __init__(self, alpha=10, beta=0.01):
        super().__init__()
        self.alpha, self.beta = alpha, beta
        enc = [32, 64, 128, 256, 512, 512, 512]
        dec = [512, 512, 256, 128, 64, 32, 16]
        ks = [7, 5, 3, 3, 3, 3, 3]
        cin = 3
        for i in range(7):
            setattr(self, "conv%d" % (i + 1), _stem(cin, enc[i], ks[i]))
            cin = enc[i]
        ups_in = [enc[6]] + dec[:6]
        skips = {7: enc[5], 6: enc[4], 5: enc[3], 4: enc[2], 3: 1 + enc[1], 2: 1 + enc[0], 1: 1}
        for lvl in range(7, 0, -1):
            setattr(self, "upconv%d" % lvl, _up3(ups_in[7 - lvl], dec[7 - lvl]))
        for lvl in range(7, 0, -1):
            setattr(self, "iconv%d" % lvl, _conv_relu(dec[7 - lvl] + skips[lvl], dec[7 - lvl]))
        for lvl in range(4, 0, -1):
            setattr(self, "predict_disp%d" % lvl, _disp_head(dec[7 - lvl]))

    def init_weights(self):
        _xavier_zero_bias(self)

    def forward(self, x):
        c = [x]
        for i in range(1, 8):
            c.append(getattr(self, "conv%d" % i)(c[-1]))
        out, disps, prev = c[7], {}, None
        for lvl in range(7, 0, -1):
            skip = c[lvl - 1]
            up = _crop(getattr(self, "upconv%d" % lvl)(out), skip)
            parts = [up] if lvl == 1 else [up, skip]
            if lvl <= 3:
                parts.append(_crop(_up2(prev), skip))
            out = getattr(self, "iconv%d" % lvl)(torch.cat(parts, 1))
            if lvl <= 4:
                prev = self.alpha * getattr(self, "predict_disp%d" % lvl)(out) + self.beta
                disps[lvl] = prev
        if self.training:
            return tuple(disps[l] for l in range(1, 5))
        return disps[1]


# ----------------------------------------------------------------------------- pose / mask nets
def _down(cin, cout, k=3):
    return nn.Sequential(nn.Conv2d(cin, cout, k, 2, (k - 1) // 2), nn.ReLU(inplace=True))


def _up4(cin, cout):
    return nn.Sequential(nn.ConvTranspose2d(cin, cout, 4, 2, 1), nn.ReLU(inplace=True))


_POSE_PLANES = [16, 32, 64, 128, 256, 256, 256, 256]
_POSE_K = [7, 5, 3, 3, 3, 3, 3, 3]


class PoseNetB6(nn.Module):
    """models/PoseNetB6.py:24-83."""

    def __init__(self, nb_ref_imgs=2):
        super().__init__()
        self.nb_ref_imgs = nb_ref_imgs
        cin = 3 * (1 + nb_ref_imgs)
        for i in range(8):
            setattr(self, "conv%d" % (i + 1), _down(cin, _POSE_PLANES[i], _POSE_K[i]))
            cin = _POSE_PLANES[i]
        self.pose_pred = nn.Conv2d(cin, 6 * nb_ref_imgs, 1)

    def init_weights(self):
        _xavier_zero_bias(self)

    def forward(self, target_image, ref_imgs):
        assert len(ref_imgs) == self.nb_ref_imgs
        x = torch.cat([target_image] + list(ref_imgs), 1)
        for i in range(8):
            x = getattr(self, "conv%d" % (i + 1))(x)
        p = self.pose_pred(x).mean(3).mean(2)
        return 0.01 * p.view(p.size(0), self.nb_ref_imgs, 6)


class MaskNet6(nn.Module):
    """models/MaskNet6.py:19-123."""

    def __init__(self, nb_ref_imgs=4, output_exp=True):
        super().__init__()
        self.nb_ref_imgs, self.output_exp = nb_ref_imgs, output_exp
        cin = 3 * (1 + nb_ref_imgs)
        for i in range(6):
            setattr(self, "conv%d" % (i + 1), _down(cin, _POSE_PLANES[i], _POSE_K[i]))
            cin = _POSE_PLANES[i]
        if output_exp:
            up = [256, 256, 128, 64, 32, 16]
            ins = [_POSE_PLANES[5]] + [up[j] + _POSE_PLANES[4 - j] for j in range(5)]
            for j, lvl in enumerate(range(6, 0, -1)):
                setattr(self, "deconv%d" % lvl, _up4(ins[j], up[j]))
            for j, lvl in enumerate(range(6, 0, -1)):
                setattr(self, "pred_mask%d" % lvl, nn.Conv2d(up[j], nb_ref_imgs, 3, padding=1))

    def init_weights(self):
        _xavier_zero_bias(self)

    def forward(self, target_image, ref_imgs):
        assert len(ref_imgs) == self.nb_ref_imgs
        x = torch.cat([target_image] + list(ref_imgs), 1)
        c = []
        for i in range(6):
            x = getattr(self, "conv%d" % (i + 1))(x)
            c.append(x)
        masks = {l: None for l in range(1, 7)}
        if self.output_exp:
            u = self.deconv6(c[5])
            masks[6] = torch.sigmoid(self.pred_mask6(u))
            for lvl in range(5, 0, -1):
                u = getattr(self, "deconv%d" % lvl)(torch.cat((u, c[lvl - 1]), 1))
                masks[lvl] = torch.sigmoid(getattr(self, "pred_mask%d" % lvl)(u))
        if self.training:
            return tuple(masks[l] for l in range(1, 7))
        return masks[1]


class PoseExpNet(nn.Module):
    """models/PoseExpNet.py:18-94 (returns (masks, pose): SURVEY.md Q12)."""

    def __init__(self, nb_ref_imgs=2, output_exp=False):
        super().__init__()
        self.nb_ref_imgs, self.output_exp = nb_ref_imgs, output_exp
        cin = 3 * (1 + nb_ref_imgs)
        for i in range(7):
            setattr(self, "conv%d" % (i + 1), _down(cin, _POSE_PLANES[i], _POSE_K[i]))
            cin = _POSE_PLANES[i]
        self.pose_pred = nn.Conv2d(cin, 6 * nb_ref_imgs, 1)
        if output_exp:
            up = [256, 128, 64, 32, 16]
            ins = [_POSE_PLANES[4]] + up[:4]
            for j, lvl in enumerate(range(5, 0, -1)):
                setattr(self, "upconv%d" % lvl, _up4(ins[j], up[j]))
            for j, lvl in enumerate(range(4, 0, -1)):
                setattr(self, "predict_mask%d" % lvl, nn.Conv2d(up[j + 1], nb_ref_imgs, 3, padding=1))

    def init_weights(self):
        _xavier_zero_bias(self)

    def forward(self, target_image, ref_imgs):
        assert len(ref_imgs) == self.nb_ref_imgs
        inp = torch.cat([target_image] + list(ref_imgs), 1)
        c = [inp]
        for i in range(7):
            c.append(getattr(self, "conv%d" % (i + 1))(c[-1]))
        p = self.pose_pred(c[7]).mean(3).mean(2)
        pose = 0.01 * p.view(p.size(0), self.nb_ref_imgs, 6)
        masks = {l: None for l in range(1, 5)}
        if self.output_exp:
            u = c[5]
            for lvl in range(5, 0, -1):
                ref = c[lvl - 1]
                u = getattr(self, "upconv%d" % lvl)(u)[:, :, 0:ref.size(2), 0:ref.size(3)]
                if lvl <= 4:
                    masks[lvl] = torch.sigmoid(getattr(self, "predict_mask%d" % lvl)(u))
        if self.training:
            return [masks[1], masks[2], masks[3], masks[4]], pose
        return masks[1], pose


# ----------------------------------------------------------------------------- Back2Future
def _feat_block(cin, cout):
    return nn.Sequential(nn.Conv2d(cin, cout, 3, 2, 1), nn.LeakyReLU(0.2),
                         nn.Conv2d(cout, cout, 3, 1, 1), nn.LeakyReLU(0.2))


def _dec_block(cin):
    chans = [cin, 128, 128, 96, 64, 32]
    layers = []
    for a, b in zip(chans[:-1], chans[1:]):
        layers += [nn.Conv2d(a, b, 3, 1, 1), nn.LeakyReLU(0.2)]
    layers.append(nn.Conv2d(32, 2, 3, 1, 1))
    return nn.Sequential(*layers)


def b2f_channel_perm():
    """models/back2future.py:56-59: idx_fwd (and its reverse idx_bwd) over the 81 displacements."""
    idx = [k for n in range(80, 71, -1) for k in range(n, -1, -9)]
    return idx, list(reversed(idx))


class Back2Future(nn.Module):
    """models/back2future.py:51-321 (``Model``); wiring per SURVEY.md appendix E."""

    FEAT = [3, 16, 32, 64, 96, 128, 192]
    DEC_IN = {6: 162, 5: 292, 4: 260, 3: 228, 2: 196}
    WARP_SCALE = {6: 0.625, 5: 1.25, 4: 2.5, 3: 5.0}

    def __init__(self, nlevels, align_corners=False):
        super().__init__()
        self.nlevels = nlevels
        self.align_corners = align_corners
        f, b = b2f_channel_perm()
        self.idx_fwd, self.idx_bwd = torch.LongTensor(f), torch.LongTensor(b)   # not buffers (as in the reference)
        for lvl in range(1, 7):
            for s in "abc":
                setattr(self, "conv%d%s" % (lvl, s), _feat_block(self.FEAT[lvl - 1], self.FEAT[lvl]))
        for lvl in range(6, 1, -1):
            setattr(self, "decoder_fwd%d" % lvl, _dec_block(self.DEC_IN[lvl]))
            setattr(self, "decoder_bwd%d" % lvl, _dec_block(self.DEC_IN[lvl]))
        self.decoder_occ6 = _dec_block(354)
        for lvl in range(5, 1, -1):
            setattr(self, "decoder_occ%d" % lvl, _dec_block(self.DEC_IN[lvl]))

    def init_weights(self):
        """models/back2future.py:106-116: bias ~ U(0,1) FIRST, then xavier weight."""
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                if m.bias is not None:
                    nn.init.uniform_(m.bias)
                nn.init.xavier_uniform_(m.weight)

    @staticmethod
    def normalize(ims):
        """models/back2future.py:118-132 (on copies)."""
        mean = (0.485, 0.456, 0.406)
        std = (0.229, 0.224, 0.225)
        out = []
        for im in ims:
            im = im * 0.5
            im = im + 0.5
            for c in range(3):
                im[:, c] = im[:, c] - mean[c]
            for c in range(3):
                im[:, c] = im[:, c] / std[c]
            out.append(im)
        return out

    def _corr_pair(self, a, b, c):
        cf = correlate9(a, b).index_select(1, self.idx_fwd)
        cb = correlate9(a, c).index_select(1, self.idx_bwd)
        return torch.cat((cf, cb), 1)

    def forward(self, im_tar, im_refs):
        n = self.normalize([im_tar] + list(im_refs))
        feats = {}
        for s, im in (("a", n[0]), ("b", n[2]), ("c", n[1])):   # b = I+, c = I-  (back2future.py:159,166)
            x = im
            for lvl in range(1, 7):
                x = getattr(self, "conv%d%s" % (lvl, s))(x)
                feats[(lvl, s)] = x
        flow_f, flow_b, up_f, up_b, occ = {}, {}, {}, {}, {}
        bw, cw = feats[(6, "b")], feats[(6, "c")]
        for lvl in range(6, 1, -1):
            a = feats[(lvl, "a")]
            corr = self._corr_pair(a, bw, cw)
            if lvl == 6:
                in_f = in_b = corr
                in_o = torch.cat((corr, a), 1)
            else:
                in_f = torch.cat((corr, a, up_f[lvl + 1]), 1)
                in_b = torch.cat((corr, a, up_b[lvl + 1]), 1)
                in_o = in_f
            flow_f[lvl] = getattr(self, "decoder_fwd%d" % lvl)(in_f)
            up_f[lvl] = _up2(flow_f[lvl])
            flow_b[lvl] = getattr(self, "decoder_bwd%d" % lvl)(in_b)
            up_b[lvl] = _up2(flow_b[lvl])
            occ[lvl] = torch.softmax(getattr(self, "decoder_occ%d" % lvl)(in_o), dim=1)
            if lvl > 2:
                s = self.WARP_SCALE[lvl]
                bw = feature_warp(feats[(lvl - 1, "b")], s * up_f[lvl], self.align_corners)
                cw = feature_warp(feats[(lvl - 1, "c")], -s * up_f[lvl], self.align_corners)   # fwd flow for both (Q9)
        full_scale = {2: 20, 3: 10, 4: 5, 5: 2.5, 6: 1.25}
        ff = [full_scale[l] * _up2(up_f[l]) for l in range(2, 7)]
        fb = [-full_scale[l] * _up2(up_b[l]) for l in range(2, 7)]
        oc = [F.interpolate(occ[l], scale_factor=4) for l in range(2, 7)]
        if self.training:
            if self.nlevels == 6:
                ff.append(0.625 * up_f[6])
                fb.append(-0.625 * up_b[6])
                oc.append(F.interpolate(occ[6], scale_factor=2))
            return ff, fb, oc
        return ff[0], fb[0], oc[0]
