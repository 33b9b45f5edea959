"""models.Back2Future (reference models/back2future.py ``Model``) on the gfx950 kernels: MFMA convs, the HIP 9x9
cost volume (replacing the spatial_correlation_sampler CUDA extension) with idx_fwd/idx_bwd + cat folded into its
store, and the HIP border-padded feature warp.  Device-agnostic construction (no ``.cuda()`` in __init__)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import config
from .. import nn as L
from .. import ops
from ..inverse_warp import feature_warp
from ..tape import run_network


def conv_feat_block(nIn, nOut):
    """back2future.py:27-33: conv s2 + LeakyReLU(0.2), conv s1 + LeakyReLU(0.2) (parameters at .0 and .2)."""
    return nn.Sequential(L.Conv2d(nIn, nOut, 3, 2, 1, act="lrelu"), L.Act(),
                         L.Conv2d(nOut, nOut, 3, 1, 1, act="lrelu"), L.Act())


def conv_dec_block(nIn):
    """back2future.py:35-48: nIn->128->128->96->64->32->2, LeakyReLU(0.2) between (parameters at .0,.2,...,.10)."""
    chans = [nIn, 128, 128, 96, 64, 32]
    layers = []
    for a, b in zip(chans[:-1], chans[1:]):
        layers += [L.Conv2d(a, b, 3, 1, 1, act="lrelu"), L.Act()]
    layers.append(L.Conv2d(32, 2, 3, 1, 1))
    return nn.Sequential(*layers)


def correlate(input1, input2):
    """back2future.py:15-25."""
    return ops.correlate(input1, input2)


class Model(nn.Module):
    FEAT = [3, 16, 32, 64, 96, 128, 192]
    DEC_IN = {6: 162, 5: 292, 4: 260, 3: 228, 2: 196}
    WARP_SCALE = {6: 0.625, 5: 1.25, 4: 2.5, 3: 5.0}
    FULL_SCALE = {2: 20, 3: 10, 4: 5, 5: 2.5, 6: 1.25}

    def __init__(self, nlevels):
        super().__init__()
        self.nlevels = nlevels
        # SURVEY.md 8a "dead work" D1: train.py:463 discards the occlusion output and no gradient reaches decoder_occ*.
        # Default False = identical API (occlusion maps are computed and returned as in the reference); a trainer that
        # does not consume them may set it (bench.py --elide-occ) and gets None in their place.
        self.elide_occ = False
        idx = [k for n in range(80, 71, -1) for k in range(n, -1, -9)]          # back2future.py:56-57
        self.idx_fwd = idx
        self.idx_bwd = list(reversed(idx))
        for lvl in range(1, 7):
            for s in "abc":
                setattr(self, "conv%d%s" % (lvl, s), conv_feat_block(self.FEAT[lvl - 1], self.FEAT[lvl]))
        self.corr = correlate
        # ImageNet statistics of normalize(); non-persistent buffers: not part of the state_dict contract
        self.register_buffer("_im_mean", torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1), persistent=False)
        self.register_buffer("_im_std", torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1), persistent=False)
        for lvl in range(6, 1, -1):
            setattr(self, "decoder_fwd%d" % lvl, conv_dec_block(self.DEC_IN[lvl]))
            setattr(self, "decoder_bwd%d" % lvl, conv_dec_block(self.DEC_IN[lvl]))
        self.decoder_occ6 = conv_dec_block(354)
        for lvl in range(5, 1, -1):
            setattr(self, "decoder_occ%d" % lvl, conv_dec_block(self.DEC_IN[lvl]))

    def init_weights(self):
        """back2future.py:106-116: U(0,1) bias first, then xavier weight."""
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                if m.bias is not None:
                    nn.init.uniform_(m.bias)
                nn.init.xavier_uniform_(m.weight)

    def normalize(self, ims):
        """back2future.py:118-132: [-1,1] -> ImageNet-normalised, on copies."""
        if all(im.shape[1] == 3 and not im.requires_grad for im in ims) and len({tuple(im.shape) for im in ims}) == 1:
            from ..loss_functions import imagenet_normalize_levels
            return imagenet_normalize_levels(ims)
        return [((im * 0.5 + 0.5) - self._im_mean) / self._im_std for im in ims]

    def warp(self, x, flo, flow_scale=1.0):
        """back2future.py:287-321 (flow_scale: the constant the reference multiplies into the flow before the call)."""
        return feature_warp(x, flo, flow_scale=flow_scale)

    def _decoders(self, tape, decs, ins):
        """The 6-layer decoder stacks of one pyramid level (decoder_fwd / decoder_bwd [/ decoder_occ]) layer by layer as
        grouped launches: same-shaped layers of the parallel stacks share one kernel launch per pass; the LeakyReLU backward
        of layers 0-4 is applied by the next layer's data-gradient epilogue (cc_amd/tape.py)."""
        xs = list(ins)
        for i in range(6):
            convs = [d[2 * i] for d in decs]
            act = "lrelu" if i < 5 else None
            # the first layer of decoder_occ6 has its own input width (354 vs 162 channels): group by shape
            shapes = sorted({tuple(c.weight.shape) for c in convs})
            ys = [None] * len(decs)
            for shp in shapes:
                idx = [k for k, c in enumerate(convs) if tuple(c.weight.shape) == shp]
                out = tape.conv_group([xs[k] for k in idx], [convs[k].weight for k in idx], [convs[k].bias for k in idx], 1, 1, act)
                for k, y in zip(idx, out):
                    ys[k] = y
            xs = ys
        return xs

    def _body(self, tape, n0, n1, n2):
        """Forward pass on the tape.  Inputs: the normalised target, I-, I+ frames."""
        ac = config.align_corners
        feats = {}
        xs = [n0, n2, n1]                                          # a = target, b = I+, c = I-  (back2future.py:159,166)
        for lvl in range(1, 7):
            blocks = [getattr(self, "conv%d%s" % (lvl, s)) for s in "abc"]
            ys = tape.conv_group(xs, [b[0].weight for b in blocks], [b[0].bias for b in blocks], 2, 1, "lrelu")
            xs = tape.conv_group(ys, [b[2].weight for b in blocks], [b[2].bias for b in blocks], 1, 1, "lrelu")
            for s, x in zip("abc", xs):
                feats[(lvl, s)] = x
        flow_f, flow_b, up_f, up_b, occ_raw = {}, {}, {}, {}, {}
        bw, cw = feats[(6, "b")], feats[(6, "c")]
        inv_b, inv_c = ops._inv_perm(self.idx_fwd, n0.t.device), ops._inv_perm(self.idx_bwd, n0.t.device)
        for lvl in range(6, 1, -1):
            a = feats[(lvl, "a")]
            B, C, H, W = a.t.shape
            if lvl == 6:
                cb_o = tape.concat(B, [162, C], H, W, a.t) if not self.elide_occ else None
                corr = tape.corr_pair(a, bw, cw, inv_b, inv_c, out=cb_o.slot(0) if cb_o is not None else None)
                in_f = in_b = corr
                if cb_o is not None:
                    cb_o.put(0, corr)
                    cb_o.put(1, a)
                    in_o = cb_o.done()
            else:
                cb_f = tape.concat(B, [162, C, 2], H, W, a.t)
                cb_b = tape.concat(B, [162, C, 2], H, W, a.t)
                corr = tape.corr_pair(a, bw, cw, inv_b, inv_c, out=cb_f.slot(0))
                cb_f.put(0, corr)
                cb_f.put(1, a)
                cb_f.put(2, up_f[lvl + 1])
                cb_b.buf[:, :162 + C].copy_(cb_f.buf[:, :162 + C])        # (corr, a) once more for the backward-flow decoder
                cb_b.parts[0], cb_b.parts[1] = corr, a
                cb_b.put(2, up_b[lvl + 1])
                in_f, in_b = cb_f.done(), cb_b.done()
                in_o = in_f
            decs = [getattr(self, "decoder_fwd%d" % lvl), getattr(self, "decoder_bwd%d" % lvl)]
            ins = [in_f, in_b]
            if not self.elide_occ:
                decs.append(getattr(self, "decoder_occ%d" % lvl))
                ins.append(in_o)
            outs = self._decoders(tape, decs, ins)
            flow_f[lvl], flow_b[lvl] = outs[0], outs[1]
            up_f[lvl] = tape.upsample2x(flow_f[lvl])
            up_b[lvl] = tape.upsample2x(flow_b[lvl])
            if not self.elide_occ:
                occ_raw[lvl] = outs[2]
            if lvl > 2:
                s = self.WARP_SCALE[lvl]
                bw = tape.feature_warp(feats[(lvl - 1, "b")], up_f[lvl], s, ac)
                cw = tape.feature_warp(feats[(lvl - 1, "c")], up_f[lvl], -s, ac)       # the FORWARD flow for both (Q9)
        ff = [tape.upsample2x(up_f[l], self.FULL_SCALE[l]) for l in range(2, 7)]        # scale fused into the up-sampling launch
        fb = [tape.upsample2x(up_b[l], -self.FULL_SCALE[l]) for l in range(2, 7)]
        if self.training and self.nlevels == 6:
            ff += tape.torch_fn([up_f[6]], lambda t: (0.625 * t,))
            fb += tape.torch_fn([up_b[6]], lambda t: (-0.625 * t,))
        if not self.training:
            ff, fb = ff[:1], fb[:1]
        self._n_flow = len(ff)
        return ff + fb + [occ_raw[l] for l in sorted(occ_raw)]

    def forward(self, im_tar, im_refs):
        n = self.normalize([im_tar] + list(im_refs))
        outs = list(run_network(self._body, n, list(self.parameters())))
        k = self._n_flow
        ff, fb, raw = outs[:k], outs[k:2 * k], outs[2 * k:]
        oc = None
        if raw:                       # occlusion maps: softmax + nearest up-sampling (stock torch; train.py:463 discards them)
            occ = dict(zip(range(2, 7), (torch.softmax(t, dim=1) for t in raw)))
            oc = [F.interpolate(occ[l], scale_factor=4) for l in range(2, 7)]
            if self.training and self.nlevels == 6:
                oc.append(F.interpolate(occ[6], scale_factor=2))
        if self.training:
            return ff, fb, oc
        return ff[0], fb[0], (None if oc is None else oc[0])
