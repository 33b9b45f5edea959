"""TEST INFRASTRUCTURE ONLY (oracle) -- CPU restatement of the reference's validation metrics
(loss_functions.py:355-467).  Plain torch fp32, written against the published definitions (KITTI flow EPE / Fl outlier
ratio, Eigen depth metrics with Garg's crop and per-image median scaling), each function citing the lines it follows.
Pinned by tests/golden/metrics.npz, which oracle/make_golden.py writes from the unmodified reference."""
import torch
import torch.nn.functional as F

EPS = 1e-8          # loss_functions.py:11


def _resize(t, size, align_corners):
    return F.interpolate(t, size=size, mode="bilinear", align_corners=align_corners)


def _epe_map(gt, pred, align_corners):
    """:355-365 / :368-377: up-sample the prediction to the ground truth and rescale u by W ratio, v by H ratio."""
    hp, wp = pred.shape[2:]
    hg, wg = gt.shape[2:]
    p = _resize(pred, (hg, wg), align_corners)
    du = gt[:, 0] - p[:, 0] * (wg / wp)
    dv = gt[:, 1] - p[:, 1] * (hg / hp)
    return torch.sqrt(du.pow(2) + dv.pow(2))


def flow_diff(gt, pred, align_corners=False):
    return _epe_map(gt, pred, align_corners)


def compute_epe(gt, pred, align_corners=False):
    """:368-388."""
    e = _epe_map(gt, pred, align_corners)
    if gt.shape[1] == 3:
        v = gt[:, 2]
        return float((e * v).sum() / (v.sum() + EPS))
    return float(e.sum() / (gt.shape[0] * gt.shape[2] * gt.shape[3]))


def outlier_err(gt, pred, tau=(3, 0.05), align_corners=False):
    """:390-409."""
    v = gt[:, 2]
    e = _epe_map(gt, pred, align_corners) * v
    mag = torch.sqrt(gt[:, 0].pow(2) + gt[:, 1].pow(2))
    bad = (e > tau[0]).float() * ((e / (mag + EPS)) > tau[1]).float() * v
    return float(bad.sum() / (v.sum() + EPS))


def compute_all_epes(gt, rigid_pred, non_rigid_pred, rigidity_mask, THRESH=0.5, align_corners=False):
    """:411-429."""
    mp = _resize(rigidity_mask, rigid_pred.shape[2:], align_corners)
    mg = _resize(rigidity_mask, gt.shape[2:], align_corners)
    nr = (mp <= THRESH).float() * non_rigid_pred
    rg = (mp > THRESH).float() * rigid_pred
    tot = nr + rg
    gt_nr = (mg <= THRESH).float() * gt
    gt_rg = (mg > THRESH).float() * gt
    return [compute_epe(gt, tot, align_corners), compute_epe(gt_rg, rg, align_corners),
            compute_epe(gt_nr, nr, align_corners), outlier_err(gt, tot, align_corners=align_corners)]


def compute_errors(gt, pred, crop=True):
    """:432-467 -> six floats."""
    acc = [0.0] * 6
    B, H, W = gt.shape
    box = torch.zeros(H, W, dtype=torch.bool)
    if crop:
        box[int(0.40810811 * H):int(0.99189189 * H), int(0.03594771 * W):int(0.96405229 * W)] = True
    for g, p in zip(gt, pred):
        ok = (g > 0) & (g < 80)
        if crop:
            ok = ok & box
        vg = g[ok]
        vp = p[ok].clamp(1e-3, 80)
        vp = vp * torch.median(vg) / torch.median(vp)
        th = torch.max(vg / vp, vp / vg)
        vals = [(vg - vp).abs().mean(), ((vg - vp).abs() / vg).mean(), (((vg - vp) ** 2) / vg).mean(),
                (th < 1.25).float().mean(), (th < 1.25 ** 2).float().mean(), (th < 1.25 ** 3).float().mean()]
        acc = [a + float(v) for a, v in zip(acc, vals)]
    return [a / B for a in acc]
