"""Where does the difference between the engine's and the reference's parameter gradients arise?  

Given the reference's own d loss / d (network outputs) of one step (oracle.step.cc_step_keep on the CPU: the gradients its loss
path sends into DispResNet6 / PoseNetB6 / MaskNet6 / Back2Future) and its parameter gradients:
  (i)  feed the REFERENCE's output gradients into the ENGINE's four network backward passes (same weights, same batch) and compare
       the parameter gradients per network -- what the networks' backward kernels contribute on their own;
  (ii) compare the ENGINE's d loss / d (network outputs) (its loss path: warps, SSIM, masks, smoothness, consensus) with the
       reference's, per output group: whole-tensor L2, the number of elements off by more than 1e-4 of the largest, and the L2 of
       the rest -- isolated elements are bilinear-tap / comparison decisions that fall the other way within rounding
       (tests/parity.py _flip_pinned), a defect would spread.
bench.py calls run() after its timed region (parity.gradient_pin on the JSON line); stand-alone:
    python tools/grad_pin.py            (runs the CPU reference step itself: ~1 min on the GPU box's host cores)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

GROUPS = (("disparities", 6), ("pose", 1), ("exp_mask", 6), ("flow_fwd", 6), ("flow_bwd", 6))


def _l2(a, b):
    return float(torch.sqrt(((a.double() - b.double()) ** 2).sum()) / torch.sqrt((b.double() ** 2).sum()).clamp_min(1e-300))


def run(init_sd, batch_cpu, ref_out_grads, ref_param_grads, dev, config="c3", truth64=None, cond=None):
    """-> dict (JSON-able).  init_sd: the four state dicts both sides started from; ref_param_grads: flat tensor in the chain order
    of the trainable parameters (as bench.py's cpu_first['grads'])."""
    from cc_amd import trainer as T, ops
    from cc_amd import loss_functions as LF
    full = config == "c3"
    nets = T.build_nets(dev, flow=full, mask=full)
    for n_, sd in zip(nets, init_sd):
        if n_ is not None:
            n_.load_state_dict(sd)
            n_.train()
    batch = (batch_cpu[0].to(dev), [r.to(dev) for r in batch_cpu[1]], batch_cpu[2].to(dev), batch_cpu[3].to(dev))
    cfg = T.StepConfig()
    ops.packs.reset()
    LF.pyramid_cache.clear()
    cut = {}
    out = T.cc_forward(nets, batch, cfg, cut=cut)
    pairs = cut.get("dp", []) + cut.get("mf", [])
    ref = [g_ for g_ in ref_out_grads]
    groups = [(nm, k) for nm, k in GROUPS if full or nm in ("disparities", "pose")]
    assert len(pairs) == sum(k for _, k in groups) == len(ref), (len(pairs), len(ref))
    eng = torch.autograd.grad(out["loss"], [d for _, d in pairs], allow_unused=True)
    res = {"head_gradients": {}, "note": "(i) = by_net_given_reference_output_gradients; (ii) = head_gradients"}
    # (ii) the loss path's output gradients
    o = 0
    for nm, k in groups:
        ge = [eng[o + i] for i in range(k)]
        gr = [ref[o + i] for i in range(k)]
        o += k
        ge = torch.cat([(a if a is not None else torch.zeros_like(b.to(dev))).reshape(-1).cpu() for a, b in zip(ge, gr) if b is not None])
        gr = torch.cat([b.reshape(-1) for b in gr if b is not None])
        d = (ge - gr).abs()
        off = d > 1e-4 * gr.abs().max()
        rest = _l2(torch.where(off, gr, ge), gr)
        res["head_gradients"][nm] = {"l2_rel": float("%.3e" % _l2(ge, gr)), "elements": int(gr.numel()), "off_1e-4_of_max": int(off.sum()),
                                     "l2_rel_without_them": float("%.3e" % rest), "max_abs_ref": float("%.3e" % float(gr.abs().max()))}
    # (i) the networks' backward passes, driven by the reference's output gradients
    for n_ in nets:
        if n_ is not None:
            for p_ in n_.parameters():
                p_.grad = None
    ts = [t for (t, _), g_ in zip(pairs, ref) if g_ is not None]
    gs = [g_.to(dev) for g_ in ref if g_ is not None]
    torch.autograd.backward(ts, gs)
    torch.cuda.synchronize()
    by_net, o = [], 0
    names = ["DispResNet6", "PoseNetB6", "MaskNet6", "Back2Future"]
    for n_, nm in zip(nets, names):
        if n_ is None:
            continue
        ps = [p_ for p_ in n_.parameters() if p_.requires_grad]
        k = sum(p_.numel() for p_ in ps)
        ge = torch.cat([(p_.grad if p_.grad is not None else torch.zeros_like(p_)).reshape(-1) for p_ in ps]).cpu()
        gr = ref_param_grads[o:o + k]
        o += k
        d2 = (ge.double() - gr.double()) ** 2
        ent = {"net": nm, "l2_rel": float("%.3e" % _l2(ge, gr)),
               "top1000_share": float("%.3f" % float(torch.topk(d2, min(1000, d2.numel())).values.sum() / d2.sum().clamp_min(1e-300)))}
        if truth64 is not None:       # both fp32 sides against the float64 gradient of the same weights and output gradients
            g64 = truth64[o - k:o]
            ent["engine_vs_fp64"] = float("%.3e" % _l2(ge.double(), g64))
            ent["reference_vs_fp64"] = float("%.3e" % _l2(gr.double(), g64))
            # per parameter tensor: where the reference's own fp32 rounding sits (largest five)
            per, q = [], 0
            for pn, p_ in [(a, b) for a, b in n_.named_parameters() if b.requires_grad]:
                m_ = p_.numel()
                per.append((pn, _l2(ge[q:q + m_], gr[q:q + m_]), _l2(ge[q:q + m_].double(), g64[q:q + m_]), _l2(gr[q:q + m_].double(), g64[q:q + m_])))
                q += m_
            per.sort(key=lambda r_: -r_[1])
            ent["worst_tensors (name, engine vs reference, engine vs fp64, reference vs fp64)"] = [
                [a, float("%.2e" % b), float("%.2e" % c), float("%.2e" % e)] for a, b, c, e in per[:(200 if os.environ.get("GRAD_PIN_ALL") else 5)]]
        by_net.append(ent)
    res["by_net_given_reference_output_gradients"] = by_net
    if cond is not None:
        for ent, c_ in zip(by_net, cond):
            ent["reference_moves_by_under_1e-7_input_noise"] = c_
    ops.packs.reset()
    return res


def _net_call(k, m, tgt, refs):
    o = m(tgt) if k == 0 else (m(tgt, refs) if k in (1, 2) else m(tgt, refs[1:3]))
    if k == 0:
        return list(o), slice(0, 6)
    if k == 1:
        return [o], slice(6, 7)
    if k == 2:
        return list(o), slice(7, 13)
    return list(o[0]) + list(o[1]), slice(13, 25)


def fp64_param_grads(init_sd, batch, og):
    """The four network backward passes in float64 (oracle modules = the reference's arithmetic, pinned bit-exact in fp32 by
    tests/golden), driven by the given output gradients: the exact parameter gradient of these weights, to which BOTH fp32 sides are
    compared.  -> flat float64 tensor in the chain order of the trainable parameters."""
    from oracle import step as S
    torch.set_default_dtype(torch.float64)
    try:
        dn = S.build_nets("oracle", flow=init_sd[3] is not None, mask=init_sd[2] is not None)
    finally:
        torch.set_default_dtype(torch.float32)
    tgt, refs = batch[0].double(), [r.double() for r in batch[1]]
    parts = []
    for k, (m, sd) in enumerate(zip(dn, init_sd)):
        if m is None:
            continue
        m.double()
        m.load_state_dict({a: (b.double() if b.is_floating_point() else b) for a, b in sd.items()})
        m.train()
        outs, sl = _net_call(k, m, tgt, refs)
        pr = [(t, g_.double()) for t, g_ in zip(outs, og[sl]) if g_ is not None]
        torch.autograd.backward([t for t, _ in pr], [g_ for _, g_ in pr])
        parts.append(torch.cat([(p_.grad if p_.grad is not None else torch.zeros_like(p_)).detach().reshape(-1)
                                for p_ in m.parameters() if p_.requires_grad]))
    return torch.cat(parts)


def conditioning(init_sd, batch, og, eps=1e-7):
    """How far does the REFERENCE's own fp32 parameter gradient of each network move when the input frames are perturbed by `eps`
    relative (one unit in the last place of an fp32 image)?  The amplification is the condition number of that gradient: two correct
    fp32 evaluations with different summation orders cannot be expected to agree better than this.  (DispResNet6: BatchNorm batch
    statistics over 16 - 208 values per channel at its deep levels; measured 2e-3 at B = 4, 256 x 832.)  -> [per-network relative L2]"""
    from oracle import step as S
    nets = S.build_nets("oracle", flow=init_sd[3] is not None, mask=init_sd[2] is not None)
    g = torch.Generator().manual_seed(11)
    res = []
    for k, (m, sd) in enumerate(zip(nets, init_sd)):
        if m is None:
            continue
        m.load_state_dict(sd)
        m.train()
        flat = []
        for rep in range(2):
            for p_ in m.parameters():
                p_.grad = None
            tgt = batch[0] if rep == 0 else batch[0] * (1 + eps * torch.randn(batch[0].shape, generator=g))
            refs = batch[1] if rep == 0 else [r * (1 + eps * torch.randn(r.shape, generator=g)) for r in batch[1]]
            outs, sl = _net_call(k, m, tgt, refs)
            pr = [(t, g_) for t, g_ in zip(outs, og[sl]) if g_ is not None]
            torch.autograd.backward([t for t, _ in pr], [g_ for _, g_ in pr])
            flat.append(torch.cat([(p_.grad if p_.grad is not None else torch.zeros_like(p_)).detach().reshape(-1)
                                   for p_ in m.parameters() if p_.requires_grad]))
        res.append(float("%.3e" % _l2(flat[1], flat[0])))
    return res


BOUNDARIES = ("conv4.1", "conv5.0", "conv5.1", "conv6.0", "conv6.1", "conv7.0", "conv7.1")


def _oracle_boundaries(init_sd, batch, og, dtype):
    """DispResNet6 (oracle module = the reference's arithmetic) forward + backward in `dtype`, driven by the output gradients og[0:6]:
    -> {boundary: (block output Y, d loss / d Y)} at the BasicBlock outputs of BOUNDARIES, and the parameter gradients by name."""
    from oracle import step as S
    torch.set_default_dtype(dtype)
    try:
        m = S.build_nets("oracle", flow=False, mask=False)[0]
    finally:
        torch.set_default_dtype(torch.float32)
    m.to(dtype)
    m.load_state_dict({a: (b.to(dtype) if b.is_floating_point() else b) for a, b in init_sd[0].items()})
    m.train()
    keep = {}

    def hook(name):
        def f(mod, inp, out):
            out.retain_grad()
            keep[name] = out
        return f
    hs = []
    pre = {}

    def hook_pre(name):
        # the block's PRE-activation: conv2(...) + shortcut (models/DispResNet6.py:31-43) -- the number whose sign is the ReLU decision
        def f(mod, inp, out):
            pre[name] = (pre.get(name + "/res"), out.detach())
        return f

    def hook_in(name, blk):
        def f(mod, inp):
            x = inp[0].detach()
            pre[name + "/res"] = x if blk.downsample is None else blk.downsample(inp[0]).detach()
        return f
    for name in BOUNDARIES:
        stage, blk = name.split(".")
        b_ = getattr(m, stage)[int(blk)]
        hs.append(b_.register_forward_hook(hook(name)))
        hs.append(b_.register_forward_pre_hook(hook_in(name, b_)))
        hs.append(b_.conv2.register_forward_hook(hook_pre(name)))
    outs = list(m(batch[0].to(dtype)))
    pr = [(t, g_.to(dtype)) for t, g_ in zip(outs, og[0:6]) if g_ is not None]
    torch.autograd.backward([t for t, _ in pr], [g_ for _, g_ in pr])
    for h in hs:
        h.remove()
    return ({k: (v.detach(), v.grad.detach(), (pre[k][1] + pre[k][0]) if (k in pre and pre[k][0] is not None) else None)
             for k, v in keep.items()},
            {k: p_.grad.detach() for k, p_ in m.named_parameters() if p_.grad is not None})


def boundaries(init_sd, batch_cpu, og, dev):
    """WHERE inside DispResNet6's backward does the engine's gradient leave the float64 one?  At the outputs Y of
    the BasicBlocks conv4.1 ... conv7.1: the gradient w.r.t. the block's PRE-activation, dL/dZ = dL/dY * 1[Y > 0] (what the tape
    hands the block's last convolution), engine / reference-fp32 against float64, and how many ReLU decisions 1[Y > 0] each fp32
    side takes differently from float64 (a flipped decision switches a whole gradient path on or off)."""
    from cc_amd import trainer as T, ops, tape
    t64, p64 = _oracle_boundaries(init_sd, batch_cpu, og, torch.float64)
    t32, p32 = _oracle_boundaries(init_sd, batch_cpu, og, torch.float32)
    nets = T.build_nets(dev, flow=False, mask=False)
    net = nets[0]
    net.load_state_dict(init_sd[0])
    net.train()
    ops.packs.reset()
    tape.TAPS = {}
    try:
        outs = list(net(batch_cpu[0].to(dev)))
        for p_ in net.parameters():
            p_.grad = None
        pr = [(t, g_.to(dev)) for t, g_ in zip(outs, og[0:6]) if g_ is not None]
        torch.autograd.backward([t for t, _ in pr], [g_ for _, g_ in pr])
        torch.cuda.synchronize()
        taps = dict(tape.TAPS)
    finally:
        tape.TAPS = None
    # the engine's block outputs: a second forward pass with hooks on the module outputs is not available on the tape, so the
    # ReLU decisions of the engine are read off its gradient (dL/dZ is exactly 0 where the engine took Y <= 0) together with the
    # float64 activations: an element counts as an engine flip when float64 says active (Y64 > 0, dL/dZ64 != 0) and the engine's
    # dL/dZ is exactly 0, or the other way round
    rows = []
    for name in BOUNDARIES:
        if name not in taps or taps[name][0] is None:
            continue
        g_e, pre = taps[name]
        g_e = g_e.cpu().double()
        y64, gy64, zpre64 = t64[name]
        y32, gy32, zpre32 = t32[name]
        z64 = gy64 * (y64 > 0)
        z32 = (gy32 * (y32 > 0)).double()
        a64 = y64 > 0
        if not pre:                     # (the tape had not applied relu' yet: apply the float64 decision, flips then show as 0)
            g_e = g_e * a64
        flips_ref = int(((y32 > 0) != a64).sum())
        act_e = g_e != 0
        flips_eng = int(((act_e != (z64 != 0)) & (gy64 != 0)).sum()) if pre else 0
        # the part of each side's error that sits on elements where the ReLU decisions agree with float64
        agree_e = (act_e == (z64 != 0))
        agree_r = ((y32 > 0) == a64)
        flip_detail = []
        if pre and flips_eng:
            idx = ((act_e != (z64 != 0)) & (gy64 != 0)).nonzero()
            e2 = float(((g_e - z64) ** 2).sum())
            for ix in idx[:4]:
                ix = tuple(int(v) for v in ix)
                flip_detail.append({"index": list(ix), "Y_fp64": float("%.3e" % float(y64[ix])), "Y_ref_fp32": float("%.3e" % float(y32[ix])),
                                    "preactivation_fp64": float("%.3e" % float(zpre64[ix])) if zpre64 is not None else None,
                                    "preactivation_ref_fp32": float("%.3e" % float(zpre32[ix])) if zpre32 is not None else None,
                                    "elements_with_abs_preactivation_fp64_below_2e-6": int((zpre64.abs() < 2e-6).sum()) if zpre64 is not None else None,
                                    "dLdY_fp64": float("%.3e" % float(gy64[ix])), "engine_dLdZ": float("%.3e" % float(g_e[ix])),
                                    "rms_dLdZ_fp64": float("%.3e" % float(z64.pow(2).mean().sqrt())),
                                    "max_abs_Y_fp64": float("%.3e" % float(y64.abs().max())),
                                    "share_of_squared_error": float("%.3f" % (float((g_e[ix] - z64[ix]) ** 2) / max(e2, 1e-300)))})
        rows.append({"boundary": name, "shape": list(y64.shape), "pre_applied_by_tape": bool(pre), "flips": flip_detail,
                     "engine_vs_fp64": float("%.3e" % _l2(g_e, z64)), "reference_vs_fp64": float("%.3e" % _l2(z32, z64)),
                     "relu_flips_engine": flips_eng, "relu_flips_reference": flips_ref,
                     "engine_vs_fp64_where_relu_agrees": float("%.3e" % _l2(torch.where(agree_e, g_e, z64), z64)),
                     "reference_vs_fp64_where_relu_agrees": float("%.3e" % _l2(torch.where(agree_r, z32, z64), z64))})
    # the parameter gradients of the same run, per tensor of conv5 / conv6
    par = []
    ge = {k: p_.grad.detach().cpu().double() for k, p_ in net.named_parameters() if p_.grad is not None}
    for k in ("conv4.1.conv2.weight", "conv5.0.conv1.weight", "conv5.0.conv2.weight", "conv5.1.conv1.weight", "conv5.1.conv2.weight",
              "conv6.0.conv1.weight", "conv6.0.conv2.weight", "conv6.1.conv1.weight", "conv6.1.conv2.weight"):
        if k in ge and k in p64:
            par.append([k, float("%.2e" % _l2(ge[k], p64[k])), float("%.2e" % _l2(p32[k].double(), p64[k]))])
    ops.packs.reset()
    return {"boundaries": rows, "weights (name, engine vs fp64, reference vs fp64)": par}


def cpu_child(inp, outp):
    """the reference (or, without oracle/_ref, the oracle) step on the host cores, in a process that sees no GPU"""
    from oracle import step as S, ref_import
    d = torch.load(inp)
    impl = ref_import.load() if ref_import.reference_available() else None
    rn = S.build_nets("ref", impl) if impl is not None else S.build_nets("oracle")
    for m, sd in zip(rn, d["init_sd"]):
        m.load_state_dict(sd)
        m.train()
    cfg = S.StepConfig()
    opt = S.make_optimizer(rn, cfg)
    _, og = S.cc_step_keep(rn, opt, d["batch"], cfg, impl)
    pg = torch.cat([(p_.grad if p_.grad is not None else torch.zeros_like(p_)).detach().reshape(-1)
                    for m in rn for p_ in m.parameters() if p_.requires_grad])
    pg64 = fp64_param_grads(d["init_sd"], d["batch"], og) if d.get("fp64", True) else None
    cond = conditioning(d["init_sd"], d["batch"], og)
    torch.save({"og": og, "pg": pg, "pg64": pg64, "cond": cond, "kind": "reference" if impl is not None else "port"}, outp)


def main():
    import json
    import subprocess
    import tempfile
    if len(sys.argv) == 4 and sys.argv[1] == "--cpu-child":
        return cpu_child(sys.argv[2], sys.argv[3])
    from cc_amd import synthetic as syn, trainer as T
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    nets = T.build_nets(dev)
    init_sd = [{k: v.detach().cpu().clone() for k, v in n_.state_dict().items()} for n_ in nets]
    del nets
    batch_cpu = syn.sample(4, 256, 832, seed=1, smooth=3)
    with tempfile.TemporaryDirectory() as td:
        inp, outp = os.path.join(td, "in.pt"), os.path.join(td, "out.pt")
        torch.save({"batch": batch_cpu, "init_sd": init_sd}, inp)
        env = dict(os.environ)
        env["HIP_VISIBLE_DEVICES"] = ""
        subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-child", inp, outp], env=env, check=True)
        r = torch.load(outp)
    res = run(init_sd, batch_cpu, r["og"], r["pg"], dev, truth64=r.get("pg64"), cond=r.get("cond"))
    res["cpu_side"] = r["kind"]
    print(json.dumps(res, indent=1))
    if os.environ.get("GRAD_PIN_BOUNDARIES", "1") == "1":
        torch.set_num_threads(min(32, os.cpu_count() or 8))
        b = boundaries(init_sd, batch_cpu, r["og"], dev)
        print("== DispResNet6 block boundaries: d loss / d (pre-activation) against float64 (reference's output gradients on all sides)")
        for row in b["boundaries"]:
            print(json.dumps(row))
        for row in b["weights (name, engine vs fp64, reference vs fp64)"]:
            print("  %-24s engine vs fp64 %.2e   reference vs fp64 %.2e" % tuple(row))
    if os.environ.get("GRAD_PIN_VARIANTS"):
        # the same comparison under kernel-selection switches of the TOOLS build: "VAR=VAL,VAR=VAL;VAR=VAL;..." -- does the figure
        # move with the algorithm a layer runs on?
        from cc_amd import _lib, build
        with _lib.use_library(build.TOOLS_OUT):
            for var in ["default"] + os.environ["GRAD_PIN_VARIANTS"].split(";"):
                kv = dict(x.split("=") for x in var.split(",")) if var != "default" else {}
                os.environ.update(kv)
                rv = run(init_sd, batch_cpu, r["og"], r["pg"], dev, truth64=r.get("pg64"), cond=r.get("cond"))
                for k in kv:
                    os.environ.pop(k)
                e = rv["by_net_given_reference_output_gradients"][0]
                rows = e[[k for k in e if k.startswith("worst")][0]]
                print("variant %-40s DispResNet6 engine vs fp64 %.3e  (vs reference %.3e)  worst %s" % (var, e["engine_vs_fp64"], e["l2_rel"], rows[:3]))


if __name__ == "__main__":
    main()
