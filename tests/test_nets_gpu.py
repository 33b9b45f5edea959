"""Networks and the full CC step on the MI355X against (a) the oracle on CPU and (b) the golden fixture the
UNMODIFIED reference produced (tests/golden/step_acF.npz)."""
import os

import numpy as np
import pytest
import torch

from cc_amd import models, synthetic as syn, trainer as T
from oracle import nets as N, step as S
from oracle.make_golden import SB, SH, SW

pytestmark = pytest.mark.gpu


def _flat(o):
    if torch.is_tensor(o):
        return [o]
    r = []
    for x in o:
        if x is not None:
            r += _flat(x)
    return r


def _rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize("name", ["disp", "pose", "mask", "flow", "dispS", "poseexp"])
def test_net_forward_backward(name):
    dev = "cuda"
    tgt, refs, K, Kinv = syn.sample(2, 128, 192, seed=1)
    mk = {"disp": (models.DispResNet6, N.DispResNet6, (), "t"), "pose": (models.PoseNetB6, N.PoseNetB6, (4,), "tr"),
          "mask": (models.MaskNet6, N.MaskNet6, (4,), "tr"), "flow": (models.Back2Future, N.Back2Future, (6,), "t2"),
          "dispS": (models.DispNetS, N.DispNetS, (), "t"), "poseexp": (models.PoseExpNet, N.PoseExpNet, (4, True), "tr")}[name]
    mine, orc = mk[0](*mk[2]), mk[1](*mk[2])
    assert list(mine.state_dict().keys()) == list(orc.state_dict().keys())
    sd = syn.seeded_state_dict(orc, 0)
    mine.load_state_dict(sd)
    orc.load_state_dict(sd)
    mine.to(dev)
    a_cpu = {"t": (tgt,), "tr": (tgt, refs), "t2": (tgt, refs[1:3])}[mk[3]]
    a_dev = {"t": (tgt.to(dev),), "tr": (tgt.to(dev), [r.to(dev) for r in refs]),
             "t2": (tgt.to(dev), [r.to(dev) for r in refs[1:3]])}[mk[3]]
    o1, o0 = _flat(mine(*a_dev)), _flat(orc(*a_cpu))
    for a, b in zip(o1, o0):
        assert _rel(a, b) < 1e-4
    sum((a * a).sum() for a in o1).backward()
    sum((a * a).sum() for a in o0).backward()
    num = den = 0.0
    for (n1, p1), (n2, p2) in zip(mine.named_parameters(), orc.named_parameters()):
        if p2.grad is None:
            continue
        num += float((p1.grad.cpu().double() - p2.grad.double()).pow(2).sum())
        den += float(p2.grad.double().pow(2).sum())
    # fp32 accumulation-order noise through ~60 layers + BatchNorm over as few as 4 values (the oracle runs on this
    # host's CPU, whose conv kernels round differently from the build container's): 1e-4 of the gradient norm (measured on
    # the emulator: 2.4e-6; a mis-accumulated fan-out in the tape shows up at 1e-2 and above)
    print(name, "parameter-gradient error / norm:", (num / den) ** 0.5)
    assert (num / den) ** 0.5 < 1e-4


@pytest.mark.parametrize("use_graph", [False, True])
def test_full_step_against_reference_golden(golden_dir, use_graph):
    """losses within 1e-4 rel of the reference CPU path (BASELINE.json north star), one Adam step included."""
    g = dict(np.load(os.path.join(golden_dir, "step_acF.npz")))
    dev = torch.device("cuda")
    bc = syn.sample(SB, SH, SW, seed=1)
    batch = (bc[0].to(dev), [r.to(dev) for r in bc[1]], bc[2].to(dev), bc[3].to(dev))
    nets = T.build_nets(dev, init=False)
    for n in nets:
        n.load_state_dict(syn.seeded_state_dict(n, 0))
    tr = T.CCTrainer(nets, T.StepConfig(), use_graph=use_graph)
    got = {k: float(v) for k, v in tr.step(batch).items()}
    for k in ("loss", "loss_1", "loss_2", "loss_3", "loss_4", "loss_5"):
        assert abs(got[k] - float(g[k])) <= 1e-4 * abs(float(g[k])), (k, got[k], float(g[k]))
    for name, v in tr.grad_norms().items():            # per-network gradient norms of the reference's loss.backward()
        want = float(g["gradnorm." + name])
        assert abs(float(v) - want) <= 1e-3 * want, (name, float(v), want)
    # the small-parameter gradients the fixture stores in full (biases of <= 64 channels)
    worst, offs = (0.0, ""), iter(tr.opt.offsets)          # (every network's range of the bucket starts on a 256-byte boundary)
    for name, n in zip(("disp", "pose", "mask", "flow"), nets):
        for pn, p in n.named_parameters():
            k, off = p.numel(), next(offs)
            key = "grad.%s.%s" % (name, pn)
            if key in g:
                gw = torch.from_numpy(g[key]).double().reshape(-1)
                gg = tr.opt.flat_g[off:off + k].detach().cpu().double()
                r = float((gg - gw).norm()) / (float(gw.norm()) + 1e-12)
                worst = max(worst, (r, key))
                assert r <= 1e-2, (key, r)
    print("worst small-parameter gradient mismatch vs the reference: %.2e (%s)" % worst)
    got2 = {k: float(v) for k, v in tr.step(batch).items()}
    assert abs(got2["loss"] - float(g["loss_after_adam"])) <= 1e-4 * abs(float(g["loss_after_adam"]))


def test_config2_against_reference_golden(golden_dir):
    g = dict(np.load(os.path.join(golden_dir, "step_acF.npz")))
    dev = torch.device("cuda")
    bc = syn.sample(SB, SH, SW, seed=1)
    batch = (bc[0].to(dev), [r.to(dev) for r in bc[1]], bc[2].to(dev), bc[3].to(dev))
    nets = T.build_nets(dev, flow=False, mask=False, init=False)
    for n in nets:
        if n is not None:
            n.load_state_dict(syn.seeded_state_dict(n, 0))
            n.train()
    out = T.cc_forward(nets, batch, T.StepConfig())
    assert abs(float(out["loss"]) - float(g["c2.loss"])) <= 1e-4 * abs(float(g["c2.loss"]))


def test_training_step_reaches_no_vendor_conv_or_batchnorm_kernel():
    """One eager CC step under the profiler: every convolution / BatchNorm runs on libccengine.so -- no MIOpen kernel
    (the round-1 step still dispatched 9 of 13 BatchNorms to MIOpen) and no ATen convolution / batch-norm kernel."""
    from torch.profiler import profile, ProfilerActivity
    dev = torch.device("cuda")
    bc = syn.sample(2, 64, 128, seed=1)
    batch = (bc[0].to(dev), [r.to(dev) for r in bc[1]], bc[2].to(dev), bc[3].to(dev))
    nets = T.build_nets(dev, init=False)
    for n in nets:
        n.load_state_dict(syn.seeded_state_dict(n, 0))
    tr = T.CCTrainer(nets, T.StepConfig(), use_graph=False)
    tr.step(batch)
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        tr.step(batch)
        torch.cuda.synchronize()
    names = [e.key for e in prof.key_averages()]
    kernels = [n for n in names if n.startswith("k_") or "Kernel" in n or "kernel" in n or "MIOpen" in n or "miopen" in n]
    if not any(n.startswith("k_") or "k_conv" in n for n in names):
        pytest.skip("the profiler returned no device kernel names on this box")
    bad = [n for n in names if "miopen" in n.lower() or "batch_norm" in n.lower() or "cudnn" in n.lower()
           or n in ("aten::convolution", "aten::conv2d", "aten::miopen_convolution", "aten::native_batch_norm")]
    assert not bad, bad
    print("device kernels seen: %d distinct, e.g. %s" % (len(kernels), kernels[:6]))


@pytest.mark.parametrize("deterministic", [False, True])
def test_step_forms_agree(deterministic, monkeypatch):
    """The three arrangements of one training step -- "per_network" (round 6, the default: every network's exchange / Adam segment /
    weight images at the end of ITS backward pass, inside the one hipGraph), "post" (one graph, optimizer behind it) and "staged"
    (two hipGraphs sharing a pool, the first segment's all-reduce between the replays) -- on one GPU: same losses, gradients and
    parameters over two steps -- bit for bit with config.deterministic (the feature-warp scatter as integer atomics), to 1e-4 of
    the gradient norm with float atomics."""
    from cc_amd import config
    monkeypatch.setattr(config, "deterministic", deterministic)
    dev = torch.device("cuda")
    bc = syn.sample(2, 128, 192, seed=1)
    batch = (bc[0].to(dev), [r.to(dev) for r in bc[1]], bc[2].to(dev), bc[3].to(dev))
    res = []
    for form in ("post", "staged", "per_network"):
        nets = T.build_nets(dev, init=False)
        for n in nets:
            n.load_state_dict(syn.seeded_state_dict(n, 0))
        tr = T.CCTrainer(nets, T.StepConfig(), use_graph=True, pipeline=form)
        l1 = {k: float(v) for k, v in tr.step(batch).items()}
        g1 = tr.opt.flat_g.clone()
        l2 = {k: float(v) for k, v in tr.step(batch).items()}
        l3 = {k: float(v) for k, v in tr.step(batch).items()}
        assert (tr.graph_b is not None) == (form == "staged") and tr.pipeline == form
        assert float(tr.opt.step_dev) == 3.0
        res.append((l1, g1, l2, l3, tr.opt.flat_p.clone()))
    a1, ga, a2, a3, pa = res[0]
    for b1, gb, b2, b3, pb in res[1:]:
        for k in a1:
            assert abs(a1[k] - b1[k]) <= 1e-6 * abs(a1[k]) + 1e-9, (k, a1[k], b1[k])
            assert abs(a2[k] - b2[k]) <= 1e-5 * abs(a2[k]) + 1e-9, (k, a2[k], b2[k])
            assert abs(a3[k] - b3[k]) <= 2e-5 * abs(a3[k]) + 1e-9, (k, a3[k], b3[k])
        if deterministic:
            assert a1 == b1 and a2 == b2 and a3 == b3 and torch.equal(ga, gb) and torch.equal(pa, pb)
            continue
        # feature-warp backward scatters with float atomics (like the reference's grid_sample): not bit-reproducible
        assert float((ga - gb).norm() / ga.norm()) < 1e-4
        assert float((pa - pb).abs().max()) <= 3 * 2.001e-4


@pytest.mark.parametrize("use_graph", [False, True])
def test_loss_phase_on_two_streams_changes_no_bit(use_graph, monkeypatch):
    """config.loss_stream (round 6): the consensus target, the mask / smoothness terms and the flow photometric loss issue their forward
    launches on Back2Future's stream beside the rigid photometric loss; the terms are CALLED in the same order and their backward
    calls stay on the step's stream, so the shared gradient accumulators see the same sequence of '=' / '+=': with the scatter kernels
    in their deterministic form every loss, gradient and updated parameter is bit-identical to the one-stream phase -- eagerly (the
    caching allocator's cross-stream hand-overs) and as a captured graph."""
    from cc_amd import config
    monkeypatch.setattr(config, "deterministic", True)
    dev = torch.device("cuda")
    bc = syn.sample(2, 128, 192, seed=1)
    batch = (bc[0].to(dev), [r.to(dev) for r in bc[1]], bc[2].to(dev), bc[3].to(dev))
    res = []
    for on in (False, True):
        monkeypatch.setattr(config, "loss_stream", on)
        nets = T.build_nets(dev, init=False)
        for n in nets:
            n.load_state_dict(syn.seeded_state_dict(n, 0))
        tr = T.CCTrainer(nets, T.StepConfig(), use_graph=use_graph)
        assert tr.net_streams, "the networks' side streams are the shipped configuration on HIP devices"
        ls = [{k: float(v) for k, v in tr.step(batch).items()} for _ in range(3)]
        res.append((ls, tr.opt.flat_g.clone(), tr.opt.flat_p.clone()))
    (la, ga, pa), (lb, gb, pb) = res
    assert la == lb
    assert torch.equal(ga, gb) and torch.equal(pa, pb)


@pytest.mark.parametrize("use_graph", [False, True])
def test_weight_images_left_by_the_pipelined_step_equal_a_full_rebuild(use_graph):
    """Per-network form: every network's weight images are rebuilt right behind ITS Adam segment, on its own stream, and the next
    step starts from them without a start-of-step refresh.  After two steps every image must be byte-identical to what one full
    rebuild from the current weights gives (a stale or half-written image would poison every later step silently)."""
    from cc_amd import ops
    dev = torch.device("cuda")
    bc = syn.sample(2, 128, 192, seed=1)
    batch = (bc[0].to(dev), [r.to(dev) for r in bc[1]], bc[2].to(dev), bc[3].to(dev))
    nets = T.build_nets(dev, init=False)
    for n in nets:
        n.load_state_dict(syn.seeded_state_dict(n, 0))
    tr = T.CCTrainer(nets, T.StepConfig(), use_graph=use_graph)
    assert tr.pipeline == "per_network"
    tr.step(batch)
    tr.step(batch)
    torch.cuda.synchronize()
    ents = [e for e in ops.packs.entries.values() if e]
    assert len(ents) > 300 and all(e["ok"] for e in ents)
    left = [e["buf"].clone() for e in ents]
    ops.packs.mark_stale()
    ops.packs.prepack_all()
    torch.cuda.synchronize()
    bad = [i for i, (a, e) in enumerate(zip(left, ents)) if not torch.equal(a, e["buf"])]
    ops.packs.end_step()
    assert not bad, (len(bad), len(ents))


def test_pipelined_step_sees_weights_changed_from_outside():
    """The per-network form keeps the weight images of the previous step's refresh; a torch in-place write to the parameters between
    two steps (load_state_dict, a manual edit) must be picked up -- under hipGraph replay too, where the captured step contains no
    start-of-step refresh."""
    dev = torch.device("cuda")
    bc = syn.sample(2, 128, 192, seed=1)
    batch = (bc[0].to(dev), [r.to(dev) for r in bc[1]], bc[2].to(dev), bc[3].to(dev))
    out = []
    for use_graph in (False, True):
        nets = T.build_nets(dev, init=False)
        for n in nets:
            n.load_state_dict(syn.seeded_state_dict(n, 0))
        tr = T.CCTrainer(nets, T.StepConfig(), use_graph=use_graph)
        assert tr.pipeline == "per_network"
        la = float(tr.step(batch)["loss"])
        for n in nets:
            n.load_state_dict(syn.seeded_state_dict(n, 5))          # other weights, written in place into the bucket
        lb = float(tr.step(batch)["loss"])
        nets2 = T.build_nets(dev, init=False)
        for n in nets2:
            n.load_state_dict(syn.seeded_state_dict(n, 5))
        tr2 = T.CCTrainer(nets2, T.StepConfig(), use_graph=False, pipeline="post")
        want = float(tr2.step(batch)["loss"])
        assert abs(lb - want) <= 1e-5 * abs(want), (use_graph, lb, want, la)
        out.append(lb)
    assert abs(out[0] - out[1]) <= 1e-5 * abs(out[0])


@pytest.mark.parametrize("use_graph", [False, True])
def test_step_is_bit_reproducible_in_deterministic_mode(use_graph, monkeypatch):
    """SURVEY.md section 5 (deterministic reductions + reproducibility test): with config.deterministic two independent runs
    of three training steps from the same weights and data produce identical losses, gradients and parameters -- every
    reduction of the step has a fixed order, the one scatter accumulates integers."""
    from cc_amd import config
    monkeypatch.setattr(config, "deterministic", True)
    dev = torch.device("cuda")
    bc = syn.sample(2, 128, 192, seed=1)
    batch = (bc[0].to(dev), [r.to(dev) for r in bc[1]], bc[2].to(dev), bc[3].to(dev))
    runs = []
    for _ in range(2):
        nets = T.build_nets(dev, init=False)
        for n in nets:
            n.load_state_dict(syn.seeded_state_dict(n, 0))
        tr = T.CCTrainer(nets, T.StepConfig(), use_graph=use_graph)
        ls = [{k: float(v) for k, v in tr.step(batch).items()} for _ in range(3)]
        runs.append((ls, tr.opt.flat_g.clone(), tr.opt.flat_p.clone()))
    (la, ga, pa), (lb, gb, pb) = runs
    assert la == lb, (la, lb)
    assert torch.equal(ga, gb) and torch.equal(pa, pb)


def test_step_with_rccl_process_group_of_one(monkeypatch):
    """The multi-GPU code paths on the one GPU a test box has: a 1-rank RCCL (backend 'nccl') process group with
    config.debug.force_comm.  per_network (the default): parameter broadcast, the trainer's own RCCL communicator, every network's
    ncclAllReduce captured on its stream as a node of the ONE graph, Adam segments, weight images; post / staged: the process
    group's asynchronous all-reduces behind / between the graphs.  All must reproduce the plain single-process step."""
    import socket
    import torch.distributed as dist
    dev = torch.device("cuda")
    bc = syn.sample(2, 128, 192, seed=1)
    batch = (bc[0].to(dev), [r.to(dev) for r in bc[1]], bc[2].to(dev), bc[3].to(dev))

    def run(form=None):
        nets = T.build_nets(dev, init=False)
        for n in nets:
            n.load_state_dict(syn.seeded_state_dict(n, 0))
        tr = T.CCTrainer(nets, T.StepConfig(), use_graph=True, comm_debug={"events": True}, pipeline=form)
        l1 = {k: float(v) for k, v in tr.step(batch).items()}
        l2 = {k: float(v) for k, v in tr.step(batch).items()}
        l3 = {k: float(v) for k, v in tr.step(batch).items()}
        torch.cuda.synchronize()
        return tr, l1, l2, l3
    _, a1, a2, a3 = run()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", str(port))
    from cc_amd import config as _cfg
    monkeypatch.setattr(_cfg.debug, "force_comm", True)
    dist.init_process_group("nccl", rank=0, world_size=1)
    results = []
    try:
        tr, b1, b2, b3 = run()
        results.append((b1, b2, b3))
        assert tr.pipeline == "per_network" and tr.graph_b is None and tr.opt._rccl is not None
        n = tr.opt.flat_g.numel()
        segs = [tr.opt.segment(i) for i in range(4)]
        assert [(lo, hi) for _, lo, hi in tr.segment_calls] == [segs[1], segs[2], segs[0], segs[3]]     # pose, mask, disp, flow
        alone = tr.calibrate_comm(reps=2)
        st = tr.comm_stats()
        assert len(alone) == 4 and all(a > 0 for a in alone)
        assert st["issue_order"] == ["pose", "mask", "disp", "flow"] and len(st["standalone_ms"]) == 4
        # config.grad_chunks: DispResNet6 hands its segment over in three chunks while its backward pass runs (on the origin stream,
        # its own communicator); same results to the summation order of the weight gradients parked at the marks
        monkeypatch.setattr(_cfg, "grad_chunks", True)
        trc, c1, c2, c3 = run()
        c5, dec = sorted(v for k, v in trc._chunk_lo.items() if k[0] == 0)
        assert [(lo, hi) for _, lo, hi in trc.segment_calls] == [segs[1], segs[2], (dec, segs[0][1]), (c5, dec), (segs[0][0], c5), segs[3]]
        assert trc.comm_stats()["issue_order"] == ["pose", "mask", "disp", "disp", "disp", "flow"]
        results.append((c1, c2, c3))
        monkeypatch.setattr(_cfg, "grad_chunks", False)
        assert abs(sum(st["segments_mb"]) - 4e-6 * n) < 0.3 and "ncclAllReduce" in st["collective"]
        for form in ("post", "staged"):
            tr, b1, b2, b3 = run(form)
            results.append((b1, b2, b3))
            assert tr.pipeline == form and (tr.graph_b is not None) == (form == "staged")
            # the record the legacy N > 1 bench line carries: exposed time of each segment's all-reduce (events around work.wait()),
            # each segment alone on an idle device, their difference
            alone = tr.calibrate_comm(reps=2)
            st = tr.comm_stats()
            assert len(alone) == 2 and all(a > 0 for a in alone)
            assert st["steps_sampled"] == 3 and len(st["exposed_ms"]) == 2 and len(st["segments_mb"]) == 2
            assert abs(sum(st["segments_mb"]) - 4e-6 * n) < 0.2 and st["overlapped_ms"] >= 0
    finally:
        dist.destroy_process_group()
    for b1, b2, b3 in results:
        for a, b in ((a1, b1), (a2, b2), (a3, b3)):
            for k in a:
                assert abs(a[k] - b[k]) <= 2e-5 * abs(a[k]) + 1e-9, (k, a[k], b[k])


def test_checkpoint_resume_on_device(tmp_path):
    """SURVEY.md 8(f) rank 3 on the device (utils.py:55-63, train.py:286-295,311-315,396-413): a device-resident, hipGraph-replayed
    CCTrainer (parameters = views of the flat bucket, device step counter, BatchNorm buffers updated inside the graph) is saved
    after two steps; a fresh set of networks + trainer resumes from the files and runs step 3: identical to step 3 of the trainer
    that never stopped (bit for bit in config.deterministic).  The files load into nets that carry the reference's keys (the
    oracle's) and into a real torch.optim.Adam."""
    import os
    from cc_amd import config, utils
    import oracle.nets as ON
    dev = torch.device("cuda")
    bc = syn.sample(2, 128, 192, seed=1)
    batch = (bc[0].to(dev), [r.to(dev) for r in bc[1]], bc[2].to(dev), bc[3].to(dev))
    old = config.deterministic
    config.deterministic = True
    try:
        nets = T.build_nets(dev, init=False)
        for n in nets:
            n.load_state_dict(syn.seeded_state_dict(n, 0))
        tr = T.CCTrainer(nets, T.StepConfig(), use_graph=True)
        tr.step(batch)
        tr.step(batch)
        torch.cuda.synchronize()
        tr.save_checkpoint(tmp_path, epoch=0, is_best=False)
        l3 = {k: float(v) for k, v in tr.step(batch).items()}
        torch.cuda.synchronize()
        p3 = tr.opt.flat_p.clone()
        m3 = tr.opt.exp_avg.clone()
        bn3 = torch.cat([b.detach().double().reshape(-1) for n in nets for b in n.buffers()])
        # files are per-tensor storages (not views that drag the whole 297 MB bucket along) with the reference's layout: the four
        # network files add up to the parameters + buffers once, the optimizer file to the two moment sets
        nbytes = {p: os.path.getsize(os.path.join(tmp_path, p + "_checkpoint.pth.tar")) for p in utils.FILE_PREFIXES}
        assert sum(nbytes[p] for p in utils.FILE_PREFIXES[:4]) < 1.1 * 4 * tr.opt.n + 4e6, nbytes
        assert nbytes["optimizer"] < 1.05 * 8 * tr.opt.n + 4e6, nbytes
        ref_nets = [ON.DispResNet6(), ON.PoseNetB6(nb_ref_imgs=4), ON.MaskNet6(nb_ref_imgs=4, output_exp=True), ON.Back2Future(nlevels=6)]
        for prefix, net in zip(utils.FILE_PREFIXES, ref_nets):
            w = torch.load(os.path.join(tmp_path, prefix + "_checkpoint.pth.tar"), map_location="cpu")
            assert set(w.keys()) == {"epoch", "state_dict"} and w["epoch"] == 1
            net.load_state_dict(w["state_dict"])                       # strict: the reference's keys and shapes
        params = [p for n in ref_nets for p in n.parameters()]
        topt = torch.optim.Adam(params, lr=1e-4, betas=(0.9, 0.999))
        topt.load_state_dict(torch.load(os.path.join(tmp_path, "optimizer_checkpoint.pth.tar"), map_location="cpu")["state_dict"])
        assert float(topt.state[params[0]]["step"]) == 2.0
        # resume: train.py's order -- networks first, then the optimizer over them
        nets2 = T.build_nets(dev, init=True)
        assert utils.resume(tmp_path, *nets2, map_location=dev) == 1
        tr2 = T.CCTrainer(nets2, T.StepConfig(), use_graph=True)
        assert utils.resume_optimizer(tmp_path, tr2, map_location=dev) is True
        assert float(tr2.opt.step_dev) == 2.0
        r3 = {k: float(v) for k, v in tr2.step(batch).items()}
        torch.cuda.synchronize()
        for k in l3:
            assert l3[k] == r3[k], (k, l3[k], r3[k])
        assert torch.equal(tr2.opt.flat_p, p3) and torch.equal(tr2.opt.exp_avg, m3)
        bn3b = torch.cat([b.detach().double().reshape(-1) for n in nets2 for b in n.buffers()])
        assert torch.equal(bn3, bn3b)
    finally:
        config.deterministic = old


@pytest.mark.parametrize("use_graph,size", [(False, (2, 128, 192)), (True, (2, 128, 192)), (True, (4, 256, 832))])
def test_network_streams_change_nothing_but_the_schedule(use_graph, size, monkeypatch):
    """config.net_streams (round 5): DispResNet6 and Back2Future on HIP streams of their own, forward and backward, one backward call
    for all four networks.  Same kernels on the same operands in the same per-tensor order: in config.deterministic mode three steps
    with and without the side streams must give IDENTICAL losses, gradient bucket and parameters (a missing cross-stream dependency
    or a buffer re-used while another stream still reads it shows up as a difference), eagerly and under hipGraph replay; the run with
    streams is repeated to see that it reproduces itself."""
    from cc_amd import config
    monkeypatch.setattr(config, "deterministic", True)
    dev = torch.device("cuda")
    bc = syn.sample(*size, seed=3, smooth=3 if size[1] > 128 else 0)      # (the last case: the benchmarked configuration)
    batch = (bc[0].to(dev), [r.to(dev) for r in bc[1]], bc[2].to(dev), bc[3].to(dev))

    def run(streams):
        monkeypatch.setattr(config, "net_streams", streams)
        nets = T.build_nets(dev, init=False)
        for n in nets:
            n.load_state_dict(syn.seeded_state_dict(n, 0))
        tr = T.CCTrainer(nets, T.StepConfig(), use_graph=use_graph)
        assert (tr.net_streams is not None) == bool(streams)
        ls = [{k: float(v) for k, v in tr.step(batch).items()} for _ in range(3)]
        torch.cuda.synchronize()
        return ls, tr.opt.flat_g.clone(), tr.opt.flat_p.clone()
    a = run(False)
    others = [run(True), run(True)] + ([run(3)] if size[1] == 128 else [])
    for other in others:
        assert a[0] == other[0], (a[0], other[0])
        assert torch.equal(a[1], other[1]) and torch.equal(a[2], other[2])
