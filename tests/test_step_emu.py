"""Host logic of the training step on CPU: the product's trainer (flat Adam bucket, step mirror of
train.py:445-568) driven through the x86 emulation build of the kernels, against the oracle; and the
data-parallel path with two gloo ranks."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cc_amd import synthetic as syn, trainer as T
from hipemu.emu import emulated_engine
from oracle import step as S


@pytest.mark.slow
def test_full_cc_step_matches_oracle():
    B, H, W = 2, 64, 128
    batch = syn.sample(B, H, W, seed=1)
    with emulated_engine():
        nets = T.build_nets("cpu", init=False)
        onets = S.build_nets("oracle")
        for a, b in zip(nets, onets):
            sd = syn.seeded_state_dict(b, 0)
            a.load_state_dict(sd)
            b.load_state_dict(sd)
            b.train()
        tr = T.CCTrainer(nets, T.StepConfig(), use_graph=False)
        got = tr.step(batch)
        ocfg = S.StepConfig()
        oopt = S.make_optimizer(onets, ocfg)
        want = S.cc_step(onets, oopt, batch, ocfg)
        for k, v in want.items():
            assert abs(float(got[k]) - v) <= 1e-4 * abs(v), (k, float(got[k]), v)
        # the fused-Adam update against torch.optim.Adam's on the oracle nets (first step: |update| = lr for every
        # parameter whose gradient is not ~0, where the sign of a 1e-12 gradient decides) -- parameter order is the same
        po = torch.cat([p.detach().reshape(-1) for n in onets for p in n.parameters()])
        d = (tr.opt.gather(tr.opt.flat_p) - po).abs()
        assert float((d > 1e-6).float().mean()) < 1e-3 and float(d.max()) <= 2.001e-4, (float((d > 1e-6).float().mean()), float(d.max()))
        # module buffers after the step: BatchNorm running statistics and num_batches_tracked (the trainer keeps the counters of
        # all layers in one buffer and bumps them with one add) -- same keys, same values as the reference modules'
        nb = 0
        for a, b in zip(nets, onets):
            sa, sb = a.state_dict(), b.state_dict()
            assert list(sa.keys()) == list(sb.keys())
            for k in sb:
                if k.endswith("num_batches_tracked"):
                    assert sa[k].shape == sb[k].shape and int(sa[k]) == int(sb[k]) == 1, k
                    nb += 1
                elif "running_" in k:
                    assert torch.allclose(sa[k], sb[k], rtol=1e-4, atol=1e-6), k
        assert nb == 13
        got2 = tr.step(batch)
        assert all(int(v) == 2 for k, v in nets[0].state_dict().items() if k.endswith("num_batches_tracked"))
        assert abs(float(got2["loss"])) < 1e3


def test_flat_adam_matches_torch_adam():
    torch.manual_seed(0)
    with emulated_engine():
        lin = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
        ref = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
        ref.load_state_dict(lin.state_dict())
        cfg = T.StepConfig()
        opt = T.FlatAdam([lin], cfg)
        topt = torch.optim.Adam(ref.parameters(), lr=cfg.lr, betas=cfg.betas)
        x = torch.randn(4, 7)
        for _ in range(3):
            opt.zero_grad()
            lin(x).pow(2).sum().backward()
            opt.all_reduce()
            opt.step(opt.grad_scale())
            topt.zero_grad()
            ref(x).pow(2).sum().backward()
            topt.step()
        for a, b in zip(lin.parameters(), ref.parameters()):
            assert float((a - b).abs().max()) < 1e-6


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _dp_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    with emulated_engine():
        net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.Conv2d(4, 2, 3, padding=1))
        if rank == 1:                                   # different init on rank 1: the broadcast must fix it
            for p in net.parameters():
                p.data.add_(1.0)
        cfg = T.StepConfig()
        opt = T.FlatAdam([net], cfg)
        opt.broadcast_from_rank0()
        x = torch.randn(2, 3, 8, 8, generator=torch.Generator().manual_seed(10 + rank))   # rank-specific shard
        opt.zero_grad()
        net(x).pow(2).mean().backward()
        opt.all_reduce()                                # ONE collective on the flat bucket
        opt.step(opt.grad_scale())
        ret[rank] = opt.flat_p.clone()
    dist.destroy_process_group()


def test_data_parallel_two_ranks_gloo():
    """world_size-2 gloo run of the DP path: flat-bucket all-reduce + identical fused Adam on every rank
    == single-process training on the mean gradient of the two shards."""
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_dp_worker, args=(world, port, ret), nprocs=world, join=True)
    assert torch.equal(ret[0], ret[1]), "ranks diverged"
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.Conv2d(4, 2, 3, padding=1))
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, betas=(0.9, 0.999))
    loss = 0
    for r in range(world):
        x = torch.randn(2, 3, 8, 8, generator=torch.Generator().manual_seed(10 + r))
        loss = loss + net(x).pow(2).mean() / world
    loss.backward()
    opt.step()
    flat = torch.cat([p.data.reshape(-1) for p in net.parameters()])
    assert float((ret[0][:flat.numel()] - flat).abs().max()) < 1e-6


def _cc_dp_worker_one_step(rank, world, port, ret, pipeline):
    """as above, one step only: returns the exchanged gradient and the updated parameters of THAT step"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "4"
    torch.set_num_threads(4)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    batch = syn.sample(2, 64, 128, seed=1 + rank)
    with emulated_engine():
        nets = T.build_nets("cpu", init=False)
        for n in nets:
            n.load_state_dict(syn.seeded_state_dict(n, 0))
            if rank == 1:
                for p in n.parameters():
                    p.data.mul_(1.01)
                for b in n.buffers():
                    if b.dtype.is_floating_point:
                        b.data.add_(0.5)
                    else:
                        b.data.add_(7)
        from cc_amd import config as _cfg
        _cfg.grad_chunks = True             # (off by default; the chunked hand-over of DispResNet6's segment is the superset of the logic)
        tr = T.CCTrainer(nets, T.StepConfig(), use_graph=False, pipeline=pipeline)
        assert tr.pipeline == pipeline and tr.split_graphs == (pipeline == "staged") and 0 < tr.n_dp < tr.opt.flat_p.numel()
        p0 = tr.opt.flat_p.clone()
        buf0 = torch.cat([b.detach().double().reshape(-1) for n in nets for b in n.buffers()])
        local = torch.zeros_like(tr.opt.flat_g)
        calls, events = [], []
        orig, orig_here, orig_seg = tr.opt.all_reduce, tr.opt.all_reduce_here, tr.opt.step_segment

        def spy(lo=0, hi=None, async_op=False):
            hi = tr.opt.flat_g.numel() if hi is None else hi
            local[lo:hi] = tr.opt.flat_g[lo:hi]             # this rank's own gradient segment, complete at issue time
            calls.append((lo, hi, async_op))
            return orig(lo, hi, async_op)

        def spy_here(lo, hi, comm=0):
            local[lo:hi] = tr.opt.flat_g[lo:hi]
            calls.append((lo, hi, False))
            events.append(("reduce", lo, hi))
            return orig_here(lo, hi, comm)

        def spy_seg(lo, hi, tick, grad_scale=1.0):
            events.append(("adam", lo, hi, bool(tick), grad_scale))
            return orig_seg(lo, hi, tick, grad_scale)
        tr.opt.all_reduce, tr.opt.all_reduce_here, tr.opt.step_segment = spy, spy_here, spy_seg
        losses = tr.step(batch)
        from cc_amd import ops
        images_fresh = all(e["ok"] for e in ops.packs.entries.values() if e)
        # (clones, and ONE assignment at the end: tensors handed to the manager live in shared memory, and a nested update of a
        # managed dict is lost)
        out = dict(p0=p0, buf0=buf0, local=local.clone(), reduced=tr.opt.flat_g.clone(), p1=tr.opt.flat_p.clone(), calls=list(calls),
                   events=list(events), n_dp=tr.n_dp, segs=[tr.opt.segment(i) for i in range(4)], loss=float(losses["loss"]),
                   chunks=sorted(v for k, v in tr._chunk_lo.items() if k[0] == 0),
                   step=float(tr.opt.step_dev), images_fresh=images_fresh, n_images=sum(1 for e in ops.packs.entries.values() if e))
        if pipeline == "per_network":
            # the weight images the per-network refreshes left behind ARE the images of the updated weights: a full rebuild gives
            # the same bytes, and the second step runs from them without a start-of-step launch
            before = [e["buf"].clone() for e in ops.packs.entries.values() if e]
            ops.packs.mark_stale()
            ops.packs.prepack_all()
            after = [e["buf"] for e in ops.packs.entries.values() if e]
            out["images_equal"] = all(torch.equal(a, b) for a, b in zip(before, after))
            ops.packs.end_step()
            losses2 = tr.step(batch)
            out["loss2"] = float(losses2["loss"])
            out["step2"] = float(tr.opt.step_dev)
        ret[rank] = out
    dist.destroy_process_group()


@pytest.mark.slow
@pytest.mark.parametrize("pipeline", ["per_network", "staged"])
def test_cc_step_data_parallel_two_ranks_gloo(pipeline):
    """world_size-2 gloo run of CCTrainer.step (train.py:300-303's DataParallel as one process per GPU): ranks agree bit for
    bit, the exchanged gradient is the sum of the two ranks' own gradients, the update equals Adam on their mean.
    per_network (the default form): the exchange is issued as the networks' segments in the order their backward passes are
    enqueued (pose, mask, disp -- in three chunks, as its backward pass passes its marks --, flow), each followed by ITS Adam
    segment with the step counter advanced once at the start, and the weight images left behind equal a full rebuild.  staged: two segments, the first one started before the second backward stage."""
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_cc_dp_worker_one_step, args=(world, port, ret, pipeline), nprocs=world, join=True)
    r0, r1 = ret[0], ret[1]
    assert torch.equal(r0["p0"], r1["p0"]), "broadcast from rank 0 did not equalise the start weights"
    assert r0["buf0"].numel() > 0 and torch.equal(r0["buf0"], r1["buf0"]), "broadcast from rank 0 did not equalise the module buffers"
    assert torch.equal(r0["reduced"], r1["reduced"]) and torch.equal(r0["p1"], r1["p1"]), "ranks diverged"
    assert torch.equal(r0["reduced"], r0["local"] + r1["local"])
    assert float(r0["local"].abs().sum()) > 0 and not torch.equal(r0["local"], r1["local"])
    n_dp, n = r0["n_dp"], r0["p0"].numel()
    if pipeline == "staged":
        assert [(lo, hi) for lo, hi, _ in r0["calls"]] == [(0, n_dp), (n_dp, n)] and all(a for _, _, a in r0["calls"])
    else:
        segs = r0["segs"]                               # disp, pose, mask, flow
        assert all(s is not None for s in segs) and segs[0][0] == 0 and segs[3][1] == n and segs[2][0] == n_dp
        assert all(a[1] == b[0] for a, b in zip(segs, segs[1:])) and all(lo % 64 == 0 for lo, _ in segs)     # they tile the bucket
        # issue order: pose, mask, disp, flow (shortest backward first); DispResNet6 -- the last finisher -- hands its segment over in
        # chunks while its backward pass still runs: the decoder's parameters, then conv5..conv7, then the rest
        c5, dec = r0["chunks"]
        assert segs[0][0] < c5 < dec < segs[0][1] and c5 % 4 == 0 and dec % 4 == 0
        want = [segs[1], segs[2], (dec, segs[0][1]), (c5, dec), (segs[0][0], c5), segs[3]]
        assert [(lo, hi) for lo, hi, _ in r0["calls"]] == want == [(lo, hi) for lo, hi, _ in r1["calls"]]
        # every segment: its all-reduce, then ITS Adam segment (no tick: the counter was advanced once at the start of the step)
        ev = r0["events"]
        assert len(ev) == 12
        for k, (lo, hi) in enumerate(want):
            assert ev[2 * k] == ("reduce", lo, hi) and ev[2 * k + 1] == ("adam", lo, hi, False, 0.5), (k, ev[2 * k], ev[2 * k + 1])
        assert r0["step"] == 1.0 and r0["step2"] == 2.0
        assert r0["images_fresh"] and r0["n_images"] > 100 and r0["images_equal"]
        assert abs(r0["loss2"]) < 1e3 and r0["loss2"] != r0["loss"]
    # single-process reference: torch.optim.Adam on the mean gradient
    p = r0["p0"].clone().requires_grad_(True)
    opt = torch.optim.Adam([p], lr=1e-4, betas=(0.9, 0.999))
    p.grad = (r0["local"] + r1["local"]) * 0.5
    opt.step()
    assert float((p.detach() - r0["p1"]).abs().max()) < 1e-6


def test_per_network_pipeline_falls_back_loudly_without_direct_rccl(monkeypatch, capsys):
    """Data-parallel start-up on a machine where the direct RCCL binding cannot be used: the trainer says so on stderr and runs
    round 5's form (process-group all-reduces behind the step) instead of failing."""
    from cc_amd import config as _cfg
    port = _free_port()
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", str(port))
    monkeypatch.setattr(_cfg.debug, "force_comm", True)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        def boom(self):
            raise OSError("librccl.so: cannot open shared object file")
        monkeypatch.setattr(T.FlatAdam, "rccl", boom)
        with emulated_engine():
            nets = T.build_nets("cpu", flow=False, mask=False, init=True)
            tr = T.CCTrainer(nets, T.StepConfig(), use_graph=False)
        assert tr.pipeline == "post" and not tr.split_graphs
        assert "falling back to pipeline='post'" in capsys.readouterr().err
    finally:
        dist.destroy_process_group()
