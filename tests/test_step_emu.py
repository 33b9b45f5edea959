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
        d = (tr.opt.flat_p[:po.numel()] - po).abs()
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


def _cc_dp_worker(rank, world, port, ret):
    """One rank of the REAL data-parallel CC step (CCTrainer.step: four nets, all losses, staged backward with the
    DispResNet6 + PoseNetB6 gradient segment reduced while MaskNet6 + Back2Future run backward, fused Adam)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "4"
    torch.set_num_threads(4)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, H, W = 2, 64, 128
    batch = syn.sample(B, H, W, seed=1 + rank)               # rank-specific shard, as bench.py draws it
    with emulated_engine():
        nets = T.build_nets("cpu", init=False)
        for n in nets:
            n.load_state_dict(syn.seeded_state_dict(n, 0))
            if rank == 1:                                   # the start-up broadcast must bring rank 1 back to rank 0's weights ...
                for p in n.parameters():
                    p.data.mul_(1.01)
                for b in n.buffers():                       # ... and buffers (BatchNorm running statistics / counters: a resume
                    if b.dtype.is_floating_point:           # that only rank 0 read from disk)
                        b.data.add_(0.5)
                    else:
                        b.data.add_(7)
        tr = T.CCTrainer(nets, T.StepConfig(), use_graph=False)
        assert tr.split_graphs and 0 < tr.n_dp < tr.opt.n
        p0 = tr.opt.flat_p.clone()
        buf0 = torch.cat([b.detach().double().reshape(-1) for n in nets for b in n.buffers()])
        local = torch.zeros_like(tr.opt.flat_g)
        calls = []
        orig = tr.opt.all_reduce

        def spy(lo=0, hi=None, async_op=False):
            hi = tr.opt.flat_g.numel() if hi is None else hi
            local[lo:hi] = tr.opt.flat_g[lo:hi]             # this rank's own gradient segment, complete at issue time
            calls.append((lo, hi, async_op))
            return orig(lo, hi, async_op)
        tr.opt.all_reduce = spy
        losses = tr.step(batch)
        ret[rank] = dict(p0=p0, buf0=buf0, local=local, reduced=tr.opt.flat_g.clone(), p1=tr.opt.flat_p.clone(), calls=calls,
                         n_dp=tr.n_dp, loss=float(losses["loss"]))
    dist.destroy_process_group()


@pytest.mark.slow
def test_cc_step_data_parallel_two_ranks_gloo():
    """world_size-2 gloo run of CCTrainer.step (train.py:300-303's DataParallel as one process per GPU): ranks agree bit for
    bit, the exchanged gradient is the sum of the two ranks' own gradients, the update equals Adam on their mean, and the
    exchange is issued as two segments with the first one started before the second backward stage."""
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_cc_dp_worker, args=(world, port, ret), nprocs=world, join=True)
    r0, r1 = ret[0], ret[1]
    assert torch.equal(r0["p0"], r1["p0"]), "broadcast from rank 0 did not equalise the start weights"
    assert r0["buf0"].numel() > 0 and torch.equal(r0["buf0"], r1["buf0"]), "broadcast from rank 0 did not equalise the module buffers"
    assert torch.equal(r0["reduced"], r1["reduced"]) and torch.equal(r0["p1"], r1["p1"]), "ranks diverged"
    assert torch.equal(r0["reduced"], r0["local"] + r1["local"])
    assert float(r0["local"].abs().sum()) > 0 and not torch.equal(r0["local"], r1["local"])
    n_dp, n = r0["n_dp"], r0["p0"].numel()
    assert [(lo, hi) for lo, hi, _ in r0["calls"]] == [(0, n_dp), (n_dp, n)] and all(a for _, _, a in r0["calls"])
    # single-process reference: torch.optim.Adam on the mean gradient
    p = r0["p0"].clone().requires_grad_(True)
    opt = torch.optim.Adam([p], lr=1e-4, betas=(0.9, 0.999))
    p.grad = (r0["local"] + r1["local"]) * 0.5
    opt.step()
    assert float((p.detach() - r0["p1"]).abs().max()) < 1e-6
