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
        got2 = tr.step(batch)                                   # after one fused-Adam update
        want2 = S.cc_step(onets, oopt, batch, ocfg)
        assert abs(float(got2["loss"]) - want2["loss"]) <= 1e-4 * abs(want2["loss"])


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
            opt.step(opt.all_reduce())
            topt.zero_grad()
            ref(x).pow(2).sum().backward()
            topt.step()
        for a, b in zip(lin.parameters(), ref.parameters()):
            assert float((a - b).abs().max()) < 1e-6


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
        scale = opt.all_reduce()                        # ONE collective on the flat bucket
        opt.step(scale)
        ret[rank] = opt.flat_p.clone()
    dist.destroy_process_group()


def test_data_parallel_two_ranks_gloo():
    """world_size-2 gloo run of the DP path: flat-bucket all-reduce + identical fused Adam on every rank
    == single-process training on the mean gradient of the two shards."""
    world, port = 2, 29533
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
