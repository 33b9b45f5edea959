"""Does packing independent layers of DIFFERENT networks into one launch pay?  (round 3, cc_conv2d_list)

Records the convolution problems of one forward pass of each network (B=4, 256x832), rebuilds them on fresh buffers as
launch-list records, captures three schedules into hipGraphs and times their replays:
  seq      every recorded call as its own list (what the per-network forward does today),
  lockstep step i of every network in one list (same-BM problems share a launch),
  greedy   at every step the largest same-BM set of track heads shares a launch.
Usage: python tools/merge_probe.py  (GPU)"""
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cc_amd import trainer as T, synthetic as syn, ops
from tools import launchlist as LL
from cc_amd._lib import engine


def record_tracks(dev, B=4, H=256, W=832):
    nets = T.build_nets(dev)
    batch = syn.sample(B, H, W, seed=1)
    tgt, refs = batch[0].to(dev), [r.to(dev) for r in batch[1]]
    E = engine()
    orig = E.call
    cur = []

    def spy(name, *a):
        if name == "cc_conv2d_fwd":
            cur.append([dict(kind=0, B=a[7], Cin=a[8], IH=a[9], IW=a[10], Cout=a[12], R=a[13], S=a[14], stride=a[15], pad=a[16],
                             OH=a[17], OW=a[18], act=a[21])])
        elif name == "cc_conv2d_fwd_group":
            cur.append([dict(kind=0, B=a[8], Cin=a[9], IH=a[10], IW=a[11], Cout=a[13], R=a[14], S=a[15], stride=a[16], pad=a[17],
                             OH=a[18], OW=a[19], act=a[22]) for _ in range(a[0])])
        elif name == "cc_conv2d_dgrad":       # ConvTranspose2d forward
            cur.append([dict(kind=1, B=a[6], Cin=a[7], IH=a[8], IW=a[9], Cout=a[11], R=a[12], S=a[13], stride=a[14], pad=a[15],
                             OH=a[16], OW=a[17], act=a[21])])
        return orig(name, *a)
    E.call = spy
    tracks = {}
    with torch.no_grad():
        for nm, f in (("disp", lambda: nets[0](tgt)), ("pose", lambda: nets[1](tgt, refs)), ("mask", lambda: nets[2](tgt, refs)),
                      ("flow", lambda: nets[3](tgt, refs[1:3]))):
            cur = []
            f()
            tracks[nm] = cur
    E.call = orig
    del nets
    torch.cuda.empty_cache()
    return tracks


def bm_of(M):
    return 128 if M > 64 else (64 if M > 32 else (32 if M > 16 else 16))


def gflop(p):
    return 2e-9 * p["B"] * p["OH"] * p["OW"] * p["Cout"] * p["Cin"] * p["R"] * p["S"] / (p["stride"] ** 2 if p["kind"] else 1)


def make_record(p, dev):
    B = p["B"]
    x = torch.randn(B, p["Cin"], p["IH"], p["IW"], device=dev)
    y = torch.empty(B, p["Cout"], p["OH"], p["OW"], device=dev)
    if p["kind"] == 0:
        w = torch.randn(p["Cout"], p["Cin"], p["R"], p["S"], device=dev) * 0.05
        return LL.conv_record(x, w, None, None, y, p["stride"], p["pad"], p["act"]), (x, w, y)
    w = torch.randn(p["Cin"], p["Cout"], p["R"], p["S"], device=dev) * 0.05
    return LL.tconv_record(x, w, None, y, p["stride"], p["pad"], p["Cout"] * p["R"] * p["S"], p["R"] * p["S"], p["R"], p["S"],
                           p["act"]), (x, w, y)


def time_graph(lists, reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for ll in lists:
            ll.run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for ll in lists:
            ll.run()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = torch.device("cuda:0")
    tracks = record_tracks(dev)
    ops.packs.reset()
    keep = []
    steps = {}
    for nm, tr in tracks.items():
        steps[nm] = []
        for st in tr:
            recs = []
            for p in st:
                r, k = make_record(p, dev)
                if r is None:
                    continue
                keep.append(k)
                recs.append((r, bm_of(p["Cout"]), gflop(p)))
            if recs:
                steps[nm].append(recs)
        print("%-5s %3d conv steps, %.1f GFLOP" % (nm, len(steps[nm]), sum(g for s in steps[nm] for _, _, g in s)))
    ops.packs.prepack_all()
    total_gf = sum(g for nm in steps for s in steps[nm] for _, _, g in s)

    def mk(recs, target=0):
        return LL.LaunchList([r for r, _, _ in recs], dev, target)
    res = {}
    # per-network sequential
    for nm in steps:
        ls = [mk(s) for s in steps[nm]]
        res["seq_" + nm] = time_graph(ls)
    seq_all = [mk(s) for nm in steps for s in steps[nm]]
    res["seq_all"] = time_graph(seq_all)
    # lockstep
    n = max(len(v) for v in steps.values())
    lock = []
    for i in range(n):
        recs = [r for nm in steps if i < len(steps[nm]) for r in steps[nm][i]]
        lock.append(mk(recs))
    res["lockstep"] = time_graph(lock)
    # greedy: heads grouped by bm; launch the bm class that holds the most GFLOP among classes with >= 2 tracks, else the head
    # of the track with the most remaining work
    pos = {nm: 0 for nm in steps}
    greedy = []
    nlaunch = 0
    while any(pos[nm] < len(steps[nm]) for nm in steps):
        heads = {nm: steps[nm][pos[nm]] for nm in steps if pos[nm] < len(steps[nm])}
        by_bm = {}
        for nm, st in heads.items():
            by_bm.setdefault(st[0][1], []).append(nm)
        multi = {b: v for b, v in by_bm.items() if len(v) >= 2}
        if multi:
            b = max(multi, key=lambda b: sum(g for nm in multi[b] for _, _, g in heads[nm]))
            pick = multi[b]
        else:
            rem = {nm: sum(g for s in steps[nm][pos[nm]:] for _, _, g in s) for nm in heads}
            pick = [max(rem, key=rem.get)]
        recs = [r for nm in pick for r in heads[nm]]
        greedy.append(mk(recs))
        nlaunch += 1
        for nm in pick:
            pos[nm] += 1
    res["greedy"] = time_graph(greedy)
    print("lists: seq %d, lockstep %d, greedy %d" % (len(seq_all), len(lock), nlaunch))
    for k, v in res.items():
        print("%-10s %7.3f ms" % (k, v))
    print("forward convs %.1f GFLOP: seq %.1f TF, lockstep %.1f TF, greedy %.1f TF" % (
        total_gf, total_gf / res["seq_all"], total_gf / res["lockstep"], total_gf / res["greedy"]))


if __name__ == "__main__":
    main()
