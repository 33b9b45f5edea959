#!/usr/bin/env python
"""Partial-slab bytes of the weight-gradient reductions of one training step, by layer: which problems make up the traffic of
k_wgrad_reduce_table (slab bytes written by the weight-gradient kernels + read back by the table)?

    python tools/reduce_bytes.py            (one eager step at B = 4, 256 x 832)
Descriptor layout: csrc/wgrad_reduce.hip (kind, ws, gw, nsplit, accumulate, o_sm, o_sc, p0 .. p8)."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cc_amd import config, synthetic as syn, trainer as T  # noqa: E402


def slab_floats(d):
    kind, nsplit, p = d[0], d[3], d[7:16]
    if kind == 0:
        return nsplit * p[0] * p[1], "generic M%d N%d" % (p[0], p[1]), p[0] * p[1]
    if kind == 1:
        return nsplit * p[0] * p[1] * p[3], "3x3/wino M%d C%d" % (p[1], p[2]), p[0] * p[1] * p[2]
    if kind == 2:
        # slab[combo][pb][t][m16][c16]: ncombo * nsplit(pb) * taps * 256
        ts, tr = p[0], p[2]
        return p[8] * nsplit * ts * tr * 256, "thin M%d C%d" % (p[5], p[6]), p[5] * p[6] * p[7] * p[1]
    if kind == 4:
        return nsplit * p[0], "bias C%d" % p[0], p[0]
    return 0, "kind %d" % kind, 0


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    nets = T.build_nets(dev)
    b = syn.sample(4, 256, 832, seed=1, smooth=3)
    batch = (b[0].to(dev), [r.to(dev) for r in b[1]], b[2].to(dev), b[3].to(dev))
    tr = T.CCTrainer(nets, T.StepConfig(), use_graph=False)
    tr.step(batch)
    config.debug.reduce_trace = []
    tr.step(batch)
    torch.cuda.synchronize()
    desc = config.debug.reduce_trace
    config.debug.reduce_trace = None
    rows = collections.defaultdict(lambda: [0, 0, 0, 0])
    tot = wtot = 0
    for i in range(0, len(desc), 16):
        d = desc[i:i + 16]
        fl, name, wfl = slab_floats(d)
        key = "%s split %d" % (name, d[3])
        r = rows[key]
        r[0] += 1; r[1] += fl * 4; r[2] += wfl * 4; r[3] = d[3]
        tot += fl * 4
        wtot += wfl * 4
    print("%d reductions, %.1f MB of partial slabs for %.1f MB of gradients (x%.2f); slab traffic = 2 x that (written, read back)" %
          (len(desc) // 16, tot / 1e6, wtot / 1e6, tot / max(wtot, 1)))
    for k, r in sorted(rows.items(), key=lambda kv: -kv[1][1])[:40]:
        print("%9.1f MB slabs %8.1f MB grads  n %2d  %s" % (r[1] / 1e6, r[2] / 1e6, r[0], k))


if __name__ == "__main__":
    main()
