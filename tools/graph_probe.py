#!/usr/bin/env python
"""Where does the replayed step's time go OUTSIDE the kernels?  (rocprofv3's kernel trace makes the host the bottleneck of a graph
replay -- its timeline shows the branches in the order the host enqueues them, not the overlap of an unprofiled run.)

  1. steady state: N steps back to back (the bench number);
  2. host side of one replay: synchronise, call step(), time until the call RETURNS (= the runtime walking the graph's nodes) and until
     the device is done (= latency of one isolated step);
  3. forward + losses only (no backward): one graph with the networks on their side streams, one without -- do the four forward
     passes overlap?
  4. backward only = (1) - (3), both forms.

    python tools/graph_probe.py [--batch 4 --height 256 --width 832]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from tools import ab_env  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=832)
    ap.add_argument("--steps", type=int, default=30)
    args = ap.parse_args()
    ab_env.apply()
    from cc_amd import config, ops, synthetic as syn, tape, trainer as T, loss_functions as LF
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    nets = T.build_nets(dev)
    cfg = T.StepConfig()
    b = syn.sample(args.batch, args.height, args.width, seed=1, smooth=3)
    batch = (b[0].to(dev), [r.to(dev) for r in b[1]], b[2].to(dev), b[3].to(dev))

    class ForwardOnly(T.CCTrainer):
        """the step without its backward pass: weight images, zero grads, four forward passes, losses (then Adam on zero gradients)"""

        def _fwd_bwd(self, batch, between=None):
            try:
                tape.BN_COUNTERS = self.bn_counters
                self.bn_counters.begin()
                LF.pyramid_cache.clear()
                ops.packs.prepack_all()
                self.opt.zero_grad()
                ops.grad_sinks = self.opt.sinks
                LF.scalar_pool.begin(batch[0].device)
                LF.head_grads.begin()
                try:
                    with torch.no_grad():
                        out = T.cc_forward(self.nets, batch, self.cfg, streams=self.net_streams)
                    self.bn_counters.commit()
                finally:
                    LF.head_grads.end()
                return {k: v.detach() for k, v in out.items() if torch.is_tensor(v) and k.startswith("loss")}
            finally:
                self._stage_end()

    def steady(tr, n):
        for _ in range(4):
            tr.step(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            tr.step(batch)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def isolated(tr, n=10):
        host, lat = [], []
        for _ in range(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tr.step(batch)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            host.append((t1 - t0) * 1e3)
            lat.append((t2 - t0) * 1e3)
        host.sort(), lat.sort()
        return host[len(host) // 2], lat[len(lat) // 2]

    want = config.net_streams
    res = {}
    for label, ns in (("side streams", want), ("one stream", False)):
        if label == "side streams" and not want:
            continue
        config.net_streams = ns
        for kind, cls in (("step", T.CCTrainer), ("forward + losses", ForwardOnly)):
            tr = cls(nets, cfg)
            ms = steady(tr, args.steps)
            h, lt = isolated(tr)
            res[(label, kind)] = ms
            print("%-13s %-17s steady %.2f ms/step   one isolated replay: call returns after %.2f ms, device done after %.2f ms"
                  % (label, kind, ms, h, lt), flush=True)
            del tr
    config.net_streams = want
    for label in ("side streams", "one stream"):
        if (label, "step") in res:
            print("%-13s backward (step - forward): %.2f ms" % (label, res[(label, "step")] - res[(label, "forward + losses")]))


if __name__ == "__main__":
    main()
