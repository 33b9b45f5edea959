"""Round 6 probe: can an RCCL all-reduce be a node of a hipGraph BRANCH -- issued from inside an autograd backward node that runs on a
side stream (where a network's backward runs in CCTrainer)?  Two ways of issuing it:
    pg      torch.distributed.all_reduce(async_op=True) + work.wait()   (ProcessGroupNCCL: own stream, events, watchdog thread)
    direct  cc_amd.rccl.Communicator.all_reduce_sum_  (ncclAllReduce on the caller's stream through ctypes)
One-rank group on one GPU: shows whether capture / replay work; it cannot show a transfer.
    python tools/rccl_capture_probe.py [pg|direct]      (default: direct, then pg in a child process -- pg can abort the process)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(mode):
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1)
    dev = torch.device("cuda:0")
    n = 56_800_000
    g = torch.ones(n, device=dev)
    w = torch.zeros(n, device=dev)
    comm = None
    if mode == "direct":
        from cc_amd import rccl
        comm = rccl.Communicator(dev)
        print("librccl version", rccl.version())
        comm.all_reduce_sum_(g)
    else:
        dist.all_reduce(g)            # communicator set-up outside any capture
    torch.cuda.synchronize()
    side = torch.cuda.Stream()

    class Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x * 2.0

        @staticmethod
        def backward(ctx, gy):
            # (the autograd engine runs this in its worker thread, on the stream of the forward: the side stream)
            g.mul_(1.5)
            if comm is not None:
                comm.all_reduce_sum_(g)
            else:
                work = dist.all_reduce(g, async_op=True)
                work.wait()
            w.add_(g)
            return gy * 2.0

    x = torch.ones(1024, device=dev, requires_grad=True)

    def body():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            y = Fn.apply(x)
        cur.wait_stream(side)
        gy = torch.ones_like(y)
        side.wait_stream(cur)
        torch.autograd.backward([y], [gy])
        cur.wait_stream(side)

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g.fill_(1.0)
    w.zero_()
    x.grad = None
    graph = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            body()
    except Exception as e:          # noqa: BLE001
        print("%s: CAPTURE FAILED: %s: %s" % (mode, type(e).__name__, str(e)[:400]))
        return 1
    print("%s: capture ok" % mode)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    want_g = 1.5 ** 3
    want_w = 1.5 + 1.5 ** 2 + 1.5 ** 3
    print("%s: g[0] %.6f (want %.6f)  w[0] %.6f (want %.6f)" % (mode, float(g[0]), want_g, float(w[0]), want_w))
    ok = abs(float(g[0]) - want_g) < 1e-5 and abs(float(w[0]) - want_w) < 1e-5
    g.fill_(1.0)
    t0 = time.perf_counter()
    for _ in range(20):
        g.fill_(1.0)
        graph.replay()
    torch.cuda.synchronize()
    print("%s: replay %.3f ms per graph (three 227 MB elementwise passes + a one-rank all-reduce)" % (mode, (time.perf_counter() - t0) / 20 * 1e3))
    time.sleep(1.0)                 # (give a watchdog thread the time to trip over captured events)
    print("%s: %s" % (mode, "OK" if ok else "MISMATCH"))
    if comm is not None:
        comm.destroy()
    dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    if len(sys.argv) > 1:
        sys.exit(run(sys.argv[1]))
    rc = 0
    for mode in ("direct", "pg"):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), mode], capture_output=True, text=True, timeout=300)
        out = [ln for ln in (r.stdout + r.stderr).splitlines() if ln.startswith(mode) or "librccl" in ln or "terminated with exception" in ln]
        print("\n".join(out[:8]))
        print("%s: exit code %d" % (mode, r.returncode))
        if mode == "direct":
            rc = r.returncode
    sys.exit(rc)
