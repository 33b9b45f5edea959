#!/usr/bin/env python
"""Two DispResNet6 encoder stages in isolation (conv5-like: two BasicBlocks at 8x26 with 512 channels; conv6-like: stride 2 onto 4x13
with the 1x1 + BatchNorm shortcut) on the tape, against the same blocks in float64 and in fp32 on the CPU: are the engine's kernels at
exactly these shapes (padded-input Winograd, stride-2 data-gradient parity classes on stacked tiny maps, 1x1 stride-2 data-gradient,
small-map BatchNorm) as close to the exact gradients as the CPU's fp32?  (round 5: `conv5.1.conv{1,2}.weight` of the full network are
3.5e-3 from float64 in the engine and 1.7e-4 in the reference.)

    python tools/block_probe.py            (GPU; or CC_EMU=1 for the x86 emulation of the kernels)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def l2(a, b):
    return float(((a.double().cpu() - b.double()) ** 2).sum().sqrt() / (b.double() ** 2).sum().sqrt().clamp_min(1e-300))


def main():
    from oracle import nets as ON
    from cc_amd.models._blocks import make_layer, basic_block
    from cc_amd.tape import run_network
    dev = torch.device("cuda", 0) if torch.cuda.is_available() else torch.device("cpu")
    torch.manual_seed(0)
    C, B, H, W = int(os.environ.get("PC", 512)), 4, 8, 26
    for mode in ("iid", "smooth"):
        ra, rb = ON._res_stage(C, C, 2, 1), ON._res_stage(C, C, 2, 2)
        for m in list(ra.modules()) + list(rb.modules()):
            if isinstance(m, torch.nn.Conv2d):
                torch.nn.init.xavier_uniform_(m.weight)
        g = torch.Generator().manual_seed(1)
        x = torch.randn(B, C, H, W, generator=g).relu() * 0.3
        if mode == "smooth":        # deep features of a low-resolution map: a per-channel level with little spatial variation
            x = (torch.rand(1, C, 1, 1, generator=g) * 0.5 + 0.02 * torch.randn(B, C, H, W, generator=g)).relu()
        ga = torch.randn(B, C, H, W, generator=g) * 1e-3
        gb = torch.randn(B, C, (H + 1) // 2, (W + 1) // 2, generator=g) * 1e-3

        def ref(dt):
            a, b = ON._res_stage(C, C, 2, 1).to(dt), ON._res_stage(C, C, 2, 2).to(dt)
            a.load_state_dict({k: v.to(dt) if v.is_floating_point() else v for k, v in ra.state_dict().items()})
            b.load_state_dict({k: v.to(dt) if v.is_floating_point() else v for k, v in rb.state_dict().items()})
            a.train(); b.train()
            xx = x.detach().clone().to(dt).requires_grad_(True)
            ya = a(xx)
            yb = b(ya)
            torch.autograd.backward([ya, yb], [ga.to(dt), gb.to(dt)])
            out = {"x": xx.grad}
            out.update({"a." + n: p.grad for n, p in a.named_parameters()})
            out.update({"b." + n: p.grad for n, p in b.named_parameters()})
            return out, ya.detach(), yb.detach()
        g64, ya64, yb64 = ref(torch.float64)
        g32, ya32, yb32 = ref(torch.float32)
        ea, eb = make_layer(C, C, 2, 1).to(dev), make_layer(C, C, 2, 2).to(dev)
        ea.load_state_dict(ra.state_dict()); eb.load_state_dict(rb.state_dict())
        ea.train(); eb.train()
        xe = x.detach().clone().to(dev).requires_grad_(True)

        def body(tape, xt):
            a1 = basic_block(tape, ea[0], xt)
            a2 = basic_block(tape, ea[1], a1)
            b1 = basic_block(tape, eb[0], a2)
            b2 = basic_block(tape, eb[1], b1)
            return [a2, b2]
        outs = run_network(body, [xe], list(ea.parameters()) + list(eb.parameters()))
        torch.autograd.backward(list(outs), [ga.to(dev), gb.to(dev)])
        ge = {"x": xe.grad}
        ge.update({"a." + n: p.grad for n, p in ea.named_parameters()})
        ge.update({"b." + n: p.grad for n, p in eb.named_parameters()})
        print("== input model %s: forward stage A engine %.2e / cpu fp32 %.2e, stage B %.2e / %.2e (vs float64)" %
              (mode, l2(outs[0], ya64), l2(ya32, ya64), l2(outs[1], yb64), l2(yb32, yb64)))
        for k in g64:
            if g64[k] is None or ge.get(k) is None or g32.get(k) is None:
                continue
            print("   %-28s engine %.2e   cpu fp32 %.2e" % (k, l2(ge[k], g64[k]), l2(g32[k], g64[k])))


if __name__ == "__main__":
    if os.environ.get("CC_EMU"):
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
        from hipemu.emu import emulated_engine
        with emulated_engine():
            main()
    else:
        main()
