#!/usr/bin/env python
"""Every distinct cc_conv2d_wgrad shape of the BASELINE step (profiles/r01_conv_calls_v3.txt) timed under several
kernel-selection settings (env switches are read per call)."""
import os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cc_amd._lib import engine, STREAM

CONFIGS = [("t512", {}),
           ("t256", {"CC_WGRAD_SPLIT_TARGET": "256"}),
           ("t128", {"CC_WGRAD_SPLIT_TARGET": "128"}),
           ("t1024", {"CC_WGRAD_SPLIT_TARGET": "1024"}),
           ("w3s256", {"CC_W3_SPLIT": "256"}),
           ("w3s1024", {"CC_W3_SPLIT": "1024"})]
KEYS = ["CC_WGRAD_SPLIT_TARGET", "CC_W3_SPLIT", "CC_NO_WGRAD_THIN", "CC_WGRAD_THIN_MAXCOMBO", "CC_WGRAD_THIN_MINPIX", "CC_WGRAD_THIN_UPB", "CC_WGRAD_THIN_NPB"]

shapes = []
for l in open(os.path.join(os.path.dirname(__file__), "..", "profiles", "r01_conv_calls_v3.txt")):
    if "cc_conv2d_wgrad" not in l:
        continue
    n = int(re.search(r"n=\s*(\d+)", l).group(1))
    kv = dict(re.findall(r"(\w+)=(-?\d+)", l.split("cc_conv2d_wgrad")[1]))
    shapes.append((n, {k: int(v) for k, v in kv.items()}))

E = engine()
tot = {c: 0.0 for c, _ in CONFIGS}
best_tot = 0.0
print("%-58s %3s " % ("shape", "n") + " ".join("%9s" % c for c, _ in CONFIGS))
for n, s in shapes:
    B, M, AH, AW, Cin, IH, IW, R, S, si, pad = (s[k] for k in ("B", "M", "AH", "AW", "Cin", "IH", "IW", "R", "S", "si", "pad"))
    a = torch.randn(B, M, AH, AW, device="cuda"); x = torch.randn(B, Cin, IH, IW, device="cuda")
    row, ref = [], None
    for cname, env in CONFIGS:
        for k in KEYS:
            os.environ.pop(k, None)
        os.environ.update(env)
        gw = torch.zeros(M * s["o_sm"] if s["o_sm"] >= Cin * R * S else Cin * s["o_sm"] + M * s["o_sc"], device="cuda")
        ws = torch.empty(E.call("cc_conv2d_wgrad_ws_bytes", B, M, AH, AW, Cin, R, S, si) // 4 + 64, device="cuda")

        def run():
            E.call("cc_conv2d_wgrad", a, x, gw, ws, B, M, AH, AW, M * AH * AW, Cin, IH, IW, Cin * IH * IW, R, S, si, pad,
                   s["o_sm"], s["o_sc"], 0, STREAM)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        if ref is None:
            ref = gw.clone()
        else:
            err = float((gw - ref).abs().max() / (ref.abs().max() + 1e-30))
            assert err < 1e-4, (cname, s, err)
        row.append(ms); tot[cname] += n * ms
    best_tot += n * min(row)
    print("M%-4d C%-4d %dx%d s%d A %3dx%-3d %38s %3d " % (M, Cin, R, S, si, AH, AW, "", n) + " ".join("%9.4f" % v for v in row), flush=True)
print("TOTAL ms/step " + " ".join("%s %.3f" % (c, tot[c]) for c, _ in CONFIGS) + "  best-of %.3f" % best_tot)
