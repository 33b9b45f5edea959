#!/usr/bin/env python
"""Which bound does each warp / loss kernel sit on?  (north_star: ">= 50 % HBM roofline on the warp + loss kernels")

Joins three committed artefacts of ONE build: the rocprofv3 kernel stats of the bench (average duration per launch), the PMC
traffic pass (HBM bytes per launch, (2 x FETCH_SIZE + WRITE_SIZE) KiB: MI355X_MICROARCH.md's gfx950 correction) and the SQ counter
pass (wave time split: SQ_WAIT_ANY = parked at s_waitcnt / barriers, SQ_WAIT_INST_ANY = issue stalls, SQ_ACTIVE_INST_ANY =
issuing; VALU share = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES -- all in quad-cycles, disjoint, summing to SQ_WAVE_CYCLES).

    python tools/loss_bounds.py profiles/r05_rocprof_kernel_stats_fin.csv profiles/pmc_traffic.json profiles/r05_pmc_sq_fin.txt
"""
import csv
import json
import re
import sys

KERNELS = ["k_inverse_warp_fwd_jobs", "k_inverse_warp_bwd_jobs", "k_flow_warp_fwd_jobs", "k_flow_warp_bwd_jobs", "k_pose2flow_fwd_jobs",
           "k_ssim_photo_jobs", "k_ssim_adjoint_jobs", "k_ssim_err_jobs", "k_edge_smooth_jobs", "k_feature_warp_bwd4", "k_corr_fwd4",
           "k_corr_bwd_f2_4", "k_adam", "k_repack_table", "k_wgrad_reduce_table", "k_splitk_epilogue_multi", "k_splitk_epilogue"]
PEAK = 8000.0      # GB/s


def norm(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0].strip()


def main():
    stats, traffic, sq = sys.argv[1:4]
    dur = {}
    for r in csv.DictReader(open(stats)):
        dur[norm(r["Name"])] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
    tr = json.load(open(traffic))["kernels"]
    sqr, cols = {}, None
    for ln in open(sq):
        if ln.startswith("kernel"):
            cols = ln.split()[2:]
            continue
        m = re.match(r"^(\S.*?)\s+(\d+)\s+([-+.e0-9 ]+)$", ln.rstrip())
        if m and cols:
            vals = [float(v) for v in m.group(3).split()]
            if len(vals) == len(cols):
                sqr[m.group(1).strip()] = dict(zip(cols, vals))
    print("%-34s %8s %9s %8s %6s | %7s %7s %7s %6s | %s" % ("kernel", "us", "HBM MB", "TB/s", "of 8", "parked", "stalled", "issuing", "VALU", "bound"))
    for k in KERNELS:
        dk = next((n for n in dur if n == k or n.startswith(k + "<")), None)
        tk = next((n for n in tr if n == k or n.startswith(k + "<")), None)
        sk = next((n for n in sqr if n == k or n.startswith(k + "<")), None)
        if not dk:
            continue
        us = dur[dk][0]
        mb = tr[tk]["hbm_bytes_per_launch"] / 1e6 if tk else float("nan")
        tbs = mb / us
        c = sqr.get(sk, {})
        wc = c.get("WAVE_CYCLES", 0) or float("nan")
        parked, stalled, issuing = c.get("WAIT_ANY", 0) / wc, c.get("WAIT_INST_ANY", 0) / wc, c.get("ACTIVE_INST_AN", 0) / wc
        valu = c.get("ACTIVE_INST_VA", 0) / wc
        frac = tbs * 1e3 / PEAK
        bound = "HBM" if frac >= 0.5 else ("latency (waves parked at waitcnt / barriers)" if parked >= 0.45 else
                                           ("issue (VALU / address arithmetic)" if issuing + stalled >= 0.6 else "mixed: latency + issue"))
        print("%-34s %8.1f %9.1f %8.2f %6.2f | %7.2f %7.2f %7.2f %6.2f | %s" % (dk[:34], us, mb, tbs, frac, parked, stalled, issuing, valu, bound))


if __name__ == "__main__":
    main()
