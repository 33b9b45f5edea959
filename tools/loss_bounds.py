#!/usr/bin/env python
"""Which bound does each warp / loss kernel sit on?  (north_star: ">= 50 % HBM roofline on the warp + loss kernels")

Joins four artefacts of ONE build: the rocprofv3 kernel stats of the bench (average duration per launch), bench.py's JSON line (the
ALGORITHMIC bytes of each job-table call: SURVEY.md 8d's per-pixel figures x the pixels of the jobs), the PMC traffic pass (bytes
moved per launch, counters calibrated per access pattern: tools/pmc_traffic.py) and the SQ counter pass (wave time split: SQ_WAIT_ANY
= parked at s_waitcnt / barriers, SQ_WAIT_INST_ANY = issue stalls, SQ_ACTIVE_INST_ANY = issuing; VALU share = SQ_ACTIVE_INST_VALU /
SQ_WAVE_CYCLES -- all in quad-cycles, disjoint, summing to SQ_WAVE_CYCLES).

The roofline fraction is quoted on the ALGORITHMIC bytes / rocprofv3 duration (column `alg of 8`, SURVEY.md 8d's convention); bytes
moved / duration stands beside it (`moved of 8`: what the memory system was asked for -- above the algorithmic figure where a
kernel re-reads, e.g. the gathers' neighbouring rows fetched into several XCDs' L2s).

    python tools/loss_bounds.py <kernel_stats.csv> profiles/pmc_traffic.json <pmc_sq.txt> <bench json log>
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


ALG = {"k_inverse_warp_fwd_jobs": "cc_inverse_warp_fwd_jobs", "k_inverse_warp_bwd_jobs": "cc_inverse_warp_bwd_jobs",
       "k_flow_warp_fwd_jobs": "cc_flow_warp_fwd_jobs", "k_flow_warp_bwd_jobs": "cc_flow_warp_bwd_jobs",
       "k_pose2flow_fwd_jobs": "cc_pose2flow_fwd_jobs", "k_ssim_photo_jobs": "cc_ssim_photo_fwd_jobs",
       "k_ssim_adjoint_jobs": "cc_ssim_photo_bwd_jobs", "k_ssim_err_jobs": "cc_ssim_err_fwd_jobs"}


def main():
    stats, traffic, sq = sys.argv[1:4]
    alg = {}
    if len(sys.argv) > 4:
        for ln in open(sys.argv[4]):
            if ln.startswith("{"):
                kk = json.loads(ln).get("kernels") or {}
                for kn, cn in ALG.items():
                    e = kk.get(cn)
                    if e and e.get("calls"):
                        alg[kn] = e["gbps"] * e["ms"] * 1e6 / e["calls"]        # bytes per call (GB/s x ms)
    alg["k_adam"] = 28.0 * 74258164 / 4           # (one launch per network since round 6: the average segment)
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
    print("%-34s %8s %9s %8s | %9s %8s %6s | %7s %7s %7s %6s | %s" % ("kernel", "us", "alg MB", "alg of 8", "moved MB", "TB/s", "of 8", "parked", "stalled", "issuing", "VALU", "bound"))
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
        amb = alg.get(k, float("nan")) / 1e6
        afrac = amb / us * 1e3 / PEAK
        bound = "HBM" if afrac >= 0.5 else ("latency (waves parked at waitcnt / barriers)" if parked >= 0.45 else
                                            ("issue (VALU / address arithmetic)" if issuing + stalled >= 0.6 else "mixed: latency + issue"))
        acc = tr[tk].get("access", "") if tk else ""
        print("%-34s %8.1f %9.1f %8.2f | %9.1f %8.2f %6.2f | %7.2f %7.2f %7.2f %6.2f | %s%s" % (
            dk[:34], us, amb, afrac, mb, tbs, frac, parked, stalled, issuing, valu, bound, "  [gather: moved > distinct]" if acc == "gather" else ""))


if __name__ == "__main__":
    main()
