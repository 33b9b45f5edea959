#!/usr/bin/env python
"""Aggregate the FETCH_SIZE and WRITE_SIZE rocprofv3 --pmc passes of bench.py into profiles/pmc_traffic.json: per device kernel, average
HBM-side bytes per dispatch.

Round 6: the counters are calibrated per ACCESS PATTERN on known byte counts (tools/fetch_calib.hip ->
profiles/fetch_calib.json).  Result on gfx950 / this rocprofv3: FETCH_SIZE reports exactly HALF the bytes of a coalesced streaming
read at 4, 8 AND 16 bytes per lane (factor 2.00 each: the L2 fetches 128-byte lines and the counter tallies 64 per request), WRITE_SIZE
is exact (factor 1.00 at 4 and 16 bytes per lane).  For a GATHER (2 x 2 bilinear taps at a displaced coordinate, the warp kernels) the
same counter reads 1.62 x the DISTINCT bytes: the requests are the same 128-byte line fetches, but neighbouring rows are fetched by
workgroups on different XCDs (own L2 each) and taps straddle lines -- so for gather kernels the bytes that crossed the L2 boundary are
2 x FETCH_SIZE like everywhere else, and they are NOT the kernel's algorithmic bytes (1.6 - 3.2 x of them).  Every kernel therefore
gets   hbm_bytes_per_launch = 2 * FETCH_SIZE + WRITE_SIZE   (bytes moved), its access class, and -- what a roofline fraction must be
quoted on first -- nothing here: the ALGORITHMIC bytes come from bench.py's `kernels` block (tools/loss_bounds.py joins them)."""
import collections, csv, json, os, re, sys

GATHER = ("k_inverse_warp_fwd", "k_inverse_warp_bwd", "k_flow_warp_fwd", "k_flow_warp_bwd", "k_feature_warp", "k_corr_", "k_pose2flow")


def norm(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0].strip()


def load(path, counter):
    tot, cnt = collections.defaultdict(float), collections.Counter()
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            k = norm(r["Kernel_Name"])
            tot[k] += float(r["Counter_Value"])
            cnt[k] += 1
    return tot, cnt


fetch_csv, write_csv, out, command = sys.argv[1:5]
import ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = ctypes.CDLL(os.path.join(ROOT, "cc_amd", "libccengine.so"))
_lib.cc_version.restype = ctypes.c_size_t
cc_version = int(_lib.cc_version())
calib = {}
try:
    calib = json.load(open(os.path.join(ROOT, "profiles", "fetch_calib.json")))
except (OSError, ValueError):
    pass
f_stream = (calib.get("read", {}).get("k_read16", {}) or {}).get("factor") or 2.0
f_write = (calib.get("write", {}).get("k_write16", {}) or {}).get("factor") or 1.0
g = (calib.get("read", {}).get("k_gather4", {}) or {})
gather_overfetch = round(f_stream * g["FETCH_SIZE_KiB"] * 1024 / g["known_bytes"], 2) if g.get("known_bytes") else None
ft, fc = load(fetch_csv, "FETCH_SIZE")
wt, wc = load(write_csv, "WRITE_SIZE")
kern = {}
for k in ft:
    if fc[k] == 0 or wc.get(k, 0) == 0:
        continue
    f_kb, w_kb = ft[k] / fc[k], wt[k] / wc[k]
    kern[k] = {"launches": fc[k], "fetch_size_kib_avg": round(f_kb, 1), "write_size_kib_avg": round(w_kb, 1),
               "access": "gather" if k.startswith(GATHER) else "stream", "fetch_factor": f_stream, "write_factor": f_write,
               "hbm_bytes_per_launch": round((f_stream * f_kb + f_write * w_kb) * 1024)}
json.dump({"command": command, "cc_version": cc_version,
           "formula": "(fetch_factor*FETCH_SIZE + write_factor*WRITE_SIZE) KiB per dispatch, separate --pmc passes; factors calibrated by "
                      "tools/fetch_calib.hip (profiles/fetch_calib.json): streaming reads 2.00 at 4 / 8 / 16 B per lane, writes 1.00",
           "gather_note": "access = gather: bytes moved (this figure) exceed the kernel's distinct bytes -- the calibration gather moves "
                          "%s x its distinct bytes" % gather_overfetch,
           "kernels": kern}, open(out, "w"), indent=1, sort_keys=True)
for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]:
    print("%-60s n=%4d fetch %10.1f KiB write %10.1f KiB -> %8.2f MB/launch (%s)" % (k[:60], v["launches"], v["fetch_size_kib_avg"], v["write_size_kib_avg"], v["hbm_bytes_per_launch"] / 1e6, v["access"]))
