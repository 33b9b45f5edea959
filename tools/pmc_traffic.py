#!/usr/bin/env python
"""Aggregate the FETCH_SIZE and WRITE_SIZE rocprofv3 --pmc passes of bench.py into profiles/pmc_traffic.json:
per device kernel, average HBM bytes per dispatch = (2 * FETCH_SIZE + WRITE_SIZE) KiB (gfx950: FETCH_SIZE reports half of
the bytes of wide coalesced reads, /opt/skills/guides/MI355X_MICROARCH.md section HBM; WRITE_SIZE is taken as is)."""
import collections, csv, json, re, sys


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
import ctypes, os
_lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cc_amd", "libccengine.so"))
_lib.cc_version.restype = ctypes.c_size_t
cc_version = int(_lib.cc_version())
ft, fc = load(fetch_csv, "FETCH_SIZE")
wt, wc = load(write_csv, "WRITE_SIZE")
kern = {}
for k in ft:
    if fc[k] == 0 or wc.get(k, 0) == 0:
        continue
    f_kb, w_kb = ft[k] / fc[k], wt[k] / wc[k]
    kern[k] = {"launches": fc[k], "fetch_size_kib_avg": round(f_kb, 1), "write_size_kib_avg": round(w_kb, 1),
               "hbm_bytes_per_launch": round((2.0 * f_kb + w_kb) * 1024)}
json.dump({"command": command, "cc_version": cc_version, "formula": "(2*FETCH_SIZE + WRITE_SIZE) KiB per dispatch, separate --pmc passes",
           "kernels": kern}, open(out, "w"), indent=1, sort_keys=True)
for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]:
    print("%-60s n=%4d fetch %10.1f KiB write %10.1f KiB -> %8.2f MB/launch" % (k[:60], v["launches"], v["fetch_size_kib_avg"], v["write_size_kib_avg"], v["hbm_bytes_per_launch"] / 1e6))
