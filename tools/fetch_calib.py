#!/usr/bin/env python
"""FETCH_SIZE / WRITE_SIZE calibration per access pattern -> profiles/fetch_calib.json.
    python tools/fetch_calib.py <fetch counter_collection.csv> <write counter_collection.csv> <stdout of tools/_bin/fetch_calib> <out.json>
factor = known bytes per launch / (counter in KiB * 1024): the number a counter reading is MULTIPLIED by to get bytes."""
import collections, csv, json, re, sys


def load(path, counter):
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0].strip()
            tot[k] += float(r["Counter_Value"])
            cnt[k] += 1
    return {k: tot[k] / cnt[k] for k in tot}


fetch_csv, write_csv, log, out = sys.argv[1:5]
known = {}
for ln in open(log):
    if ln.startswith("CALIB bytes"):
        t = ln.split()[2:]
        known = {t[i]: int(t[i + 1]) for i in range(0, len(t), 2)}
f, w = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
res = {"what": "bytes moved per unit of the rocprofv3 counter (KiB * 1024), per access pattern, on gfx950 / this rocprofv3; "
               "tools/fetch_calib.hip moves a known byte count in each pattern", "read": {}, "write": {}}
for k, b in sorted(known.items()):
    if k.startswith("k_write"):
        res["write"][k] = {"known_bytes": b, "WRITE_SIZE_KiB": round(w.get(k, 0.0), 1), "factor": round(b / (w[k] * 1024), 4) if w.get(k) else None,
                           "FETCH_SIZE_KiB": round(f.get(k, 0.0), 1)}
    else:
        res["read"][k] = {"known_bytes": b, "FETCH_SIZE_KiB": round(f.get(k, 0.0), 1), "factor": round(b / (f[k] * 1024), 4) if f.get(k) else None,
                          "WRITE_SIZE_KiB": round(w.get(k, 0.0), 1)}
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
for sec in ("read", "write"):
    for k, v in res[sec].items():
        print("%-10s known %12d B  counter %12.1f KiB  -> factor %s" % (k, v["known_bytes"], v["FETCH_SIZE_KiB" if sec == "read" else "WRITE_SIZE_KiB"], v["factor"]))
