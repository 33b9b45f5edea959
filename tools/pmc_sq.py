#!/usr/bin/env python
"""Aggregate a rocprofv3 --pmc SQ_* pass per kernel: sums of every counter, launches, and MFMA-busy share
(SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES-style ratios are printed raw; see MI355X_MICROARCH.md for units)."""
import collections, csv, re, sys


def norm(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0].strip()


tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k = norm(r["Kernel_Name"])
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r.get("Dispatch_Id"), k)
        if key not in seen:
            seen.add(key)
            cnt[k] += 1
names = sorted({c for v in tot.values() for c in v})
print("kernel".ljust(46), "n".rjust(4), " ".join(c.replace("SQ_", "")[:14].rjust(15) for c in names))
rows = sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", kv[1].get("SQ_BUSY_CYCLES", 0)))
import os
for k, v in rows[:int(os.environ.get('PMC_ROWS', '28'))]:
    print(k[:46].ljust(46), str(cnt[k]).rjust(4), " ".join(("%.4g" % v.get(c, 0)).rjust(15) for c in names))
