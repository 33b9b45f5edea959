#!/usr/bin/env python
"""MFMA utilisation per conv kernel from a rocprofv3 --pmc pass (north_star: "MFMA utilisation (convs) against gfx950 peaks"):
    busy       = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES)    share of the SIMD-cycles of BUSY CUs in which the matrix pipe works
    of chip    = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x GRBM_GUI_ACTIVE)   the same over all 1024 SIMDs for the kernel's whole duration
    TFLOP/s    = 512 x SQ_INSTS_VALU_MFMA_MOPS_F32 / duration          (cross-check of the FLOP counts bench.py uses; duration from the
                                                                        kernel stats csv of the serial profile, if given)
Counters are summed over the chip per dispatch (MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles, 64 per v_mfma_f32_32x32x2_f32).
    python tools/pmc_mfma.py <counter_collection.csv> [<kernel_stats.csv>]"""
import collections, csv, re, sys


def norm(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0].strip()


tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = norm(r["Kernel_Name"])
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r.get("Dispatch_Id"), k)
    if key not in seen:
        seen.add(key)
        cnt[k] += 1
dur = {}
if len(sys.argv) > 2:
    for r in csv.DictReader(open(sys.argv[2])):
        dur[norm(r["Name"])] = float(r["AverageNs"])
print("%-44s %5s %10s %8s %8s %10s %9s" % ("kernel", "n", "MFMA inst", "busy", "of chip", "TFLOP/s", "of 157.3"))
rows = sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0))
for k, v in rows[:24]:
    mb, cu, ga = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), v.get("SQ_BUSY_CU_CYCLES", 0), v.get("GRBM_GUI_ACTIVE", 0)
    if mb <= 0:
        continue
    mops = v.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0)
    tf = (512.0 * mops / cnt[k]) / dur[k] * 1e-3 if k in dur and dur[k] > 0 else float("nan")
    print("%-44s %5d %10.3g %8.3f %8.3f %10.1f %9.3f" % (k[:44], cnt[k], v.get("SQ_INSTS_VALU_MFMA_F32", 0) / cnt[k],
                                                       mb / (4.0 * cu) if cu else float("nan"), mb / (1024.0 * ga) if ga else float("nan"),
                                                       tf, tf / 157.3))
