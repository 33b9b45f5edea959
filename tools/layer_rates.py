#!/usr/bin/env python
"""Per-layer-shape device-kernel rates of one eager step.

    CC_TIMING_DETAIL=1 CC_TIMING_DUMP=gpurun_out/layers.tsv python bench.py --steps 5 --warmup 3
    python tools/layer_rates.py gpurun_out/layers.tsv

The tools build of the library (tools/_bin/libccengine_tools.so) brackets the main device kernel of every conv / data-gradient /
weight-gradient call with HIP events; with CC_TIMING_DETAIL=1 the record name carries the layer geometry, the split-K factor (k)
and the number of workgroups (wg).  This prints the records by time, with the rate against the 157.3 TFLOP/s fp32 MFMA peak and
the time the same work would take at 100 TFLOP/s (what the best layers reach), i.e. where the convolution time above the
best-case rate sits."""
import sys, collections

PEAK = 157.3
rows = []
for ln in open(sys.argv[1]):
    ln = ln.rstrip("\n")
    if not ln:
        continue
    nm, n, ms, gf = ln.split("\t")
    rows.append((float(ms), int(n), float(gf), nm))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
totg = sum(r[2] for r in rows)
print("%d records, %.3f ms, %.1f GFLOP, %.1f TFLOP/s overall" % (len(rows), tot, totg, totg / tot))
print("%8s %3s %8s %7s %6s %8s  %s" % ("ms", "n", "GFLOP", "TF", "frac", "excess", "record"))
for ms, n, gf, nm in rows:
    tf = gf / ms if ms > 0 else 0.0
    print("%8.3f %3d %8.2f %7.1f %6.2f %8.3f  %s" % (ms, n, gf, tf, tf / PEAK, ms - gf / 100.0, nm))
fam = collections.defaultdict(lambda: [0.0, 0.0, 0])
for ms, n, gf, nm in rows:
    k = nm.split(">")[0] + ">"
    fam[k][0] += ms; fam[k][1] += gf; fam[k][2] += n
print("\nby kernel:")
for k, (ms, gf, n) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
    print("%8.3f ms %4d launches %8.1f GFLOP %6.1f TF  excess over 100 TF %.3f ms  %s" % (ms, n, gf, gf / ms, ms - gf / 100.0, k))
print("\nby rate bucket:")
for lo, hi in [(0, 10), (10, 25), (25, 40), (40, 55), (55, 70), (70, 85), (85, 1000)]:
    sel = [r for r in rows if r[0] > 0 and lo <= r[2] / r[0] < hi]
    print("%4d-%4d TF: %7.3f ms %8.1f GFLOP %4d launches" % (lo, hi, sum(r[0] for r in sel), sum(r[2] for r in sel), sum(r[1] for r in sel)))
