#!/usr/bin/env python
"""Post-process a rocprofv3 kernel trace of bench.py: isolate ONE hipGraph-replayed training step (the dispatches
between the last two k_adam launches) and print, per kernel name, launches / total / average duration, plus the
idle time between consecutive kernels (launch gaps)."""
import csv, sys, collections, re

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
# round 6 (per-network pipeline): the optimizer is one k_adam per network INSIDE the step, and the step begins with a stand-alone
# k_adam_tick (cc_adam_tick) -- a step = the dispatches from one such tick up to the next.  Older traces: the dispatches between the
# ends of two optimizer launches (k_adam_tick + k_adam back to back behind the step).
ticks = [i for i, r in enumerate(rows) if "k_adam_tick" in r[2] and not (i + 1 < len(rows) and "k_adam(" in rows[i + 1][2].replace("k_adam_tick", ""))]
if len(ticks) >= 3:
    adam = [i - 1 for i in ticks]           # (last dispatch of the previous step)
    cands = [rows[ticks[i - 1]: ticks[i]] for i in range(max(1, len(ticks) - 3), len(ticks))]
else:
    adam = [i for i, r in enumerate(rows) if "k_adam" in r[2]]
    adam = [i for j, i in enumerate(adam) if j + 1 == len(adam) or adam[j + 1] != i + 1]     # the optimizer is 2 back-to-back launches
    assert len(adam) >= 2, "need two optimizer launches"
    cands = [rows[adam[i - 1] + 1: adam[i] + 1] for i in range(max(1, len(adam) - 3), len(adam))]
# of the last (up to) three replayed steps, the one with the shortest wall time: the profiler's buffer flushes now and then stall
# a step for milliseconds in the middle of the graph.  (The host copies of the next batch that precede a replay are dropped.)
cands = [[r for r in st if "copyBuffer" not in r[2] or r[0] > st[0][0] + 20000] for st in cands]
step = min(cands, key=lambda st: st[-1][1] - st[0][0])
# step period: end of the optimizer launch to the end of the next one (includes whatever idles BETWEEN two replays)
ends = [rows[i][1] for i in adam]
periods = [(b - a) / 1e6 for a, b in zip(ends[:-1], ends[1:])][-4:]
if periods:
    print("step period (optimizer end to optimizer end), last %d steps: %s ms" % (len(periods), " ".join("%.3f" % p for p in periods)))
t0, t1 = step[0][0], step[-1][1]
busy = sum(e - s for s, e, *_ in step)
# with the networks on streams of their own (round 5) kernels overlap: time covered by at least one kernel, and by two or more
covered, two, cur_end, sec_end = 0, 0, None, None
ev = sorted([(s_, 1) for s_, e_, *_ in step] + [(e_, -1) for s_, e_, *_ in step])
depth, last = 0, ev[0][0]
for t_, d_ in ev:
    if depth >= 1:
        covered += t_ - last
    if depth >= 2:
        two += t_ - last
    depth += d_
    last = t_
gaps = (step[-1][1] - step[0][0]) - covered
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n, _q in step:
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = n.split("(")[0][:70]
    agg[n][0] += 1
    agg[n][1] += e - s
print("one replayed step: %d kernels, wall %.3f ms, sum of kernel durations %.3f ms, covered by >= 1 kernel %.3f ms (by >= 2: %.3f ms), gaps %.3f ms"
      % (len(step), (t1 - t0) / 1e6, busy / 1e6, covered / 1e6, two / 1e6, gaps / 1e6))
print("%-72s %6s %10s %9s %6s" % ("kernel", "calls", "total_us", "avg_us", "%"))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-72s %6d %10.1f %9.2f %6.2f" % (n, c, t / 1e3, t / 1e3 / c, 100.0 * t / (t1 - t0)))

if len(sys.argv) > 2 and sys.argv[2] == "--seq":
    # the step as a sequence: start offset, duration, gap to the previous kernel's end, name (chains of one stream read top to bottom)
    print("\nsequence (us from the step's first dispatch):")
    prev = step[0][0]
    for s_, e_, n, q_ in step:
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"^void ", "", n).split("(")[0][:60]
        print("%9.1f %8.1f %7.1f  q%-3s %s" % ((s_ - t0) / 1e3, (e_ - s_) / 1e3, (s_ - prev) / 1e3, q_, n))
        prev = max(prev, e_)

if len(sys.argv) > 2 and sys.argv[2] in ("--queues", "--seq"):
    # per hardware queue (= branch of the replayed graph): kernels, busy time, first start / last end, the last three kernels --
    # which chain finishes last (round 6: DispResNet6's, with its Adam segment and weight images at the very end)
    print("\nper hardware queue of the replayed step (us from its first dispatch):")
    byq = collections.defaultdict(list)
    for s_, e_, n, q_ in step:
        byq[q_].append((s_, e_, re.sub(r"^void ", "", re.sub(r"\(anonymous namespace\)::", "", n)).split("(")[0][:40]))
    for q_, ks in sorted(byq.items(), key=lambda kv: kv[1][-1][1]):
        print("queue %-3s %4d kernels  busy %8.1f  first %8.1f  last end %8.1f   ...%s" % (
            q_, len(ks), sum(e_ - s_ for s_, e_, _ in ks) / 1e3, (ks[0][0] - t0) / 1e3, (max(e_ for _, e_, _ in ks) - t0) / 1e3,
            " | ".join(n for _, _, n in ks[-3:])))
