#!/bin/bash
# one-step trace of the forced-communication (two-graph) step: where do the +0.8 ms against the single-graph step go?
mkdir -p gpurun_out
export TMPDIR=/tmp
( cd /tmp && CC_FORCE_COMM=1 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_r3l -o r3l -- python /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing ) > gpurun_out/rocprof_r3l.log 2>&1; echo "rocprof rc=$?"
T=$(find gpurun_out/prof_r3l -name "*kernel_trace.csv" | head -1)
python tools/step_trace.py "$T" > gpurun_out/step_trace_r3l.txt 2>&1; head -3 gpurun_out/step_trace_r3l.txt
python - "$T" <<'PY'
import csv, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in csv.DictReader(open(sys.argv[1]))))
# largest gaps between consecutive kernels in the last ~1100 kernels
rows = rows[-1200:]
gaps = sorted(((rows[i + 1][0] - rows[i][1], rows[i][2], rows[i + 1][2]) for i in range(len(rows) - 1)), reverse=True)[:12]
for g, a, b in gaps:
    print("%8.1f us  after %-50s before %s" % (g / 1e3, a, b))
PY
find gpurun_out/prof_r3l -name "*kernel_trace.csv" -size +20M -delete
