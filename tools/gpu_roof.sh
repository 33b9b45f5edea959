timeout 300 python bench.py --no-cpu-baseline 2>/dev/null > gpurun_out/bench_roof.log; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/bench_roof.log') if l.startswith('{')][0])
print(d["value"], d["ms_per_step"]); r=d["roofline"]
print({k:v for k,v in r.items() if k not in ("by_kernel","by_call_group","timing","traffic_source")})
print(r["by_kernel"])
PY
