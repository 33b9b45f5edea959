#!/bin/bash
# what the driver does at round end, on one box: GPU tests, smoke(), default bench line
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -x -q -m gpu ) > gpurun_out/driver_pytest.log 2>&1; echo "pytest rc=$?"; tail -1 gpurun_out/driver_pytest.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > gpurun_out/driver_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/driver_smoke.log
( timeout 420 python bench.py --gpus 1 --steps 20 --warmup 3 ) > gpurun_out/driver_bench.log 2> gpurun_out/driver_bench.err; echo "bench rc=$?"
python -c "
import json
for l in open('gpurun_out/driver_bench.log'):
    if l.startswith('{'):
        d=json.loads(l); print({k:d[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','scaling','vs_baseline','dtype','data')}); print(d['config']); r=d['roofline']; print({k:r[k] for k in ('bound','kernel','achieved','peak','unit','frac','traffic')}); print(d['cpu_baseline']); print(d.get('parity'))
"
