#!/bin/bash
# the multi-view backward (gs_preprocess_bwd_views): its parity tests, the multi-view step tests, then config 4's 8-view step
set -e
mkdir -p gpurun_out/views
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread -k "preprocess_bwd" > gpurun_out/views/t1.log 2>&1 || { tail -n 40 gpurun_out/views/t1.log; exit 1; }
tail -n 2 gpurun_out/views/t1.log
timeout -k 10 500 python -u -m pytest tests/test_gpu_allreduce.py tests/test_gpu_parity.py -x -q --timeout 400 --timeout-method thread -k "allreduce or config4 or view" > gpurun_out/views/t2.log 2>&1 || { tail -n 40 gpurun_out/views/t2.log; exit 1; }
tail -n 2 gpurun_out/views/t2.log
timeout -k 10 300 python bench.py --config config4_8views_6M_sh3_1297x840 --views-per-step 8 --steps 5 --warmup 2 --no-cpu-baseline --no-next-rows > gpurun_out/views/bench_c4_8v.json 2> gpurun_out/views/bench_c4_8v.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/views/bench_c4_8v.json').read().strip().splitlines()[-1])
print('config4 8 views', d['ms_per_step'], d['value'], {k:round(v,4) for k,v in d['stage_ms'].items()})
PY
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-next-rows > gpurun_out/views/bench_c3.json 2> gpurun_out/views/bench_c3.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/views/bench_c3.json').read().strip().splitlines()[-1])
print('config3', d['ms_per_step'], {k:round(v,4) for k,v in d['stage_ms'].items()})
PY
