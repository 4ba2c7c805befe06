#!/bin/bash
# the multi-view forward (gs_preprocess_fwd_views): its parity test, the multi-view step tests, then the 8-view steps
set -e
mkdir -p gpurun_out/views
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread -k "preprocess_fwd_views or preprocess_bwd" > gpurun_out/views/t1.log 2>&1 || { tail -n 40 gpurun_out/views/t1.log; exit 1; }
tail -n 2 gpurun_out/views/t1.log
timeout -k 10 500 python -u -m pytest tests/test_gpu_allreduce.py tests/test_gpu_parity.py -x -q --timeout 400 --timeout-method thread -k "allreduce or config4 or view" > gpurun_out/views/t2.log 2>&1 || { tail -n 40 gpurun_out/views/t2.log; exit 1; }
tail -n 2 gpurun_out/views/t2.log
for c in "--config config4_8views_6M_sh3_1297x840" "--views-per-step 8"; do
timeout -k 10 300 python bench.py $c --steps 5 --warmup 2 --no-cpu-baseline --no-next-rows --no-exact-mode > gpurun_out/views/b.json 2> gpurun_out/views/b.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/views/b.json').read().strip().splitlines()[-1])
print(d['config']['workload'], d['ms_per_step'], d['value'], {k:round(v,4) for k,v in d['stage_ms'].items()})
PY
done
