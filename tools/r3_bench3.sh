#!/bin/bash
# round 3, after gs_preprocess_bwd_views: the multi-view legs of tools/r3_bench2.sh again, all GPU tests before them
set -o pipefail
mkdir -p gpurun_out/views
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread > gpurun_out/views/full.log 2>&1 || { tail -n 40 gpurun_out/views/full.log; exit 1; }
tail -n 2 gpurun_out/views/full.log
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-next-rows --no-cpu-baseline --no-exact-mode --config config4_8views_6M_sh3_1297x840 > gpurun_out/views/config4_8views.json 2> gpurun_out/views/config4_8views.err && \
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-next-rows --no-cpu-baseline --no-exact-mode --views-per-step 8 > gpurun_out/views/config3_8views.json 2> gpurun_out/views/config3_8views.err && \
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-next-rows --no-cpu-baseline --no-exact-mode --force-dist --parallel allreduce --views-per-step 4 > gpurun_out/views/forcedist_4views.json 2> gpurun_out/views/forcedist_4views.err && \
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-next-rows --no-cpu-baseline --no-exact-mode --force-dist > gpurun_out/views/forcedist_auto.json 2> gpurun_out/views/forcedist_auto.err
echo rc=$?
for f in gpurun_out/views/*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d['ms_per_step'], d['value'], d['config'].get('views_per_step'), d.get('parallel_probe_ms_per_step'))
PY
done
