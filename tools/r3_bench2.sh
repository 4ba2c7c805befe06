#!/bin/bash
# round 3: bench.py's new legs -- default line, config 4's 8-view step on one GPU, the all-reduce step with 4 views at world 1 (forced collectives)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --steps 50 --warmup 5 --no-next-rows --no-cpu-baseline > gpurun_out/r3b_default.json 2> gpurun_out/r3b_default.err && \
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-next-rows --no-cpu-baseline --no-exact-mode --config config4_8views_6M_sh3_1297x840 > gpurun_out/r3b_config4_8views.json 2> gpurun_out/r3b_config4_8views.err && \
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-next-rows --no-cpu-baseline --no-exact-mode --views-per-step 8 > gpurun_out/r3b_config3_8views.json 2> gpurun_out/r3b_config3_8views.err && \
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-next-rows --no-cpu-baseline --no-exact-mode --force-dist --parallel allreduce --views-per-step 4 > gpurun_out/r3b_forcedist_4views.json 2> gpurun_out/r3b_forcedist_4views.err && \
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-next-rows --no-cpu-baseline --no-exact-mode --force-dist > gpurun_out/r3b_forcedist_auto.json 2> gpurun_out/r3b_forcedist_auto.err
echo rc=$?
for f in gpurun_out/r3b_*.err; do echo "== $f"; tail -n 5 $f; done
