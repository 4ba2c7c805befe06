#!/bin/bash
# round 3, first GPU call: fuzz at the default stop threshold (oracle at the same threshold), fuzz at a 2^-26 floor against the
# never-stopping oracle, baseline bench, bench at the 2^-26 floor
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -u tools/fuzz_parity.py 0 200 1e-5 > gpurun_out/r3_fuzz_default.log 2>&1; echo "fuzz default rc=$?" && \
timeout -k 10 400 python -u tools/fuzz_parity.py 0 200 1.4901161193847656e-08 0 > gpurun_out/r3_fuzz_floor26.log 2>&1; echo "fuzz floor rc=$?" && \
timeout -k 10 300 python bench.py --steps 50 --warmup 5 --no-next-rows --no-cpu-baseline > gpurun_out/r3_bench_base.json 2> gpurun_out/r3_bench_base.err && \
timeout -k 10 300 python bench.py --steps 50 --warmup 5 --no-next-rows --no-cpu-baseline --no-exact-mode --t-eps 1.4901161193847656e-08 > gpurun_out/r3_bench_floor26.json 2> gpurun_out/r3_bench_floor26.err
tail -3 gpurun_out/r3_fuzz_default.log gpurun_out/r3_fuzz_floor26.log
