#!/bin/bash
set -o pipefail
timeout -k 10 500 python -u -m pytest tests/test_gpu_parity.py tests/test_w_equals_one.py -m gpu -x -q --timeout 400 --timeout-method thread -k "golden or midsize or early_stop or twice or bands or w_equals or config3" > gpurun_out/r3_flush_tests.log 2>&1; echo tests rc=$?; tail -n 8 gpurun_out/r3_flush_tests.log | cut -c1-300
STEPS=40 bash tools/time_variants.sh base flush1 base flush1
