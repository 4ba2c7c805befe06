#!/bin/bash
# usage: tools/r3_final.sh TAG -- all GPU tests, smoke, the multi-view bench legs, counters + kernel stats of the default bench (tools/pmc_collect.sh TAG), the default bench line
set -o pipefail
tag=$1
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread > $out/pytest.log 2>&1 || { tail -n 40 $out/pytest.log; exit 1; }
tail -n 2 $out/pytest.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1 || { tail -n 20 $out/smoke.log; exit 1; }
tail -n 1 $out/smoke.log
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-next-rows --no-cpu-baseline --no-exact-mode --config config4_8views_6M_sh3_1297x840 > $out/config4_8views.json 2> $out/config4_8views.err && \
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-next-rows --no-cpu-baseline --no-exact-mode --views-per-step 8 > $out/config3_8views.json 2> $out/config3_8views.err && \
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-next-rows --no-cpu-baseline --no-exact-mode --force-dist --parallel allreduce --views-per-step 4 > $out/forcedist_4views.json 2> $out/forcedist_4views.err && \
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-next-rows --no-cpu-baseline --no-exact-mode --force-dist > $out/forcedist_auto.json 2> $out/forcedist_auto.err && \
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-next-rows --no-cpu-baseline --no-exact-mode --force-dist --parallel exchange > $out/forcedist_exchange.json 2> $out/forcedist_exchange.err || { echo "a bench leg failed"; tail -n 5 $out/*.err; exit 1; }
for f in $out/config4_8views $out/config3_8views $out/forcedist_4views $out/forcedist_auto $out/forcedist_exchange; do python - "$f.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d['ms_per_step'], d['value'], d['config'].get('views_per_step'), d.get('parallel_probe_ms_per_step'))
PY
done
tools/pmc_collect.sh $tag || exit 1
python3 tools/make_counters.py $tag > $out/make_counters.log 2>&1 || { tail -n 5 $out/make_counters.log; exit 1; }   # on the box: so that the bench below reports the traffic; run it again in the build container
timeout -k 10 400 python bench.py > $out/bench_default.json 2> $out/bench_default.err || { tail -n 5 $out/bench_default.err; exit 1; }
tail -c 600 $out/bench_default.json
