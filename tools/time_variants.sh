#!/bin/bash
# usage: [BENCH_ARGS='--config ...'] [STEPS=n] tools/time_variants.sh NAME...   ("" or base = the in-tree library) -> frame ms and render_bwd / render_fwd stage ms per variant
for v in "$@"; do
  if [ "$v" = base ]; then unset GSPLAT_LIB; else export GSPLAT_LIB=$PWD/gaussian-splatting_amd/lib/var_$v.so; fi
  timeout -k 10 150 python bench.py --no-next-rows --steps ${STEPS:-30} --warmup 5 --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']; print('$v', d['ms_per_step'], 'exact', (d.get('exact_mode') or {}).get('ms_per_step'), ' '.join(f'{k}={v:.4f}' for k, v in s.items()))" || exit 1
done
