#!/bin/bash
# Hardware counters of config 4's 8-view step (the multi-view per-Gaussian kernels), one rocprofv3 run per counter group.
# usage (GPU box): bash tools/pmc_views.sh TAG   ->  gpurun_out/TAG_views_pmc_summary.txt
set -e
tag=${1:-run}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out
BENCH="python3 bench.py --config config4_8views_6M_sh3_1297x840 --steps 3 --warmup 1 --no-cpu-baseline --no-exact-mode --no-next-rows"
i=0
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $P --output-format csv -d $out/${tag}_views_pmc$i -- $BENCH > /dev/null 2> $out/${tag}_views_pmc$i.err
  echo pass $i done
done
python3 tools/pmc_summary.py $out/${tag}_views_pmc*/*/*counter_collection.csv > $out/${tag}_views_pmc_summary.txt
grep -E "views_kernel|render_bwd" $out/${tag}_views_pmc_summary.txt | cut -c1-330
