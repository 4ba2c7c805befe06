#!/bin/bash
# Hardware counters and the kernel trace of the bench command (config 3, default t_eps), one rocprofv3 run per counter
# group (a --pmc run holds at most 8 SQ / 4 TCC counters; never combined with trace domains other than --kernel-trace).
# Run on the GPU box:  gpurun --timeout 900 -- 'bash tools/pmc_collect.sh r2x'
# Leaves gpurun_out/<tag>_*; tools/make_counters.py turns them into profiles/counters.json + the committed summaries.
set -e
tag=${1:-run}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out
BENCH="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-exact-mode --no-next-rows"
# the trace runs more steps than the counter passes: the first launches of a process are slower (render_bwd 0.70 -> 0.65 ms over
# its first ten launches), and the per-kernel mean of this trace is what bench.py's event time is compared with
TRACE_BENCH="python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-exact-mode --no-next-rows"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace -- $TRACE_BENCH > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_trace.err
echo trace done
i=0
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $P --output-format csv -d $out/${tag}_pmc$i -- $BENCH > /dev/null 2> $out/${tag}_pmc$i.err
  echo pass $i done
done
python3 tools/pmc_summary.py $out/${tag}_pmc*/*/*counter_collection.csv > $out/${tag}_pmc_summary.txt
cp $(ls $out/${tag}_trace/*/*kernel_stats.csv | head -1) $out/${tag}_kernel_stats.csv
python3 bench.py --steps 100 --warmup 10 > $out/${tag}_bench.json 2> $out/${tag}_bench.err
echo all done
