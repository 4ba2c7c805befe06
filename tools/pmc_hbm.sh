# HBM traffic counters per kernel (each pass its own run; TCC block holds 4 counters, FETCH_SIZE uses 3, WRITE_SIZE 2)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for P in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  tag=$(echo $P | tr ' ' '_')
  timeout -s KILL 240 rocprofv3 --pmc $P --output-format csv -d gpurun_out/hbm_$tag -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-mode > /dev/null 2> gpurun_out/hbm_$tag.err
done
python3 tools/pmc_summary.py gpurun_out/hbm_*/*/*counter_collection.csv > gpurun_out/hbm_summary.txt
