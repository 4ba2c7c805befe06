set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU"
timeout -s KILL 120 rocprofv3 --pmc $P1 --output-format csv -d gpurun_out/pmc1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-mode > /dev/null 2> gpurun_out/pmc1.err
timeout -s KILL 120 rocprofv3 --pmc $P2 --output-format csv -d gpurun_out/pmc2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-mode > /dev/null 2> gpurun_out/pmc2.err
ls gpurun_out/pmc1/* gpurun_out/pmc2/*
