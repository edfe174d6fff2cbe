#!/bin/bash
# SQ counters per kernel over the last training step (separate from the kernel-trace/stats run), summarised by scripts/pmc_summary.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_step; mkdir -p gpurun_out/pmc_step
timeout -k 10 500 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d gpurun_out/pmc_step -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph > gpurun_out/pmc_step.log 2>&1 || { tail -3 gpurun_out/pmc_step.log; exit 1; }
f=$(ls gpurun_out/pmc_step/*/*counter_collection.csv | head -1)
python3 scripts/pmc_summary.py $f 40 > gpurun_out/pmc_sq_summary.txt
rm -rf gpurun_out/pmc_step
head -30 gpurun_out/pmc_sq_summary.txt
