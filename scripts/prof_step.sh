#!/bin/bash
# rocprofv3 kernel trace of a short bench run -> per-kernel summary of the last step (gpurun_out/step_summary.txt)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pt
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pt -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/prof_bench.log 2>&1
f=$(ls gpurun_out/pt/*/*kernel_trace.csv | head -1)
python3 scripts/trace_summary.py $f 60 > gpurun_out/step_summary.txt
rm -rf gpurun_out/pt
tail -1 gpurun_out/prof_bench.log | cut -c1-200
cat gpurun_out/step_summary.txt
