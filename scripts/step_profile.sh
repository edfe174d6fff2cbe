#!/bin/bash
# kernel-by-kernel summary of one training step (single stream = true durations): scripts/step_profile.sh OUTDIR [streams]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/sp}; S=${2:-1}; mkdir -p $O; rm -rf $O/pt; mkdir -p $O/pt
ADDK_STREAMS=$S timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pt -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_line.json 2>/dev/null
python3 scripts/trace_summary.py $(ls $O/pt/*/*kernel_trace.csv | head -1) 90 > $O/step_last_step_summary_${S}stream.txt
rm -rf $O/pt
head -45 $O/step_last_step_summary_${S}stream.txt
