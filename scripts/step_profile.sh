#!/bin/bash
# rocprofv3 kernel trace of bench.py's step (default two streams and single stream = true durations) -> gpurun_out/prof_step/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_step; rm -rf $O; mkdir -p $O
for s in ${STREAMS:-2 1}; do
  rm -rf $O/pt; mkdir -p $O/pt
  ADDK_STREAMS=$s timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pt -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1 || exit 1
  cp $(ls $O/pt/*/*kernel_stats.csv | head -1) $O/step_kernel_stats_${s}stream.csv
  python3 scripts/trace_summary.py $(ls $O/pt/*/*kernel_trace.csv | head -1) 120 > $O/step_last_step_summary_${s}stream.txt
  python3 scripts/overlap.py $(ls $O/pt/*/*kernel_trace.csv | head -1) > $O/step_overlap_${s}stream.txt
  rm -rf $O/pt
done
head -2 $O/step_last_step_summary_*stream.txt
