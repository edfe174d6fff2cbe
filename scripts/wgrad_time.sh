#!/bin/bash
# Durations of the weight-gradient kernels inside the training step (single stream, rocprofv3 --kernel-trace over bench steps).   bash scripts/wgrad_time.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/pt_wg; ADDK_STREAMS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt_wg -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras > /tmp/pt_wg.log 2>&1 || { echo "bench failed"; tail -5 /tmp/pt_wg.log; }
python3 scripts/trace_summary.py $(ls /tmp/pt_wg/*/*kernel_trace.csv | head -1) 200 | grep -E "launches/step|wgrad_"
