#!/bin/bash
# per-launch durations of the stand-alone harness for one shape: SHAPES="stem2" bash scripts/kernel_trace.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
hipcc -O2 --offload-arch=gfx950 -Iinclude scripts/conv_bench.cpp -Lauto-dynamic-deeplab_amd -laddk -Wl,-rpath,$PWD/auto-dynamic-deeplab_amd -o /tmp/conv_bench 2>/dev/null || exit 1
export SHAPES="${SHAPES:-stem2}"
/tmp/conv_bench 10
rm -rf gpurun_out/kt; timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- /tmp/conv_bench 2 > /dev/null 2>&1 || exit 1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/kt/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
for r in rows[-40:]:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:60]
    print('%-62s %8.1f us  grid %s' % (n, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('Grid_Size_X', r.get('Grid_Size', ''))))
PY
rm -rf gpurun_out/kt
