#!/bin/bash
# SQ counters of the small cell-sized launches on the stand-alone harness (wave lifetime, stall split)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
hipcc -O2 --offload-arch=gfx950 -Iinclude scripts/conv_bench.cpp -Lauto-dynamic-deeplab_amd -laddk -Wl,-rpath,$PWD/auto-dynamic-deeplab_amd -o /tmp/conv_bench 2>/dev/null || exit 1
export SHAPES="${SHAPES:-cell 1x1 80}"
rm -rf gpurun_out/pmc_s
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d gpurun_out/pmc_s -- /tmp/conv_bench 2 > /dev/null 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_s/*/*counter_collection.csv')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:36]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    if (r['Dispatch_Id'], k) not in seen: seen.add((r['Dispatch_Id'], k)); cnt[k] += 1
for k, v in agg.items():
    n = cnt[k]; w = v['SQ_WAVES'] / n
    if w == 0: continue
    print('%-38s n=%d waves=%6.0f  wave-life=%7.0f cyc  busy=%7.0f  active=%4.1f%% wait_any=%4.1f%% wait_inst=%4.1f%%  valu/wave=%5.0f salu/wave=%5.0f' % (
        k, n, w, 4 * v['SQ_WAVE_CYCLES'] / v['SQ_WAVES'], v['SQ_BUSY_CYCLES'] / n, 100 * v['SQ_ACTIVE_INST_ANY'] / v['SQ_WAVE_CYCLES'],
        100 * v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES'], 100 * v['SQ_WAIT_INST_ANY'] / v['SQ_WAVE_CYCLES'], v['SQ_INSTS_VALU'] / v['SQ_WAVES'], v['SQ_INSTS_SALU'] / v['SQ_WAVES']))
PY
rm -rf gpurun_out/pmc_s
