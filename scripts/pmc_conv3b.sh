#!/bin/bash
# SQ counters of the halo-patch convolution on the stand-alone harness (scripts/conv_bench.cpp), one shape, chosen arithmetic:
#   SHAPES="decoder 3x3 304" ADDK_MATH=bf16x6 bash scripts/pmc_conv3b.sh
# Two passes of 8 SQ counters each (MI355X_MICROARCH.md, rocprofv3 PMC slots); no trace domains besides --kernel-trace.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
hipcc -O2 --offload-arch=gfx950 -Iinclude scripts/conv_bench.cpp -Lauto-dynamic-deeplab_amd -laddk -Wl,-rpath,$PWD/auto-dynamic-deeplab_amd -o /tmp/conv_bench 2>/dev/null || exit 1
export SHAPES="${SHAPES:-decoder 3x3 304}"
rm -rf gpurun_out/pmc_a gpurun_out/pmc_b
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_a -- /tmp/conv_bench 2 > /dev/null 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_b -- /tmp/conv_bench 2 > /dev/null 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for d in ('pmc_a', 'pmc_b'):
    f = glob.glob('gpurun_out/%s/*/*counter_collection.csv' % d)
    if not f: continue
    seen = set()
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:44]
        agg[k][d + ':' + r['Counter_Name']] += float(r['Counter_Value'])
        if d == 'pmc_a' and (r['Dispatch_Id'], k) not in seen: seen.add((r['Dispatch_Id'], k)); cnt[k] += 1
for k, v in agg.items():
    n = max(cnt[k], 1); wc = v['pmc_a:SQ_WAVE_CYCLES']
    if not wc: continue
    print(k, 'launches', n)
    print('   waves/launch %.0f  wave-life %.0f cyc  busy cycles/launch %.0f' % (v['pmc_a:SQ_WAVES'] / n, 4 * wc / v['pmc_a:SQ_WAVES'], v['pmc_a:SQ_BUSY_CYCLES'] / n))
    print('   of wave cycles: issuing %.1f%%  wait_any (s_waitcnt/barrier) %.1f%%  wait_inst (issue stall) %.1f%%' % (
        100 * v['pmc_a:SQ_ACTIVE_INST_ANY'] / wc, 100 * v['pmc_a:SQ_WAIT_ANY'] / wc, 100 * v['pmc_a:SQ_WAIT_INST_ANY'] / wc))
    print('   VALU/wave %.0f  MFMA busy cycles / (busy cycles x 4 SIMD-ish) = %.3g / %.3g' % (v['pmc_a:SQ_INSTS_VALU'] / v['pmc_a:SQ_WAVES'], v['pmc_a:SQ_VALU_MFMA_BUSY_CYCLES'], v['pmc_a:SQ_BUSY_CYCLES']))
    wb = v['pmc_b:SQ_WAVE_CYCLES'] or 1
    print('   LDS: insts/wave %.0f  bank-conflict cycles %.3g  idx-active cycles %.3g  wait_inst_lds %.1f%%  MFMA insts/wave %.0f' % (
        v['pmc_b:SQ_INSTS_LDS'] / max(v['pmc_b:SQ_WAVES'], 1), v['pmc_b:SQ_LDS_BANK_CONFLICT'], v['pmc_b:SQ_LDS_IDX_ACTIVE'], 100 * v['pmc_b:SQ_WAIT_INST_LDS'] / wb, v['pmc_b:SQ_INSTS_MFMA'] / max(v['pmc_b:SQ_WAVES'], 1)))
PY
rm -rf gpurun_out/pmc_a gpurun_out/pmc_b
