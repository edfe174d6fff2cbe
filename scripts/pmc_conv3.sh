#!/bin/bash
# HBM-side traffic of the decoder 3x3 conv (halo-patch kernel) from rocprofv3 PMC passes: FETCH_SIZE and WRITE_SIZE in
# SEPARATE runs (MI355X_MICROARCH.md, HBM section), on the stand-alone harness (scripts/conv_bench.cpp).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
hipcc -O2 --offload-arch=gfx950 -Iinclude scripts/conv_bench.cpp -Lauto-dynamic-deeplab_amd -laddk -Wl,-rpath,$PWD/auto-dynamic-deeplab_amd -o /tmp/conv_bench 2>/dev/null || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- /tmp/conv_bench 2 > /dev/null 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob, json, collections
out = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob('gpurun_out/pmc_%s/*/*counter_collection.csv' % c)[0]
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == c:
            per[r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]].append(float(r['Counter_Value']))
    out[c] = {k: v for k, v in per.items()}
res = {}
for k in out['FETCH_SIZE']:
    fs, ws = out['FETCH_SIZE'][k], out['WRITE_SIZE'].get(k, [])
    res[k] = {'launches': len(fs), 'FETCH_SIZE_KB_first_shape': fs[0], 'WRITE_SIZE_KB_first_shape': ws[0] if ws else None,
              'FETCH_SIZE_KB_all': fs[:12], 'WRITE_SIZE_KB_all': ws[:12]}
json.dump(res, open('gpurun_out/pmc_conv3_raw.json', 'w'), indent=1)
for k, v in res.items():
    print(k, v['launches'], v['FETCH_SIZE_KB_first_shape'], v['WRITE_SIZE_KB_first_shape'])
PY
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
