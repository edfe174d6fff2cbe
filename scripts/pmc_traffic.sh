#!/bin/bash
# HBM-side traffic (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate --pmc passes) and SQ counters of the dominant launch on the stand-alone harness:
#   ADDK_MATH=f16x3 bash scripts/pmc_traffic.sh OUTDIR
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/pmc}; mkdir -p $O
export ADDK_MATH=${ADDK_MATH:-f16x3} SHAPES="decoder 3x3 304"
bash scripts/pmc_conv3b.sh > $O/pmc_conv3b_$ADDK_MATH.txt 2>&1
grep -A5 "conv3b_kernel<4, 3, 0" $O/pmc_conv3b_$ADDK_MATH.txt | head -8
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$c
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- /tmp/conv_bench 2 > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, json, os
O = '$O'; M = os.environ['ADDK_MATH']
res = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob('%s/pmc_%s/*/*counter_collection.csv' % (O, c))
    if not f: continue
    for r in csv.DictReader(open(f[0])):
        if r['Counter_Name'] == c:
            k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
            res.setdefault(k, {}).setdefault(c, []).append(float(r['Counter_Value']))
out = {}
for k, v in res.items():
    if 'conv3b_kernel<4, 3, 0' in k:
        fs, ws = v.get('FETCH_SIZE', [0])[0], v.get('WRITE_SIZE', [0])[0]
        out = {'kernel': k, 'shape': 'decoder 3x3 304->256 @ [2,128,256], ' + M, 'FETCH_SIZE_KB_raw': fs, 'WRITE_SIZE_KB': ws,
               'fetch_bytes_corrected_x2_gfx950': fs * 1024 * 2, 'write_bytes': ws * 1024, 'traffic_bytes_per_launch': fs * 1024 * 2 + ws * 1024,
               'algorithmic_bytes_per_launch': 149602304.0}
json.dump(out, open(O + '/pmc_traffic_decoder_conv3b_%s.json' % M, 'w'), indent=1)
print(out)
PY
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
