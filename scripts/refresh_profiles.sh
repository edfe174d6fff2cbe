#!/bin/bash
# Regenerate the committed measurement artefacts under gpurun_out/ (copy the ones to keep into profiles/):
#   bench line (with cpu_baseline), rocprofv3 kernel stats + last-step summary of the same command, segment roofline,
#   inference latencies.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > gpurun_out/bench_fp32.json 2> gpurun_out/bench_fp32.err || exit 1
tail -c 600 gpurun_out/bench_fp32.json
rm -rf gpurun_out/pt; mkdir -p gpurun_out/pt
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pt -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1 || exit 1
f=$(ls gpurun_out/pt/*/*kernel_trace.csv | head -1)
python3 scripts/trace_summary.py $f 80 > gpurun_out/step_last_step_summary.txt
cp $(ls gpurun_out/pt/*/*kernel_stats.csv | head -1) gpurun_out/step_kernel_stats.csv
rm -rf gpurun_out/pt
timeout -k 10 200 python3 scripts/segment_roofline.py --mode eval > gpurun_out/segment_roofline_eval.json 2>/dev/null
timeout -k 10 200 python3 scripts/segment_roofline.py --mode train > gpurun_out/segment_roofline_train.json 2>/dev/null
timeout -k 10 300 python3 scripts/bench_infer.py > gpurun_out/infer_1024x2048.json 2>/dev/null
timeout -k 10 300 python3 tests/tools/time_oracle_infer.py > gpurun_out/infer_cpu_oracle.json 2>/dev/null
head -3 gpurun_out/step_last_step_summary.txt
python3 -c "
import json
for m in ('eval','train'):
    d=json.load(open('gpurun_out/segment_roofline_%s.json'%m)); print(m, d['forward_ms_total'], d['aspp_plus_cell'])
print(open('gpurun_out/infer_1024x2048.json').read()[:600])
"
