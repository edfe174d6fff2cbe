#!/bin/bash
# [r5] split-fp16 bring-up, stage 2: weight gradients + whole-network gates + step time
cd $(dirname $0)/..
mkdir -p gpurun_out
rm -f gpurun_out/halo_kernel_errors.txt
timeout -k 10 900 python -m pytest tests/test_gpu_fast_kernels.py tests/test_gpu_round5.py tests/test_gpu_round4.py -x -q -m gpu > gpurun_out/f16_stage2_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/f16_stage2_pytest.log
tail -5 gpurun_out/f16_stage2_pytest.log
timeout -k 10 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -x -q -m gpu -k "x3" > gpurun_out/f16_stage2_net.log 2>&1
echo "pytest rc $?" >> gpurun_out/f16_stage2_net.log
tail -5 gpurun_out/f16_stage2_net.log
for m in bf16x3 bf16x6; do
  timeout -k 10 600 python bench.py --math $m --steps 20 --warmup 3 --no-extras > gpurun_out/f16_stage2_bench_$m.json 2> gpurun_out/f16_stage2_bench_$m.err || echo "bench $m rc $?"
  python - <<P
import json
try:
    d=json.loads(open('gpurun_out/f16_stage2_bench_$m.json').read().strip().split('\n')[-1])
    print('$m', d['ms_per_step'], d.get('first_step_losses'), d.get('first_step_loss_vs_cpu_oracle'), d['roofline']['launch_ms'])
except Exception as e: print('$m parse failed', e)
P
done
