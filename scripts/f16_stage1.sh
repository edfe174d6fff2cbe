#!/bin/bash
# [r5] split-fp16 ("f16x3") bring-up: kernel parity against fp64 and stand-alone times beside bf16x6 on one box
cd $(dirname $0)/..
mkdir -p gpurun_out
rm -f gpurun_out/halo_kernel_errors.txt
timeout -k 10 500 python -m pytest tests/test_gpu_fast_kernels.py tests/test_gpu_round5.py -x -q -m gpu -k "halo_patch or stride2_conv or conv3n or narrow" > gpurun_out/f16_stage1_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/f16_stage1_pytest.log
tail -15 gpurun_out/f16_stage1_pytest.log
hipcc -O2 --offload-arch=gfx950 -Iinclude scripts/conv_bench.cpp -Lauto-dynamic-deeplab_amd -laddk -ldl -Wl,-rpath,$PWD/auto-dynamic-deeplab_amd -o /tmp/conv_bench 2>/dev/null || exit 1
for m in bf16x6 bf16x3; do
for sh in "stem 3x3" "dil 5x5 d2 40->40" "dil 3x3 d2 40->40" "dil 5x5 d2 80->80 @63x127" "stem2" "decoder 3x3 256" "aspp 3x3"; do
  echo "== $m"; SHAPES="$sh" PACKED=1 NOWGRAD=1 ADDK_MATH=$m timeout -k 10 120 /tmp/conv_bench 30 2>&1 | grep -v amdgpu.ids
done; done > gpurun_out/f16_stage1_times.txt
cat gpurun_out/f16_stage1_times.txt
