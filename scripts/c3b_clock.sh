#!/bin/bash
# The shader clock INSIDE conv3b_kernel: a diagnostic build of the library (-DADDK_C3B_DIAG: every workgroup
# adds its s_memtime / s_memrealtime lifetime to a device counter) under the stand-alone harness, on the scratch copy of the GPU box.
#   bash scripts/c3b_clock.sh > gpurun_out/r03_c3b_in_kernel_clock.txt
cd $GRAFT_REPO_ROOT/auto-dynamic-deeplab_amd/csrc || exit 1
rm -f conv3.o
make -j16 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-result -ffp-contract=off -DADDK_C3B_DIAG" > /dev/null 2>&1 || exit 1
cd $GRAFT_REPO_ROOT
hipcc -O2 --offload-arch=gfx950 -Iinclude scripts/conv_bench.cpp -Lauto-dynamic-deeplab_amd -laddk -ldl -Wl,-rpath,$PWD/auto-dynamic-deeplab_amd -o /tmp/conv_bench_diag 2>/dev/null || exit 1
for sh in "decoder 3x3 304" "decoder 3x3 256" "aspp 3x3" "stem 3x3" "stem2"; do SHAPES="$sh" ADDK_MATH=bf16x6 /tmp/conv_bench_diag 20 2>&1 | grep -v amdgpu.ids; done
