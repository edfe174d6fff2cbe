#!/bin/bash
# The shader clock and the phase split INSIDE wgrad_h3b_kernel: a diagnostic build (-DADDK_WG_DIAG, into /tmp) under the stand-alone harness.
#   bash scripts/wgrad_phases.sh > gpurun_out/wgrad_phases.txt
set -e
SRC=$GRAFT_REPO_ROOT/auto-dynamic-deeplab_amd/csrc
D=/tmp/addk_wgdiag; rm -rf $D; mkdir -p $D/pkg/csrc $D/include
cp $SRC/*.hip $SRC/*.h $SRC/*.cpp $SRC/Makefile $SRC/*.o $D/pkg/csrc/
cp $GRAFT_REPO_ROOT/include/addk.h $D/include/
rm -f $D/pkg/csrc/wgrad.o
(cd $D/pkg/csrc && make -j16 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-result -ffp-contract=off -DADDK_WG_DIAG ${WG_EXTRA_DEFINES}" 2>&1 | tail -3)
cd $GRAFT_REPO_ROOT
hipcc -O2 --offload-arch=gfx950 -Iinclude scripts/conv_bench.cpp -L$D/pkg -laddk -ldl -Wl,-rpath,$D/pkg -o /tmp/conv_bench_wgdiag 2>&1 | grep -v warning | tail -5
for sh in "decoder 3x3 304" "decoder 3x3 256" "aspp 3x3" "stem 3x3"; do SHAPES="$sh" ADDK_MATH=bf16x6 /tmp/conv_bench_wgdiag 20 2>&1 | grep -v amdgpu.ids | grep -E -A1 "wgrad|inside split"; done
