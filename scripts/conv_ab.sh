#!/bin/bash
# A/B of two builds of the library on ONE box: build/classic/libaddk.so (A) against the in-tree one (B), ABAB, stand-alone conv launches.
cd $(dirname $0)/..
hipcc -O2 --offload-arch=gfx950 -Iinclude scripts/conv_bench.cpp -Lbuild/classic -laddk -ldl -Wl,-rpath,$PWD/build/classic -o /tmp/conv_bench_A 2>/dev/null || exit 1
hipcc -O2 --offload-arch=gfx950 -Iinclude scripts/conv_bench.cpp -Lauto-dynamic-deeplab_amd -laddk -ldl -Wl,-rpath,$PWD/auto-dynamic-deeplab_amd -o /tmp/conv_bench_B 2>/dev/null || exit 1
for sh in "stem 3x3" "dil 5x5 d2 40->40" "dil 3x3 d2 40->40" "dil 5x5 d2 80->80 @63x127" "dil 3x3 d2 80->80" "dil 5x5 d2 160" "dil 3x3 d2 160" "stem2" "decoder 3x3 256" "decoder 3x3 304" "aspp 3x3" "aspp 1x1"; do
  for rep in 1 2; do for v in A B; do
    SHAPES="$sh" PACKED=1 NOWGRAD=1 ADDK_MATH=bf16x6 /tmp/conv_bench_$v 30 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"
  done; done
done
