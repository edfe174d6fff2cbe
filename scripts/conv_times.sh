#!/bin/bash
# Stand-alone times of the narrow / stem convolution launches on the PRODUCT library (forward + data gradient, weights packed ahead as in a plan).
#   scripts/conv_times.sh [OUTFILE]
cd $(dirname $0)/..
hipcc -O2 --offload-arch=gfx950 -Iinclude scripts/conv_bench.cpp -Lauto-dynamic-deeplab_amd -laddk -ldl -Wl,-rpath,$PWD/auto-dynamic-deeplab_amd -o /tmp/conv_bench 2>/dev/null || exit 1
for sh in "stem 3x3" "dil 5x5 d2 40->40" "dil 3x3 d2 40->40" "dil 5x5 d2 80->80 @63x127" "dil 3x3 d2 80->80" "dil 5x5 d2 160" "dil 3x3 d2 160" "stem2" "decoder 3x3 256" "aspp 3x3"; do
  SHAPES="$sh" PACKED=1 NOWGRAD=1 ADDK_MATH=bf16x6 /tmp/conv_bench 30 2>&1 | grep -v amdgpu.ids
done | tee ${1:-/dev/null}
