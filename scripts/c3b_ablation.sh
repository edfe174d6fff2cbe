#!/bin/bash
# Where the time of the narrow / stem launches of conv3b_kernel goes: diagnostic builds of conv3.hip with an in-kernel
# phase clock (-DADDK_C3B_DIAG) and with parts of the kernel removed (-DADDK_C3B_ABL=mask: WRONG numbers, timing only), timed shape by shape
# through the stand-alone harness scripts/conv_bench.cpp.
#   scripts/c3b_ablation.sh build            (CPU: cross-compiles build/abl/<variant>/libaddk.so, ~5 min on 8 cores)
#   scripts/c3b_ablation.sh run OUTFILE      (GPU box)
ROOT=$(cd $(dirname $0)/.. && pwd)
SRC=$ROOT/auto-dynamic-deeplab_amd/csrc
VARIANTS=${VARIANTS:-"d0: a1:1 a2:2 a4:4 a8:8 a16:16 a15:15 a31:31"}
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$ROOT/include -Wno-unused-result -ffp-contract=off"
if [ "$1" = build ]; then
  (cd $SRC && make -j8 > /dev/null) || exit 1
  for v in $VARIANTS; do
    name=${v%%:*}; mask=${v##*:}; d=$ROOT/build/abl/$name; mkdir -p $d
    ( for u in conv3 conv3b_tr3 conv3b_tr5 conv3b_row conv3b_s2 conv3n; do
        /opt/rocm/bin/hipcc $FLAGS -DADDK_C3B_DIAG ${mask:+-DADDK_C3B_ABL=$mask} -c $SRC/$u.hip -o $d/$u.o 2>> $d/build.log || exit 1
      done
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $SRC/*.o | grep -v "/conv3") $d/conv3*.o -o $d/libaddk.so && rm $d/conv3*.o && echo "built $name" ) &
    while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 2; done
  done
  wait
  exit 0
fi
out=${2:-/dev/stdout}
cd $ROOT
for v in $VARIANTS; do
  name=${v%%:*}; mask=${v##*:}; d=build/abl/$name
  hipcc -O2 --offload-arch=gfx950 -Iinclude scripts/conv_bench.cpp -L$d -laddk -ldl -Wl,-rpath,$PWD/$d -o /tmp/conv_bench_$name 2>/dev/null || { echo "link failed: $name"; exit 1; }
done
{
for sh in "stem 3x3" "dil 5x5 d2 40->40" "dil 3x3 d2 40->40" "dil 5x5 d2 80->80 @63x127" "dil 3x3 d2 80->80" "dil 5x5 d2 160" "stem2" "decoder 3x3 304"; do
  for v in $VARIANTS; do
    name=${v%%:*}; mask=${v##*:}
    echo "--- [$name] ablation mask ${mask:-0} (1 no weight loads, 2 no prologue/split, 4 no statistics, 8 first patch only, 16 one MFMA per (tap, tile))"
    SHAPES="$sh" PACKED=1 NOWGRAD=1 ADDK_MATH=${MATH:-f16x3} /tmp/conv_bench_$name 20 2>&1 | grep -v amdgpu.ids
  done
done
} > $out
