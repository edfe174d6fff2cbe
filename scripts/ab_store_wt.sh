#!/bin/bash
# A/B of write-through (sc1) 16-byte global stores in EVERY kernel (-DADDK_ST_WT, csrc/common.h st4) against the default write-back stores:
# a second build of the library in /tmp, the bench's step time both ways on the same box, twice each (ABAB).
#   bash scripts/ab_store_wt.sh > gpurun_out/r04_ab_store_wt.txt
set -e
SRC=$GRAFT_REPO_ROOT/auto-dynamic-deeplab_amd/csrc
D=/tmp/addk_wt; rm -rf $D; mkdir -p $D/pkg/csrc $D/include
cp $SRC/*.hip $SRC/*.h $SRC/*.cpp $SRC/Makefile $D/pkg/csrc/
cp $GRAFT_REPO_ROOT/include/addk.h $D/include/
(cd $D/pkg/csrc && make -j16 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-result -ffp-contract=off -DADDK_ST_WT" > /dev/null 2>&1)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for v in default wt; do
    if [ $v = wt ]; then export ADDK_LIB=$D/pkg/libaddk.so; else unset ADDK_LIB; fi
    python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v run $rep: %.3f ms/step  loss %.7f  dominant conv %.4f ms' % (d['ms_per_step'], d['loss'], d['roofline']['launch_ms']))"
  done
done
