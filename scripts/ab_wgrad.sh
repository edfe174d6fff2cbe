#!/bin/bash
# A/B of the wide-conv weight gradient on the stand-alone harness: scripts/_old/libaddk.so (previous build) vs the tree's;
# COLD=1 adds a 1 GiB fill in front of every timed run (operands from HBM instead of the caches)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in ${VARIANTS:-old new}; do
  L=$PWD/auto-dynamic-deeplab_amd; [ $v = old ] && L=$PWD/scripts/_old
  [ -f $L/libaddk.so ] || continue
  hipcc -O2 --offload-arch=gfx950 -Iinclude scripts/conv_bench.cpp -L$L -laddk -Wl,-rpath,$L -o /tmp/conv_bench_$v 2>/dev/null || exit 1
  echo "== $v"; SHAPES="${SHAPES:-decoder}" /tmp/conv_bench_$v 10 | grep -E "${MODES:-wgrad}"
done
