#!/bin/bash
# Upper bound of what pre-split bf16 planes could give the split-bf16 weight-gradient kernels: a build whose fp32 -> 3 x bf16 split
# costs ONE conversion (the planes m and l are copies of h: numerically WRONG, same instruction stream otherwise) against the product build;
# the kernels' durations from rocprofv3 --kernel-trace over bench steps.    bash scripts/wgrad_split_bound.sh > gpurun_out/wgrad_split_bound.txt
set -e
SRC=$GRAFT_REPO_ROOT/auto-dynamic-deeplab_amd/csrc
D=/tmp/addk_fake; rm -rf $D; mkdir -p $D/pkg/csrc $D/include
cp $SRC/*.hip $SRC/*.h $SRC/*.cpp $SRC/Makefile $SRC/*.o $D/pkg/csrc/
cp $GRAFT_REPO_ROOT/include/addk.h $D/include/
rm -f $D/pkg/csrc/wgrad.o
(cd $D/pkg/csrc && make -j16 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-result -ffp-contract=off -DADDK_WG_FAKE_SPLIT" > /dev/null 2>&1)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in product fake_split; do
  if [ $v = fake_split ]; then export ADDK_LIB=$D/pkg/libaddk.so; else unset ADDK_LIB; fi
  rm -rf /tmp/pt_$v; ADDK_STREAMS=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt_$v -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
  echo "== $v"; python3 scripts/trace_summary.py $(ls /tmp/pt_$v/*/*kernel_trace.csv | head -1) 200 | grep -E "launches/step|wgrad_h3b|wgrad_hkb"
done
