#!/bin/bash
# Where wgrad_h3b's time goes: ablation builds of wgrad.hip (numerically WRONG, same launch list) against the product build;
# kernel durations from rocprofv3 --kernel-trace over bench steps.       bash scripts/wgrad_ablate.sh > gpurun_out/wgrad_ablate.txt
#   ABL=1 no global loads / split / LDS stores inside the segment loop     ABL=2 no LDS fragment reads inside the loop     ABL=3 MFMA chain only
set -e
SRC=$GRAFT_REPO_ROOT/auto-dynamic-deeplab_amd/csrc
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in ${WG_VARIANTS:-0 1 2 3}; do
  D=/tmp/addk_abl$v; rm -rf $D; mkdir -p $D/pkg/csrc $D/include
  cp $SRC/*.hip $SRC/*.h $SRC/*.cpp $SRC/Makefile $SRC/*.o $D/pkg/csrc/
  cp $GRAFT_REPO_ROOT/include/addk.h $D/include/
  rm -f $D/pkg/csrc/wgrad.o
  (cd $D/pkg/csrc && make -j16 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-result -ffp-contract=off ${WG_DEFINE:--DADDK_WG_ABL}=$v" > /dev/null 2>&1)
  export ADDK_LIB=$D/pkg/libaddk.so
  rm -rf /tmp/pt_$v; ADDK_STREAMS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt_$v -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras > /tmp/pt_$v.log 2>&1 || { echo "variant $v: bench failed (expected when the loss check trips)"; tail -3 /tmp/pt_$v.log; }
  echo "== ${WG_DEFINE:--DADDK_WG_ABL}=$v"; python3 scripts/trace_summary.py $(ls /tmp/pt_$v/*/*kernel_trace.csv | head -1) 200 | grep -E "launches/step|wgrad_h3b|wgrad_hkb"
done
