#!/bin/bash
# Diagnostic build of libaddk.so with per-phase stamps in sepf_kernel (into /tmp, the in-tree library is untouched), then scripts/sepf_phases.py.
#   bash scripts/sepf_phases.sh > gpurun_out/r04_sepf_phases.txt
set -e
SRC=$GRAFT_REPO_ROOT/auto-dynamic-deeplab_amd/csrc
D=/tmp/addk_diag; rm -rf $D; mkdir -p $D/pkg/csrc $D/include
cp $SRC/*.hip $SRC/*.h $SRC/*.cpp $SRC/Makefile $D/pkg/csrc/
cp $GRAFT_REPO_ROOT/include/addk.h $D/include/
cp $SRC/*.o $D/pkg/csrc/ 2>/dev/null || true
rm -f $D/pkg/csrc/sepf.o
(cd $D/pkg/csrc && make -j16 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-result -ffp-contract=off -DADDK_SEPF_DIAG" > /dev/null 2>&1)
cd $GRAFT_REPO_ROOT
ADDK_LIB=$D/pkg/libaddk.so python3 scripts/sepf_phases.py 2>&1 | grep -v amdgpu.ids
