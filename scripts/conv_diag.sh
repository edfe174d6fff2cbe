#!/bin/bash
# build the harness on the box and time the heavy convs (env VARS="NAME=v1,v2" sweeps one environment variable)
set -e
hipcc -O2 --offload-arch=gfx950 -Iinclude scripts/conv_bench.cpp -Lauto-dynamic-deeplab_amd -laddk -Wl,-rpath,$PWD/auto-dynamic-deeplab_amd -o /tmp/conv_bench 2>/dev/null
name=${SWEEP:-ADDK_DIAG}
for d in ${VALS:-0}; do echo "== $name=$d"; env $name=$d timeout -k 10 120 /tmp/conv_bench 20; done
