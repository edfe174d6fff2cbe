#!/bin/bash
# step time under a tuning knob's values, ABAB on one box:  scripts/knob_sweep.sh NAME v1 v2 ...
cd $(dirname $0)/..
K=$1; shift
for rep in 1 2; do for v in "$@"; do
  ms=$(env $K=$v timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print('%.2f' % json.loads(sys.stdin.read().strip().split('\n')[-1])['ms_per_step'])")
  echo "$K=$v  $ms ms"
done; done
