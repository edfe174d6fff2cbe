#!/bin/bash
# A/B of the stream modes (plan.Graph.stream_mode): step time of config 2, captured and eager, one process per variant, ABAB on one box.
# usage: scripts/ab_streams.sh OUTDIR
out=${1:-gpurun_out/ab_streams}; mkdir -p $out
run() {   # name, ADDK_STREAM_PRIO, ADDK_STREAMS, extra flags
  ADDK_STREAM_PRIO=$2 ADDK_STREAMS=$3 python bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 5 $4 > $out/$1.json 2> $out/$1.err || { echo "$1 FAILED"; tail -5 $out/$1.err; return 1; }
  python - "$out/$1.json" "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('%-22s %.3f ms/step  (%s)' % (sys.argv[2], d['ms_per_step'], d.get('ms_per_step_by_mode')))
PY
}
run base2_a   0   2 && run p2_a     p2  2 && run w3n_a    w3n 3 && run w3_a     w3  3 && \
run base2_b   0   2 && run p2_b     p2  2 && run w3n_b    w3n 3 && run w3_b     w3  3 && \
run base3     0   3 && \
run eager_base2 0 2 --no-graph && run eager_p2 p2 2 --no-graph && run eager_w3 w3 3 --no-graph && run eager_w3n w3n 3 --no-graph
