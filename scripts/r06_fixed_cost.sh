#!/bin/bash
# short-clip steps (the frame-independent part of the step): product default (hipGraph) at T = 4 / 8, PVSG_KV_FUSE A/B
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r06_fixed}
mkdir -p $OUT
B="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-baseline off --sub-benchmarks off --projection off"
for T in 4 8; do
  for KV in off on; do
    PVSG_KV_FUSE=$KV $B --frames $T 2>/dev/null | tail -1 > $OUT/bench_T${T}_kv_${KV}.json
    python3 - $OUT/bench_T${T}_kv_${KV}.json <<'PY'
import json, sys
l = json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], 'ms_per_step %.3f' % l['ms_per_step'], 'fps %.1f' % l['value'],
      'rows_post %.1f us' % (l['kernels'].get('pvsg_decoder_rows_post', {}).get('avg_ms', 0) * 1e3))
PY
  done
done
