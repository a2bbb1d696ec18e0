#!/bin/bash
# one rank's share of the frame-sharded clip (T = 4 of 32) with every exchange running over RCCL at world 1: eager launches vs
# the whole shard step (backbone + head + the nine record all-gathers) replayed as one hipGraph (PVSG_SHARD_GRAPH=on)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r06_shard_graph}
mkdir -p $OUT
B="python $GRAFT_REPO_ROOT/bench.py --frames 4 --steps 20 --warmup 5 --cpu-baseline off --sub-benchmarks off --projection off"
for v in off on off on; do
  PVSG_FORCE_COLLECTIVES=1 PVSG_SHARD_GRAPH=$v $B 2>/dev/null | grep "^{" | tail -1 > $OUT/t4_forced_graph_$v.json
  python3 - $OUT/t4_forced_graph_$v.json $v <<'PY' | tee -a $OUT/ab.txt
import json, sys
l = json.load(open(sys.argv[1]))
print('PVSG_FORCE_COLLECTIVES=1 PVSG_SHARD_GRAPH=%s' % sys.argv[2], 'ms_per_step %.3f' % l['ms_per_step'], 'fps %.1f' % l['value'],
      'exchanges', l.get('collectives', {}).get('exchange_us_per_step'))
PY
done
python $GRAFT_REPO_ROOT/bench.py --frames 4 --steps 20 --warmup 5 --cpu-baseline off --sub-benchmarks off --projection off 2>/dev/null | grep "^{" | tail -1 | python3 -c "
import json, sys
l = json.loads(sys.stdin.read()); print('no process group (hipGraph, no exchanges): ms_per_step %.3f' % l['ms_per_step'])" | tee -a $OUT/ab.txt
