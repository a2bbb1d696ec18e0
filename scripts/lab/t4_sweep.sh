#!/bin/bash
# 4- and 8-frame steps (product default: hipGraph replay) under the small-shape knobs
B="python $GRAFT_REPO_ROOT/bench.py --steps 15 --warmup 4 --cpu-baseline off --sub-benchmarks off --projection off --no-kernel-timing"
run() { echo "$1 T=$2: $(env $1 $B --frames $2 2>/dev/null | tail -1 | python3 -c 'import json,sys; print("%.3f" % json.loads(sys.stdin.read())["ms_per_step"])')"; }
for T in 4 8; do
  run "PVSG_X=0" $T
  run "PVSG_SLICE_MAX_BLOCKS=320" $T
  run "PVSG_SLICE_MAX_BLOCKS=640 PVSG_SLICE_TARGET=1024" $T
  run "PVSG_F16X2_TILE=128" $T
  run "PVSG_KV_BATCH=off" $T
  run "PVSG_X=0" $T
done
