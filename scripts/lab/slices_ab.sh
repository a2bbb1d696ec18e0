#!/bin/bash
# PVSG_CONV_SLICES off / auto: the per-image IPS flow and the 4-frame step; then the tests
for S in off auto; do echo CONV_SLICES=$S; PVSG_CONV_SLICES=$S python $GRAFT_REPO_ROOT/scripts/lab/ips_image_breakdown.py 2>&1 | grep "graph on"; done
B="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --cpu-baseline off --sub-benchmarks off --projection off"
for S in off auto off auto; do PVSG_CONV_SLICES=$S $B --frames 4 2>/dev/null | tail -1 | python3 -c "
import json,sys
l=json.loads(sys.stdin.read()); print('T4 slices $S ms_per_step %.3f'%l['ms_per_step'])"; done
