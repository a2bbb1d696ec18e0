#!/bin/bash
# heuristic knobs of pvsg_conv_slices on the per-image IPS flow
export B1_MODES=on
for cfg in "512 8 160" "512 4 160" "768 4 160" "1024 4 256" "768 8 256" "1024 2 256"; do
  set -- $cfg
  echo "target $1 min_steps $2 max_blocks $3: $(PVSG_SLICE_TARGET=$1 PVSG_SLICE_MIN_STEPS=$2 PVSG_SLICE_MAX_BLOCKS=$3 python $GRAFT_REPO_ROOT/scripts/lab/ips_image_breakdown.py 2>&1 | grep 'graph on')"
done
