#!/bin/bash
# The fence-free rendezvous of decoder_rows_post (tests/test_decoder_rows.py -k stale: 400 launches x 2 forms against the un-split
# kernel) under the conditions of the 8-rank one-device runs: a 32-CU mask per process, eight processes at once, several rounds.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
ROUNDS=${1:-3}
for round in $(seq 1 $ROUNDS); do
  pids=()
  for slot in 0 1 2 3 4 5 6 7; do
    lo=$((slot * 32)); hi=$((lo + 31))
    HSA_CU_MASK="0:$lo-$hi" PVSG_SHARED_GPU_WARNING=off python -m pytest tests/test_decoder_rows.py -q -m gpu -k "stale" -p no:cacheprovider > /tmp/stress_${round}_$slot.txt 2>&1 &
    pids+=($!)
  done
  for p in "${pids[@]}"; do wait $p; done
  for slot in 0 1 2 3 4 5 6 7; do echo "round $round slot $slot: $(tail -1 /tmp/stress_${round}_$slot.txt)"; done
done
echo "unmasked, one process, 5 times:"
for i in 1 2 3 4 5; do python -m pytest tests/test_decoder_rows.py -q -m gpu -k "stale" -p no:cacheprovider 2>&1 | tail -1; done
