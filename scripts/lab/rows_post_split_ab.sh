#!/bin/bash
# decoder_rows_post: FFN split over eight workgroups per row tile (rendezvous behind __threadfence) vs one workgroup per tile
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r06_rows_split}
mkdir -p $OUT
B="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-baseline off --sub-benchmarks off --projection off --frames 4"
for v in on off on off; do
  PVSG_DECODER_ROWS_SPLIT=$v $B 2>/dev/null | tail -1 > $OUT/t4_$v.json
  python3 - $OUT/t4_$v.json $v <<'PY' | tee -a $OUT/ab.txt
import json, sys
l = json.load(open(sys.argv[1]))
print('PVSG_DECODER_ROWS_SPLIT=%s' % sys.argv[2], 'ms_per_step %.3f' % l['ms_per_step'],
      'rows_post %.1f us' % (l['kernels'].get('pvsg_decoder_rows_post', {}).get('avg_ms', 0) * 1e3))
PY
done
