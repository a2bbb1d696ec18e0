#!/bin/bash
# PVSG_KV_BATCH on / off (key / value projections of the three layers of a level from one GEMM): steps at T = 4, 32 (alternating),
# the per-image IPS flow, and the tests that depend on the decoder
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-kvb}
mkdir -p $OUT
B="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --cpu-baseline off --sub-benchmarks off --projection off"
for rep in 1 2; do
for T in 4 32; do
  for KV in off on; do
    PVSG_KV_BATCH=$KV $B --frames $T 2>/dev/null | tail -1 > $OUT/bench_T${T}_${KV}_$rep.json
    python3 - $OUT/bench_T${T}_${KV}_$rep.json <<'PY'
import json, sys
l = json.load(open(sys.argv[1]))
k = l['kernels']
print(sys.argv[1].split('/')[-1], 'ms_per_step %.3f' % l['ms_per_step'], 'fps %.1f' % l['value'],
      'gemm_f16x2 %.3f ms (%d)' % (k.get('pvsg_gemm_f16x2', {}).get('ms_per_step', 0), k.get('pvsg_gemm_f16x2', {}).get('calls_per_step', 0)),
      'combine %.1f us' % (k.get('pvsg_xattn_combine', {}).get('avg_ms', 0) * 1e3))
PY
  done
done
done
for KV in off on; do echo KV_BATCH=$KV; PVSG_KV_BATCH=$KV python $GRAFT_REPO_ROOT/scripts/lab/ips_image_breakdown.py 2>&1 | grep "graph on"; done
