#!/bin/bash
# relation head by size and route: fused row kernels (PVSG_RELATION_GEMM_ROWS huge), token-GEMM route (0), library (PVSG_RELATION_ROWS=off)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r06_routes}
mkdir -p $OUT
for NT in "100 32" "100 64" "100 128" "30 400" "100 300" "40 1000"; do
  for R in rows gemm; do
    if [ $R = rows ]; then export PVSG_RELATION_GEMM_ROWS=1000000000; else export PVSG_RELATION_GEMM_ROWS=0; fi
    python $GRAFT_REPO_ROOT/scripts/rel_rows_bench.py $NT 2>/dev/null | tail -1 | python3 -c "
import json, sys
l = json.loads(sys.stdin.read())
print('N=%-4d T=%-5d route=%-5s graph %.3f ms  (encoders %.3f, temporal %.3f)   library graph %.3f ms' % (
    l['N'], l['T'], '$R', l['rows_on']['graph_ms'], l['parts_eager_ms']['encoders'], l['parts_eager_ms']['temporal'], l['rows_off']['graph_ms']))" | tee -a $OUT/routes.txt
  done
done
