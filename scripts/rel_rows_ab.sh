#!/bin/bash
# A/B of the relation row kernels' launch options (each setting is read once per process)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r06_rel_ab}
mkdir -p $OUT
for v in on off; do
  echo "PVSG_REL_XCD_SPLIT=$v" >> $OUT/ab.txt
  for i in 1 2; do PVSG_REL_XCD_SPLIT=$v python $GRAFT_REPO_ROOT/scripts/rel_rows_bench.py 100 32 2>/dev/null | tail -1 >> $OUT/ab.txt; done
done
cat $OUT/ab.txt
