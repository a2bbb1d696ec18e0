#!/bin/bash
# A/B of the relation row kernels' launch options (each setting is read once per process): per-kernel times from rocprofv3
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r06_rel_ab}
mkdir -p $OUT
N=${2:-100}; T=${3:-32}
for v in on off; do
  rm -rf /tmp/rp_ab
  PVSG_REL_TAIL_SPLIT=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_ab -o ab -- python $GRAFT_REPO_ROOT/scripts/rel_rows_bench.py $N $T > $OUT/bench_tail_$v.log 2>&1
  f=$(find /tmp/rp_ab -name '*kernel_stats.csv' | head -1)
  echo "PVSG_REL_TAIL_SPLIT=$v N=$N T=$T" >> $OUT/ab.txt
  grep "rel_\|pair\|top_pairs" "$f" | python3 -c "
import csv, sys
for r in csv.reader(sys.stdin):
    print('  %-40s calls %5s avg %9.1f us' % (r[0][:40], r[1], float(r[3]) / 1e3))" >> $OUT/ab.txt
  tail -1 $OUT/bench_tail_$v.log >> $OUT/ab.txt
done
cat $OUT/ab.txt
