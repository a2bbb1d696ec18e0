#!/bin/bash
# per-kernel times of the relation head (row kernels): rocprofv3 kernel trace of scripts/rel_rows_bench.py
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r06_rel}
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_rel -o rel -- python $GRAFT_REPO_ROOT/scripts/rel_rows_bench.py 100 32 > $OUT/bench.log 2>&1
f=$(find /tmp/rp_rel -name '*kernel_stats.csv' | head -1)
python3 - "$f" > $OUT/kernel_stats_head.csv <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
print(','.join(rows[0]))
for r in rows[1:31]:
    print(','.join(['"%s"' % r[0][:90]] + r[1:]))
PY
cat $OUT/kernel_stats_head.csv
tail -2 $OUT/bench.log
