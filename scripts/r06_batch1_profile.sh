#!/bin/bash
# per-kernel times of the shipped per-image flows (IPS one image per call; VPS per-frame + MinVIS): rocprofv3 kernel trace
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r06_batch1}
mkdir -p $OUT
python $GRAFT_REPO_ROOT/scripts/shipped_config_bench.py 8 2>/dev/null | tail -1 > $OUT/shipped_T8.json
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_b1 -o b1 -- python $GRAFT_REPO_ROOT/scripts/shipped_config_bench.py 8 > $OUT/under_rocprof.log 2>&1
f=$(find /tmp/rp_b1 -name '*kernel_stats.csv' | head -1)
python3 - "$f" > $OUT/kernel_stats_head.csv <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
print(','.join(rows[0]))
for r in rows[1:41]:
    print(','.join(['"%s"' % r[0][:100]] + r[1:]))
PY
cat $OUT/shipped_T8.json
cut -c1-190 $OUT/kernel_stats_head.csv | head -42
