#!/bin/bash
# launch sequence of ONE image through the shipped IPS detector (one image per call): rocprofv3 kernel trace, last image
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-b1seq}; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_b1s -o b1 -- python $GRAFT_REPO_ROOT/scripts/lab/ips_image_breakdown.py > $OUT/under_rocprof.log 2>&1
f=$(find /tmp/rp_b1s -name '*kernel_trace.csv' | head -1)
python $GRAFT_REPO_ROOT/scripts/seq_dump.py $f --marker stem7x7 > $OUT/seq.txt
head -3 $OUT/seq.txt
tail -4 $OUT/under_rocprof.log
