cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/b1a; mkdir -p $OUT
python $GRAFT_REPO_ROOT/scripts/lab/ips_image_breakdown.py > $OUT/breakdown.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_b1 -o b1 -- python $GRAFT_REPO_ROOT/scripts/lab/ips_image_breakdown.py > $OUT/under_rocprof.log 2>&1
f=$(find /tmp/rp_b1 -name '*kernel_stats.csv' | head -1)
cp $f $OUT/kernel_stats.csv
tail -3 $OUT/breakdown.txt
