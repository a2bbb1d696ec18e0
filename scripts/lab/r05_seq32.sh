#!/bin/bash
# launch sequence of one 32-frame step (per-launch durations in order) -> gpurun_out/r05_seq32/
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_seq32; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp32
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp32 -- python $R/bench.py --steps 3 --warmup 2 --cpu-baseline off --sub-benchmarks off --no-flop-count --no-kernel-timing --projection off > $O/bench_line_under_rocprof.json 2>/dev/null
KT=$(find /tmp/rp32 -name "*kernel_trace.csv" | head -1)
python $R/scripts/steady_stats.py $KT --steps 3 > $O/steady_kernel_stats.csv
python $R/scripts/gap_stats.py $KT > $O/gaps.txt 2>&1
python $R/scripts/seq_dump.py $KT > $O/seq.txt 2>&1
ls -la $O
