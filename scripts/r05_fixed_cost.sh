#!/bin/bash
# Round 5, VERDICT item 1a: where the frame-independent part of a step sits.  Kernel trace of the T=4 clip step (eager and
# graphed) and of the one-image-per-call IPS flow -> steady statistics, gaps, the launch sequence of one step.
#   scripts/r05_fixed_cost.sh <tag>    -> gpurun_out/<tag>/*
set -u
TAG=${1:-r05_fixed}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py"
for T in 4 8; do $B --frames $T --steps 20 --warmup 5 --cpu-baseline off --sub-benchmarks off 2>/dev/null | tail -1 > $O/bench_line_T$T.json; done
for G in off on; do
  rm -rf /tmp/rp_T4
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_T4 -- $B --frames 4 --graph $G --steps 3 --warmup 3 --cpu-baseline off --sub-benchmarks off --no-flop-count --no-kernel-timing > $O/bench_line_T4_graph_${G}_under_rocprof.json 2>/dev/null
  KT=$(find /tmp/rp_T4 -name "*kernel_trace.csv" | head -1)
  python $R/scripts/steady_stats.py $KT --steps 3 > $O/steady_kernel_stats_T4_graph_$G.csv
  python $R/scripts/gap_stats.py $KT > $O/gaps_T4_graph_$G.txt 2>&1
  python $R/scripts/seq_dump.py $KT > $O/seq_T4_graph_$G.txt 2>&1
done
cd $R
python scripts/shipped_config_bench.py 2>/dev/null | tail -1 > $O/shipped_config_bench.json
ls -la $O
