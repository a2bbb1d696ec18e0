#!/bin/bash
# Everything a round's profiles/ entries come from, in one pass on the MI355X box (run from the repo root):
#   scripts/round_profiles.sh r03        -> gpurun_out/r03_final/*   (copy what is to be judged into profiles/)
# PMC passes use --kernel-trace only (never combined with sys/hip traces).
set -u
TAG=${1:-rNN}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${TAG}_final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py"
$B > $O/bench_line.json 2> $O/bench_line.err
for T in 4 8 16; do $B --frames $T --steps 20 --warmup 5 --cpu-baseline off --sub-benchmarks off 2>/dev/null | tail -1 > $O/bench_line_T$T.json; done
rm -rf /tmp/rp_bench
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_bench -- $B --steps 3 --warmup 2 --cpu-baseline off --sub-benchmarks off --no-flop-count --projection off > $O/bench_line_under_rocprof.json 2>/dev/null
KT=$(find /tmp/rp_bench -name "*kernel_trace.csv" | head -1)
KS=$(find /tmp/rp_bench -name "*kernel_stats.csv" | head -1)
python $R/scripts/steady_stats.py $KT --steps 3 > $O/steady_kernel_stats.csv
head -41 $KS > $O/rocprof_kernel_stats_top40.csv
python $R/scripts/gap_stats.py $KT > $O/gaps.txt 2>&1
python $R/scripts/pmc_traffic.py --out $O/pmc_traffic.json --keep-csv $O/pmc_bench > $O/pmc_traffic.log 2>&1
python $R/scripts/mfma_util.py > $O/mfma_util.txt 2> $O/mfma_util.err
cd $R
python scripts/shipped_config_bench.py 2>/dev/null | tail -1 > $O/shipped_config_bench.json
python scripts/ips_pipeline_bench.py 2>/dev/null | tail -1 > $O/ips_pipeline_bench.json
python scripts/kbench.py xattn --frames 32 > $O/kbench_xattn.jsonl 2>/dev/null
python scripts/kbench.py maskgemm --frames 32 > $O/kbench_maskgemm.jsonl 2>/dev/null
python scripts/kbench.py msda --frames 32 > $O/kbench_msda.jsonl 2>/dev/null
python scripts/conv1x1_bf16x3_bench.py > $O/conv1x1_bf16x3_bench.jsonl 2>/dev/null
python scripts/gemm_bf16x3_bench.py own > $O/gemm_bf16x3_bench.jsonl 2>/dev/null
ls -la $O
