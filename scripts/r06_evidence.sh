#!/bin/bash
# Round 6: everything profiles/r06_* is copied from, in one pass on the MI355X box:
#   scripts/round_profiles.sh r06  (bench lines, kernel trace, PMC traffic, MFMA utilisation, sub-benchmarks)
#   + the headline line with the WHOLE 32-frame clip through the CPU oracle in the same run (--cpu-full)
#   + the IPS-batch and 1080p lines (BASELINE configs 2 and 5), the forced-collectives lines, the relation-head profile.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash scripts/round_profiles.sh r06 > /dev/null 2>&1
O=$R/gpurun_out/r06_final
B="python $R/bench.py"
$B --cpu-full 2>/dev/null | grep "^{" | tail -1 > $O/bench_line_cpu_full.json
$B --mode ips --frames 8 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_line_ips_T8.json
$B --frames 8 --height 1080 --width 1920 --graph off --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/bench_line_1080p_T8.json
PVSG_FORCE_COLLECTIVES=1 $B --frames 4 --steps 20 --warmup 5 --cpu-baseline off --sub-benchmarks off 2>/dev/null | grep "^{" | tail -1 > $O/bench_line_T4_forced_collectives.json
PVSG_FORCE_COLLECTIVES=1 PVSG_SHARD_GRAPH=on $B --frames 4 --steps 20 --warmup 5 --cpu-baseline off --sub-benchmarks off 2>/dev/null | grep "^{" | tail -1 > $O/bench_line_T4_forced_collectives_shard_graph.json
bash scripts/rel_rows_prof.sh r06_final/rel > /dev/null 2>&1
ls -la $O
python - <<PY
import json, glob
for f in sorted(glob.glob('$O/bench_line*.json')):
    try:
        d = json.load(open(f)); print(f.split('/')[-1], d['ms_per_step'], d['value'], d.get('roofline', {}).get('frac'), d.get('roofline_function', {}).get('kernel_function'))
    except Exception as e: print(f, 'ERR', e)
PY
