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
# second half of round 6 (small shapes): per-image flow A/Bs, range-merge kernels, 4-frame step against 1/8 of the 32-frame one
{
  echo "# scripts/lab/ips_image_breakdown.py (one 720p image per call through the shipped IPS detector, hipGraph on), same box"
  for cfg in "PVSG_CONV_SLICES=off PVSG_IMAGE_TAIL=host PVSG_INST_X4=0 PVSG_KV_BATCH=off PVSG_FUSE_LN_MINEFF=0.85" "PVSG_CONV_SLICES=auto PVSG_IMAGE_TAIL=host PVSG_INST_X4=0" \
             "PVSG_CONV_SLICES=auto PVSG_IMAGE_TAIL=host PVSG_INST_X4=1" "PVSG_CONV_SLICES=auto PVSG_IMAGE_TAIL=device PVSG_INST_X4=1"; do
    echo "$cfg: $(env $cfg B1_MODES=on python scripts/lab/ips_image_breakdown.py 2>&1 | grep 'graph on')"
  done
} > $O/batch1_ab.txt
python scripts/lab/combine_time.py 2>/dev/null | grep "^B " > $O/combine_time.txt
bash scripts/lab/t4_vs_t32.sh r06_final/t4cmp > $O/t4_vs_t32.txt 2>/dev/null
B1_MODES=on bash scripts/lab/b1_seq.sh r06_final/b1seq > /dev/null 2>&1
python scripts/lab/seq_short.py $O/b1seq/seq.txt > $O/batch1_seq.txt 2>/dev/null
python scripts/lab/vps_breakdown.py 2>/dev/null | grep "wall ms" > $O/vps_breakdown.txt
ls $O
