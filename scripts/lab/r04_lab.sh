#!/bin/bash
# Round 4 laboratory pass on the GPU box (repo root): the measurements DESIGN.md section 3.14 quotes, into gpurun_out/r04_lab/
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_lab
mkdir -p $O
cd $R
{
echo "== scripts/split_ab.py all 10: both split forms, time per launch and max error vs float64 next to the library f32 result"
timeout 300 python scripts/split_ab.py all 10 2>/dev/null
} > $O/split_ab.jsonl
{
echo "== scripts/lab/conv3x3_ab.py (PVSG_SPLIT=f16x2, then bf16x3)"
timeout 200 python scripts/lab/conv3x3_ab.py 2>/dev/null
PVSG_SPLIT=bf16x3 timeout 200 python scripts/lab/conv3x3_ab.py 2>/dev/null
} > $O/conv3x3_ab.txt
{
echo "== scripts/lab/power_probe.py: rocm-smi socket power / sclk while one layer loops for 3 s"
for m in f16x2 bf16x3; do timeout 60 python scripts/lab/power_probe.py ffn1 $m 3 2>/dev/null; done
timeout 60 python scripts/lab/power_probe.py ffn2 f16x2 3 2>/dev/null
echo "-- PVSG_F16X2_TILE=256"
PVSG_F16X2_TILE=256 timeout 60 python scripts/lab/power_probe.py ffn1 f16x2 3 2>/dev/null
PVSG_F16X2_TILE=256 timeout 60 python scripts/lab/power_probe.py ffn2 f16x2 3 2>/dev/null
} > $O/power_probe.txt
{
echo "== scripts/lab/abl_split.sh + split_ab.py gemm 5: timing ablations of the 128 x 128 register-staged kernels (PVSG_F16X2_DMA=0)"
echo "   0 as built, 1 no epilogue stores, 2 no W loads, 3 no A loads, 4 neither, 5 no W loads and no W LDS writes, 6 no MFMAs"
scripts/lab/abl_split.sh 0 1 2 3 4 5 6 7
for n in 0 1 2 3 4 5 6; do echo "ABL $n"; PVSG_F16X2_DMA=0 PVSG_LIB_PATH=/tmp/libpvsg_abl$n.so timeout 120 python scripts/split_ab.py gemm 5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   %-32s bf16x3 %.3f ms  f16x2 %.3f ms' % (d['layer'], d['ms_bf16x3'], d['ms_f16x2']))"; done
echo "== token GEMM forms: PVSG_F16X2_DMA=0 (register-staged weights) / default (LDS-DMA) / PVSG_F16X2_TILE=256"
for e in "PVSG_F16X2_DMA=0" "PVSG_F16X2_DMA=1" "PVSG_F16X2_TILE=256"; do echo $e; env $e timeout 120 python scripts/split_ab.py gemm 10 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   %-32s f16x2 %.3f ms' % (d['layer'], d['ms_f16x2']))"; done
echo "== scripts/lab/phase_split.py (lab build 7; the per-workgroup atomics perturb the totals, the proportions are what counts)"
PVSG_LIB_PATH=/tmp/libpvsg_abl7.so timeout 200 python scripts/lab/phase_split.py 2>/dev/null
} > $O/split_lab.txt
{
for g in off on off on; do timeout 300 python bench.py --steps 10 --warmup 3 --cpu-baseline off --sub-benchmarks off --graph $g 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('--graph $g: %.2f ms per step, %.1f frames/s' % (d['ms_per_step'], d['value']))"; done
} > $O/graph_T32.txt
for m in f16x2 bf16x3; do PVSG_SPLIT=$m timeout 300 python bench.py --steps 10 --warmup 3 --cpu-baseline off --sub-benchmarks off --checksum 2>/dev/null | tail -1 > $O/bench_split_$m.json; done
PVSG_FUSE_LN=off timeout 300 python bench.py --steps 10 --warmup 3 --cpu-baseline off --sub-benchmarks off --checksum 2>/dev/null | tail -1 > $O/bench_fuse_ln_off.json
PVSG_CONV3X3=f32 timeout 300 python bench.py --steps 10 --warmup 3 --cpu-baseline off --sub-benchmarks off --checksum 2>/dev/null | tail -1 > $O/bench_conv3x3_f32.json
python scripts/pmc_traffic.py --out $O/pmc_traffic.json --keep-csv $O/pmc_bench > $O/pmc_traffic.log 2>&1
ls -la $O
