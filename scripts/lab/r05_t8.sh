cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_t8; mkdir -p $O
python -m pytest tests/test_gemm_f16x2.py tests/test_gemm_bf16x3.py tests/test_fused_encoder.py -q -m gpu -x 2>&1 | tail -3
for v in default 128; do
  if [ $v = default ]; then unset PVSG_F16X2_TILE; else export PVSG_F16X2_TILE=$v; fi
  python bench.py --steps 10 --warmup 3 --cpu-baseline off --sub-benchmarks off --projection off 2>/dev/null | grep '^{' | tail -1 > $O/bench_tile_$v.json
done
unset PVSG_F16X2_TILE
python bench.py --steps 10 --warmup 3 --cpu-baseline off --sub-benchmarks off --projection off 2>/dev/null | grep '^{' | tail -1 > $O/bench_tile_default2.json
python - <<PY
import json
for n in ('default','128','default2'):
    d=json.load(open('$O/bench_tile_%s.json'%n)); print(n, d['ms_per_step'], d['value'], {k:(round(v['ms_per_step'],3), v['calls_per_step']) for k,v in d['kernels'].items() if k=='pvsg_gemm_f16x2'})
PY
