cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_t6; mkdir -p $O
python -m pytest tests/test_gemm_f16x2.py tests/test_fused_encoder.py -q -m gpu -x 2>&1 | tail -5 > $O/pytest.txt; cat $O/pytest.txt
python scripts/lab/ln_ab.py 32 2>/dev/null | tail -1 | tee $O/ln_ab_T32.json
python scripts/lab/ln_ab.py 16 2>/dev/null | tail -1 | tee $O/ln_ab_T16.json
for v in 128 256; do PVSG_LN_TILE=$v python bench.py --steps 10 --warmup 3 --cpu-baseline off --sub-benchmarks off --projection off 2>/dev/null | grep '^{' | tail -1 > $O/bench_ln$v.json; done
PVSG_FUSE_LN=off python bench.py --steps 10 --warmup 3 --cpu-baseline off --sub-benchmarks off --projection off 2>/dev/null | grep '^{' | tail -1 > $O/bench_lnoff.json
python - <<PY
import json
for n in ('ln128','ln256','lnoff'):
    d=json.load(open('$O/bench_%s.json'%n)); print(n, d['ms_per_step'], d['value'], {k:(round(v['ms_per_step'],3), v['calls_per_step']) for k,v in d['kernels'].items() if 'layernorm' in k or k=='pvsg_gemm_f16x2'})
PY
