cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_gn; mkdir -p $O
python -m pytest tests/test_glue_kernels.py tests/test_fused_encoder.py tests/test_conv1x1.py tests/test_modules_gpu.py -q -m gpu -x 2>&1 | tail -4 | cut -c1-200
for v in on off on off; do
  PVSG_GN_EPILOGUE=$v python bench.py --steps 10 --warmup 3 --cpu-baseline off --sub-benchmarks off --projection off 2>/dev/null | grep '^{' | tail -1 > $O/bench_gn_$v.json
  python -c "
import json; d=json.load(open('$O/bench_gn_$v.json')); k=d['kernels']; print('gn_epilogue $v', d['ms_per_step'], {n:(round(k[n]['ms_per_step'],3),k[n]['calls_per_step']) for n in k if 'group_norm' in n or 'conv1x1' in n})"
done
