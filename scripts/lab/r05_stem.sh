cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_stem; mkdir -p $O
timeout 600 python -m pytest tests/test_winograd.py -q -m gpu -k stem 2>&1 | tail -3 | cut -c1-250
timeout 900 python -m pytest tests/test_modules_gpu.py tests/test_bottleneck_tail.py -q -m gpu -x 2>&1 | tail -2 | cut -c1-250
for v in f16x2 f32 f16x2 f32; do
  PVSG_STEM=$v python bench.py --steps 10 --warmup 3 --cpu-baseline off --sub-benchmarks off --projection off 2>/dev/null | grep '^{' | tail -1 > $O/bench_$v.json
  python -c "
import json; d=json.load(open('$O/bench_$v.json')); k=d['kernels']; print('stem $v', d['ms_per_step'], {n:(round(k[n]['ms_per_step'],3),k[n]['calls_per_step']) for n in k if 'stem' in n})"
done
