cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_bneck; mkdir -p $O
for i in 1 2 3; do timeout 600 python -m pytest tests/test_bottleneck_tail.py tests/test_capi.py -q -m gpu 2>&1 | tail -1 | cut -c1-250; done
timeout 900 python -m pytest tests/test_modules_gpu.py tests/test_unitrack.py -q -m gpu -x 2>&1 | tail -2 | cut -c1-250
for v in on off on off; do
  PVSG_BNECK_FUSE_NEXT_STAGE=$v python bench.py --steps 10 --warmup 3 --cpu-baseline off --sub-benchmarks off --projection off 2>/dev/null | grep '^{' | tail -1 > $O/bench_ns_$v.json
  python -c "
import json; d=json.load(open('$O/bench_ns_$v.json')); k=d['kernels']; print('next_stage $v', d['ms_per_step'], {n:(round(k[n]['ms_per_step'],3),k[n]['calls_per_step']) for n in k if 'bottleneck' in n or 'conv1x1' in n})"
done
