cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_bneck; mkdir -p $O
timeout 600 python -m pytest tests/test_bottleneck_tail.py tests/test_conv1x1.py -q -m gpu -x 2>&1 | tail -8 | cut -c1-250
for v in on off on off; do
  PVSG_BNECK_FUSE=$v python bench.py --steps 10 --warmup 3 --cpu-baseline off --sub-benchmarks off --projection off 2>/dev/null | grep '^{' | tail -1 > $O/bench_$v.json
  python -c "
import json; d=json.load(open('$O/bench_$v.json')); k=d['kernels']; print('bneck_fuse $v', d['ms_per_step'], {n:(round(k[n]['ms_per_step'],3),k[n]['calls_per_step']) for n in k if 'bottleneck' in n or 'conv1x1' in n})"
done
