cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_kv; mkdir -p $O
python -m pytest tests/test_decoder_rows.py tests/test_modules_gpu.py tests/test_decoder_parity_at_size.py tests/test_parallel_gpu.py -q -m gpu -x 2>&1 | tail -6 | cut -c1-220
for v in on off on off; do
  PVSG_KV_FUSE=$v python bench.py --steps 10 --warmup 3 --cpu-baseline off --sub-benchmarks off --projection off 2>/dev/null | grep '^{' | tail -1 > $O/bench_kv_$v.json
  python -c "
import json; d=json.load(open('$O/bench_kv_$v.json')); print('kv_fuse $v', d['ms_per_step'], d['value'])"
done
for v in on off; do
  PVSG_KV_FUSE=$v python bench.py --frames 4 --steps 20 --warmup 5 --cpu-baseline off --sub-benchmarks off --projection off 2>/dev/null | grep '^{' | tail -1 > $O/bench_kv_T4_$v.json
  python -c "
import json; d=json.load(open('$O/bench_kv_T4_$v.json')); print('T4 kv_fuse $v', d['ms_per_step'], d['value'])"
done
