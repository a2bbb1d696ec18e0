cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_t3; mkdir -p $O
python -m pytest tests/test_decoder_parity_at_size.py tests/test_modules_gpu.py tests/test_tubes.py tests/test_parallel_gpu.py -q -m gpu -x -s 2>&1 | tail -30 > $O/pytest.txt
tail -8 $O/pytest.txt | cut -c1-250
python scripts/lab/host_phases.py 4 2>/dev/null | tail -1 | tee $O/host_phases_T4.json
python scripts/lab/host_phases.py 8 2>/dev/null | tail -1 | tee $O/host_phases_T8.json
for T in 4 8; do python bench.py --frames $T --steps 20 --warmup 5 --cpu-baseline off --sub-benchmarks off 2>/dev/null | tail -1 > $O/bench_line_T$T.json; done
python scripts/lab/ips_image_breakdown.py 2>/dev/null | tail -3 | tee $O/ips_image_breakdown.txt
python - <<PY
import json
for T in (4,8):
    d=json.load(open('$O/bench_line_T%d.json'%T)); print(T, d['ms_per_step'], d['value'])
PY
