cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_t5; mkdir -p $O
python bench.py --steps 10 --warmup 3 > $O/bench_line_T32.json 2> $O/bench_T32.err
tail -3 $O/bench_T32.err
python bench.py --mode ips --frames 8 --steps 10 --warmup 3 > $O/bench_line_ips_T8.json 2> $O/bench_ips.err
tail -3 $O/bench_ips.err
python bench.py --frames 8 --height 1080 --width 1920 --graph off --steps 10 --warmup 3 --sub-benchmarks off > $O/bench_line_1080p_T8.json 2> $O/bench_1080.err
tail -3 $O/bench_1080.err
PVSG_FORCE_COLLECTIVES=1 python bench.py --frames 4 --steps 10 --warmup 3 --cpu-baseline off --sub-benchmarks off --projection off > $O/bench_line_T4_forced_collectives.json 2> $O/bench_fc.err
tail -3 $O/bench_fc.err
python - <<PY
import json
for n in ('T32','ips_T8','1080p_T8','T4_forced_collectives'):
    try:
        d=json.loads(open('$O/bench_line_%s.json'%n).read().strip().split('\n')[-1])
        print(n, d['ms_per_step'], d['value'], d.get('roofline',{}).get('kernel'), d.get('roofline',{}).get('frac'), d.get('cpu_baseline',{}).get('value'), d.get('parity_on_cpu_sample'))
        print('   ', d.get('projected_strong_scaling'), d.get('product_default_hipgraph'), d.get('collectives'))
    except Exception as e: print(n,'ERR',e)
PY
