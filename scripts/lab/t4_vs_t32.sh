#!/bin/bash
# per-entry kernel time of the 4-frame step (a frame shard of the 8-GPU run) next to 1/8 of the 32-frame step's
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-t4cmp}; mkdir -p $OUT
B="python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --cpu-baseline off --sub-benchmarks off --projection off --timing all --graph off"
$B --frames 4 2>/dev/null | tail -1 > $OUT/t4.json
$B --frames 32 2>/dev/null | tail -1 > $OUT/t32.json
python3 - $OUT <<'PY'
import json, sys
d = sys.argv[1]
a, b = json.load(open(d + '/t4.json')), json.load(open(d + '/t32.json'))
ka, kb = a['kernels'], b['kernels']
def fam(k): return k.split('[')[0]
fa, fb = {}, {}
for k, v in ka.items(): fa[fam(k)] = fa.get(fam(k), 0) + v['ms_per_step']
for k, v in kb.items(): fb[fam(k)] = fb.get(fam(k), 0) + v['ms_per_step']
print('step ms: T4 %.2f  T32 %.2f (/8 = %.2f)' % (a['ms_per_step'], b['ms_per_step'], b['ms_per_step'] / 8))
rows = sorted(fa, key=lambda k: -(fa[k] - fb.get(k, 0) / 8))
print('%-40s %8s %8s %8s' % ('entry', 'T4 ms', 'T32/8', 'excess'))
for k in rows[:24]:
    print('%-40s %8.3f %8.3f %8.3f' % (k, fa[k], fb.get(k, 0) / 8, fa[k] - fb.get(k, 0) / 8))
print('sum excess %.2f' % sum(fa[k] - fb.get(k, 0) / 8 for k in fa))
PY
