#!/bin/bash
# full GPU suite + smoke + bench lines at 4 / 8 / 32 frames + the T=4 launch sequence (round-5 working script)
TAG=${1:-r05_full}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG
mkdir -p $O
python -m pytest tests -q -m gpu -x 2>&1 | tail -25 > $O/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 > $O/smoke.txt
for T in 4 8; do python bench.py --frames $T --steps 20 --warmup 5 --cpu-baseline off --sub-benchmarks off 2>/dev/null | tail -1 > $O/bench_line_T$T.json; done
python bench.py --steps 10 --warmup 3 --cpu-baseline off --sub-benchmarks off 2>/dev/null | tail -1 > $O/bench_line_T32.json
bash scripts/r05_fixed_cost.sh $TAG/fixed > /dev/null 2>&1
tail -5 $O/pytest.txt; cat $O/smoke.txt
python - <<PY
import json
for T in (4,8,32):
    try:
        d=json.load(open('$O/bench_line_T%d.json'%T)); print(T, d['ms_per_step'], d['value'])
    except Exception as e: print(T, 'ERR', e)
PY
