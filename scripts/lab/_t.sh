cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_final; mkdir -p $O
python bench.py 2>/dev/null | grep "^{" | tail -1 > $O/bench_line.json
python bench.py --frames 16 --steps 20 --warmup 5 --cpu-baseline off --sub-benchmarks off 2>/dev/null | grep "^{" | tail -1 > $O/bench_line_T16.json
python - <<PY
import json
for f in ("bench_line","bench_line_T16"):
    d=json.load(open("$O/%s.json"%f)); r=d["roofline"]; print(f, d["ms_per_step"], d["value"], r["kernel"], round(r["frac"],3), r.get("traffic"), r.get("limited_by","")[:60])
PY
