cd $GRAFT_REPO_ROOT
for g in auto off; do for v in on off on off; do
PVSG_KV_FUSE=$v python bench.py --frames 4 --graph $g --steps 20 --warmup 5 --cpu-baseline off --sub-benchmarks off --projection off --no-kernel-timing 2>/dev/null | grep -E "^\{" | tail -1 | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print('T4 graph $g kv $v', d['ms_per_step'])"
done; done
for v in on off; do
PVSG_KV_FUSE=$v python bench.py --frames 8 --steps 20 --warmup 5 --cpu-baseline off --sub-benchmarks off --projection off --no-kernel-timing 2>/dev/null | grep -E "^\{" | tail -1 | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print('T8 kv $v', d['ms_per_step'])"
done
