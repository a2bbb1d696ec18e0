cd $GRAFT_REPO_ROOT
PVSG_W256_RAGGED=1 timeout 900 python -m pytest tests/test_gemm_f16x2.py -q -m gpu 2>&1 | tail -1
for i in 1 2; do
python scripts/lab/power_probe.py proj544 f16x2 2 2>/dev/null | tail -2
PVSG_W256_RAGGED=1 python scripts/lab/power_probe.py proj544 f16x2 2 2>/dev/null | tail -2
done
