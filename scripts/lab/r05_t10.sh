cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gemm_f16x2.py tests/test_gemm_bf16x3.py -q -m gpu 2>&1 | tail -1
python scripts/lab/gemm_tile_ab.py 32 2>/dev/null | tail -1
