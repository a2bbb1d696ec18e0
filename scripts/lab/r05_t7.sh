cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_t7; mkdir -p $O
python scripts/lab/gemm_tile_ab.py 32 2>/dev/null | tail -1 | tee $O/gemm_tile_ab_T32.json
python scripts/lab/gemm_tile_ab.py 4 2>/dev/null | tail -1 | tee $O/gemm_tile_ab_T4.json
