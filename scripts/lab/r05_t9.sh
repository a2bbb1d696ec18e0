cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_t9; mkdir -p $O
python -m pytest tests/test_parallel_gpu.py tests/test_pair_score.py tests/test_postprocess.py tests/test_stream_guard.py tests/test_tubes.py tests/test_unitrack.py tests/test_winograd.py tests/test_xattn.py -q -m gpu 2>&1 | tail -8 | cut -c1-200 > $O/pytest.txt; cat $O/pytest.txt
