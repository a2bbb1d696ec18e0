cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_t4; mkdir -p $O
python -m pytest tests/test_decoder_parity_at_size.py tests/test_modules_gpu.py tests/test_tubes.py tests/test_parallel_gpu.py -q -m gpu -x -s 2>&1 | grep -v Warning | tail -30 > $O/pytest.txt
tail -12 $O/pytest.txt | cut -c1-250
