cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_t1
python -m pytest tests/test_device_tail.py tests/test_decoder_rows.py tests/test_stream_guard.py tests/test_tubes.py tests/test_postprocess.py tests/test_capi.py -q -m gpu 2>&1 | tail -60 > gpurun_out/r05_t1/pytest_a.txt
cat gpurun_out/r05_t1/pytest_a.txt
