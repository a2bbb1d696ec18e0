cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_t2
python -m pytest tests/test_decoder_parity_at_size.py -q -m gpu 2>&1 | tail -60 > gpurun_out/r05_t2/pytest.txt
cat gpurun_out/r05_t2/pytest.txt | cut -c1-250
