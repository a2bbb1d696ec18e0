cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_parity32; mkdir -p $O
free -g | head -2
PVSG_FULL_CLIP_ORACLE=1 timeout 1500 python -m pytest "tests/test_decoder_parity_at_size.py::test_config3_clip_720p_depends_on_the_decoder" -q -m gpu -s 2>&1 | grep -v Warning | tail -12 | tee $O/decoder_parity_T32.txt
