cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_b
python bench.py --frames 4 --steps 5 --warmup 2 --cpu-baseline off --sub-benchmarks off > gpurun_out/r05_b/b4.txt 2>&1
tail -30 gpurun_out/r05_b/b4.txt
