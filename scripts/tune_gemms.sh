#!/bin/bash
# Regenerates openpvsg_amd/tuning/gemm_gfx950.csv on an MI355X (about 140 s per clip length): TunableOp times the rocBLAS /
# hipBLASLt solutions of every fp32 GEMM shape bench.py issues -- at 32 frames (one GPU) and at 16 / 8 / 4 frames (what a rank
# holds when the clip is sharded over 2 / 4 / 8 GPUs) -- and writes the winners (results accumulate in one file).
set -e
cd "$(dirname "$0")/.."
export PVSG_GEMM_TABLE=off PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 \
       PYTORCH_TUNABLEOP_FILENAME=$PWD/gpurun_out/tunableop.csv PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=150 \
       PYTORCH_TUNABLEOP_MAX_TUNING_ITERATIONS=40 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=20
mkdir -p gpurun_out
rm -f gpurun_out/tunableop0.csv
for T in 32 16 8 4; do
  python bench.py --frames $T --steps 3 --warmup 2 --cpu-baseline off --sub-benchmarks off --no-flop-count > gpurun_out/tune_T$T.log 2>&1
  wc -l gpurun_out/tunableop0.csv
done
cp gpurun_out/tunableop0.csv openpvsg_amd/tuning/gemm_gfx950.csv
