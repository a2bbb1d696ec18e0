#!/bin/bash
# Regenerates openpvsg_amd/tuning/gemm_gfx950.csv on an MI355X (about 140 s): TunableOp times the rocBLAS / hipBLASLt
# solutions of every fp32 GEMM shape bench.py issues and writes the winners.
set -e
cd "$(dirname "$0")/.."
export PVSG_GEMM_TABLE=off PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 \
       PYTORCH_TUNABLEOP_FILENAME=$PWD/gpurun_out/tunableop.csv PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=150 \
       PYTORCH_TUNABLEOP_MAX_TUNING_ITERATIONS=40 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=20
mkdir -p gpurun_out
python bench.py --steps 5 --warmup 2 --cpu-baseline off
cp gpurun_out/tunableop0.csv openpvsg_amd/tuning/gemm_gfx950.csv
