#!/bin/bash
# builds and runs scripts/coresidency_repro2 (RCCL all_gather and pvsg_xattn_combine_packed as victims); one JSON line per case
# usage (GPU box, repo root): bash scripts/coresidency_repro2.sh >> profiles/r04_coresidency_repro.jsonl
set -u
B=/tmp/coresidency_repro2
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o $B scripts/coresidency_repro2.hip -I/opt/rocm/include -L/opt/rocm/lib -lrccl \
      -Lopenpvsg_amd/lib -lopenpvsg_hip -Wl,-rpath,$PWD/openpvsg_amd/lib -lpthread 2>/dev/null || exit 1
N=${N:-400}
for v in combine allgather; do
  timeout 120 $B none $N $v
  for k in bf16_16x16x32 f16_16x16x32; do
    timeout 120 $B $k $N $v
    REPRO_SAME_PROCESS=1 timeout 120 $B $k $N $v
    REPRO_VICTIM_CU_MASK=0:0-127 REPRO_SPIN_CU_MASK=0:128-255 timeout 120 $B $k $N $v
  done
done
