#!/bin/bash
# Timing ablations of the f16x2 GEMM kernel: builds lab copies of the library with -DPVSG_ABL=n (csrc/gemm_bf16x3.hip) under
# /tmp and runs scripts/split_ab.py gemm on each.  Results are meaningless numerically; only the times matter.
#   1 no epilogue stores, 2 no W loads, 3 no A loads, 4 neither, 5 no W loads and no W LDS writes, 6 no MFMAs (token GEMM, register-staged form);
#   7 phase timing (scripts/lab/phase_split.py); 8 convolution kernel: no pixel loads, 9: no pixel loads and no split, 11: no epilogue stores
#   (6 removes the MFMAs of the convolution kernels as well)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OBJ=$R/openpvsg_amd/lib/obj
for n in ${@:-0 1 2 3 4 5 6}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DPVSG_ABL=$n -c $R/openpvsg_amd/csrc/gemm_bf16x3.hip -o /tmp/gemm_abl$n.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $OBJ/*.o | grep -v gemm_bf16x3.o) /tmp/gemm_abl$n.o -o /tmp/libpvsg_abl$n.so || exit 1
done
