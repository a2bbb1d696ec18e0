#!/bin/bash
# Timing ablations of the f16x2 GEMM kernel: builds lab copies of the library with -DPVSG_ABL=n (the split sources: csrc/token_gemm.hip, conv1x1_split.hip, conv3x3_halo.hip, bottleneck_tail.hip) under
# /tmp and runs scripts/split_ab.py gemm on each.  Results are meaningless numerically; only the times matter.
#   1 no epilogue stores, 2 no W loads, 3 no A loads, 4 neither, 5 no W loads and no W LDS writes, 6 no MFMAs (token GEMM, register-staged form);
#   7 phase timing (scripts/lab/phase_split.py); 8 convolution kernel: no pixel loads, 9: no pixel loads and no split, 11: no epilogue stores
#   (6 removes the MFMAs of the convolution kernels as well)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OBJ=$R/openpvsg_amd/lib/obj
for n in ${@:-0 1 2 3 4 5 6}; do
  for f in token_gemm conv1x1_split conv3x3_halo bottleneck_tail; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DPVSG_ABL=$n -c $R/openpvsg_amd/csrc/$f.hip -o /tmp/${f}_abl$n.o || exit 1
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $OBJ/*.o | grep -v "token_gemm.o\|conv1x1_split.o\|conv3x3_halo.o\|bottleneck_tail.o") /tmp/token_gemm_abl$n.o /tmp/conv1x1_split_abl$n.o /tmp/conv3x3_halo_abl$n.o /tmp/bottleneck_tail_abl$n.o -o /tmp/libpvsg_abl$n.so || exit 1
done
