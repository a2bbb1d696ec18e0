#!/bin/bash
# Cache-policy hints on the 1x1 convolution's streaming accesses (nt loads of pixels / identity, nt stores of the output): A/B on one box.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
OBJ=$R/openpvsg_amd/lib/obj
O=gpurun_out/r05_nt_lab; mkdir -p $O
build() {  # name, extra flags
  for f in token_gemm conv1x1_split conv3x3_halo bottleneck_tail; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $2 -c $R/openpvsg_amd/csrc/$f.hip -o /tmp/${f}_$1.o || exit 1
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $OBJ/*.o | grep -v "token_gemm.o\|conv1x1_split.o\|conv3x3_halo.o\|bottleneck_tail.o") /tmp/token_gemm_$1.o /tmp/conv1x1_split_$1.o /tmp/conv3x3_halo_$1.o /tmp/bottleneck_tail_$1.o -o /tmp/libpvsg_$1.so || exit 1
}
build ntst "-DPVSG_NT_ST=2" &
build ntld "-DPVSG_NT_LD=2" &
build ntboth "-DPVSG_NT_ST=2 -DPVSG_NT_LD=2" &
wait
{
for v in base ntst ntld ntboth base ntboth; do
  echo "== $v"
  if [ $v = base ]; then unset PVSG_LIB_PATH; else export PVSG_LIB_PATH=/tmp/libpvsg_$v.so; fi
  for sh in c256_64 c64_256 c512_128 c256_1024; do python scripts/lab/power_probe_conv.py $sh 2 2>/dev/null | tail -1; done
done
unset PVSG_LIB_PATH
} 2>&1 | tee $O/nt_lab.txt
