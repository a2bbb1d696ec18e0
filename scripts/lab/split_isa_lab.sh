#!/bin/bash
# the instruction sequence of the f16x2 limb split, A/B on one box.
#   isa0      product build (SLP-vectorised: v_pk_mul_f32 + v_pk_fma_f32 beside the MFMAs)
#   isa0_noslp  the same source with -fno-slp-vectorize (scalar v_mul_f32 / v_fma_f32)
#   isa1      v_fma_mix_f32 + v_fma_mixlo/hi_f16 (5 VALU per pair, none packed; csrc/split_common.h PVSG_SPLIT_ISA=1)
#   isa1_noslp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
OBJ=$R/openpvsg_amd/lib/obj
O=gpurun_out/r05_split_lab; mkdir -p $O
build() {  # name, extra flags
  for f in token_gemm conv1x1_split conv3x3_halo bottleneck_tail; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $2 -c $R/openpvsg_amd/csrc/$f.hip -o /tmp/${f}_$1.o || exit 1
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $OBJ/*.o | grep -v "token_gemm.o\|conv1x1_split.o\|conv3x3_halo.o\|bottleneck_tail.o") /tmp/token_gemm_$1.o /tmp/conv1x1_split_$1.o /tmp/conv3x3_halo_$1.o /tmp/bottleneck_tail_$1.o -o /tmp/libpvsg_$1.so || exit 1
}
build isa0_noslp "-fno-slp-vectorize"
build isa1 "-DPVSG_SPLIT_ISA=1"
build isa1_noslp "-DPVSG_SPLIT_ISA=1 -fno-slp-vectorize"
{
echo "== correctness of the v_fma_mix form (tests/test_gemm_f16x2.py through PVSG_LIB_PATH)"
PVSG_LIB_PATH=/tmp/libpvsg_isa1.so python -m pytest tests/test_gemm_f16x2.py -q -m gpu 2>&1 | tail -1
for v in isa0 isa0_noslp isa1 isa1_noslp; do
  echo "== $v"
  if [ $v = isa0 ]; then unset PVSG_LIB_PATH; else export PVSG_LIB_PATH=/tmp/libpvsg_$v.so; fi
  python scripts/lab/gemm_tile_ab.py 32 2>/dev/null | tail -1
  for sh in c256_64 c512_128 c256_1024; do python scripts/lab/power_probe_conv.py $sh 2 2>/dev/null | tail -1; done
done
unset PVSG_LIB_PATH
} 2>&1 | tee $O/split_lab.txt
