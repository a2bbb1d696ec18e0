#!/bin/bash
# 3x3 halo kernel: weight slabs asked for one tap ahead (2 buffers, product) vs two taps ahead (3 buffers, PVSG_HALO_NBUF=3)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
OBJ=$R/openpvsg_amd/lib/obj
O=gpurun_out/r05_halo_nbuf; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DPVSG_HALO_NBUF=3 -c $R/openpvsg_amd/csrc/conv3x3_halo.hip -o /tmp/conv3x3_halo_nbuf3.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $OBJ/*.o | grep -v conv3x3_halo.o) /tmp/conv3x3_halo_nbuf3.o -o /tmp/libpvsg_nbuf3.so || exit 1
{
PVSG_LIB_PATH=/tmp/libpvsg_nbuf3.so python -m pytest tests/test_winograd.py tests/test_glue_kernels.py -q -m gpu 2>&1 | tail -1
for v in base nbuf3 base nbuf3; do
  if [ $v = base ]; then unset PVSG_LIB_PATH; else export PVSG_LIB_PATH=/tmp/libpvsg_$v.so; fi
  python scripts/lab/halo_time.py 2>/dev/null | tail -1
done
unset PVSG_LIB_PATH
} 2>&1 | tee $O/halo_nbuf.txt
