#!/bin/bash
# Same-box A/B of the decoder's query-row kernels: the tree's form against scripts/lab/rows_pipeline_experiment.patch (self-attention
# K / V of all 100 keys requested at the kernel's start + single-pass soft-max; every GEMM's first ring of weight fragments requested
# one stage early: RowsMM start / run).  Record: profiles/r06_rows_pipeline_ab.txt -- not adopted (no gain in the step).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OBJ=$R/openpvsg_amd/lib/obj
rm -rf /tmp/exp && mkdir -p /tmp/exp/openpvsg_amd/csrc /tmp/exp/include
cp $R/openpvsg_amd/csrc/*.h $R/openpvsg_amd/csrc/decoder_rows.hip /tmp/exp/openpvsg_amd/csrc/ && cp $R/include/*.h /tmp/exp/include/
(cd /tmp/exp && patch -p1 -s < $R/scripts/lab/rows_pipeline_experiment.patch) || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -c /tmp/exp/openpvsg_amd/csrc/decoder_rows.hip -o /tmp/decoder_rows_exp.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $OBJ/*.o | grep -v "decoder_rows.o") /tmp/decoder_rows_exp.o -o /tmp/libpvsg_exp.so || exit 1
B="python $R/bench.py --cpu-baseline off --sub-benchmarks off --steps 30 --warmup 5"
for rep in 1 2; do
  for lib in "" /tmp/libpvsg_exp.so; do
    for T in 4 32; do
      echo "lib=${lib:-tree} T=$T: $(PVSG_LIB_PATH=$lib $B --frames $T 2>/dev/null | grep '^{' | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), round(d["value"],1))')"
    done
    echo "lib=${lib:-tree} image: $(PVSG_LIB_PATH=$lib B1_MODES=on python $R/scripts/lab/ips_image_breakdown.py 2>&1 | grep 'graph on')"
  done
done
