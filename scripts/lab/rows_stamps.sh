#!/bin/bash
# Stage times of decoder_rows_post (one clip, 100 queries): lab build with -DPVSG_ROWS_STAMPS=1, scripts/lab/rows_stamps.py reads the stamps.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OBJ=$R/openpvsg_amd/lib/obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DPVSG_ROWS_STAMPS=1 "$@" -c $R/openpvsg_amd/csrc/decoder_rows.hip -o /tmp/decoder_rows_stamps.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $OBJ/*.o | grep -v "decoder_rows.o") /tmp/decoder_rows_stamps.o -o /tmp/libpvsg_stamps.so || exit 1
PVSG_LIB_PATH=/tmp/libpvsg_stamps.so python $R/scripts/lab/rows_stamps.py
