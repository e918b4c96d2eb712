#!/bin/bash
# Build a variant of the library with extra compiler flags for some sources (measurement only):
#   tools/build_variant.sh NAME "FLAGS" file1.hip [file2.hip ...]   ->  auto_oo_amd/lib/liboovqe_hip_NAME.so
# (run it with OOVQE_LIB_PATH=auto_oo_amd/lib/liboovqe_hip_NAME.so)
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
NAME="$1"; FLAGS="$2"; shift 2
SRC="$ROOT/auto_oo_amd/csrc"; OBJ="$SRC/obj"; VOBJ="$SRC/obj/variant_$NAME"
mkdir -p "$VOBJ"
objs=()
for o in "$OBJ"/*.o; do
    base="$(basename "${o%.o}")"
    skip=0
    for f in "$@"; do [ "${f%.hip}" = "$base" ] && skip=1; done
    [ $skip -eq 0 ] && objs+=("$o")
done
pids=()
for f in "$@"; do
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-function $FLAGS -c "$SRC/$f" -o "$VOBJ/${f%.hip}.o" &
    pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-z,defs -o "$ROOT/auto_oo_amd/lib/liboovqe_hip_$NAME.so" "${objs[@]}" "$VOBJ"/*.o
echo "built auto_oo_amd/lib/liboovqe_hip_$NAME.so"
