#!/bin/bash
# Build liboovqe_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT" "$HERE/obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wall -Wno-unused-function"
pids=()
for src in "$HERE"/*.hip; do
    obj="$HERE/obj/$(basename "${src%.hip}").o"
    if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ "$HERE/common.h" -nt "$obj" ] || [ "$HERE/circuit_small.h" -nt "$obj" ] || [ "$HERE/../../include/oovqe.h" -nt "$obj" ]; then
        $HIPCC $FLAGS -c "$src" -o "$obj" &
        pids+=($!)
    fi
done
rc=0
for p in "${pids[@]:-}"; do
    if [ -n "$p" ]; then wait "$p" || rc=1; fi
done
[ $rc -eq 0 ] || { echo "compile failed" >&2; exit 1; }
# -z defs: an undefined symbol (e.g. a kernel stub the host pass silently dropped) fails the build
$HIPCC --offload-arch=gfx950 -shared -fPIC -Wl,-z,defs -o "$OUT/liboovqe_hip.so" "$HERE"/obj/*.o
echo "built $OUT/liboovqe_hip.so"
