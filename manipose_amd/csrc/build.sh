#!/bin/bash
# Build libmanipose_hip.so for gfx950 (MI355X) in-tree.  Usage: manipose_amd/csrc/build.sh [-j N]
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../libmanipose_hip.so"
OBJ="$HERE/_obj"
mkdir -p "$OBJ"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function"
pids=()
for f in "$HERE"/*.hip; do
  o="$OBJ/$(basename "${f%.hip}").o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ "$HERE/common.h" -nt "$o" ] || [ "$HERE/kernels.h" -nt "$o" ] || [ "$HERE/../../include/manipose_hip.h" -nt "$o" ]; then
    $HIPCC $FLAGS -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$OBJ"/*.o
echo "built $OUT"
