#!/bin/bash
# Variant build of the library for same-box A/B runs: tools/build_variant.sh <out.so> <file.hip> [-DNAME=value ...]
# Recompiles ONE source with extra defines against the in-tree objects of the others (manipose_amd/csrc/_obj must be up to date).
set -euo pipefail
OUT="$1"; SRC="$2"; shift 2
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/../manipose_amd/csrc" && pwd)"
TMP="$(mktemp -d)"
cp "$HERE"/_obj/*.o "$TMP"/
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-slp-vectorize -Xclang -target-feature -Xclang -packed-fp32-ops -Wno-unused-function "$@" \
  -c "$HERE/$SRC" -o "$TMP/$(basename "${SRC%.hip}").o" 2> >(grep -v "is not a recognized feature for this target" >&2)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$TMP"/*.o
rm -rf "$TMP"
echo "built $OUT"
