#!/bin/bash
# Same-box A/B of library builds on the four split-precision forward GEMMs of a block (tools/gemm_x3_bench.py), alternating, ROUNDS rounds:
#   tools/ab_gemm_block.sh <a.so> <b.so> ...     ("base" = the in-tree library)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
ROUNDS="${ROUNDS:-2}"
for r in $(seq 1 "$ROUNDS"); do
  for lib in "$@"; do
    if [ "$lib" = "base" ]; then unset MANIPOSE_HIP_LIB; else export MANIPOSE_HIP_LIB="$PWD/$lib"; fi
    echo "== round $r $lib"
    timeout -k 10 120 python tools/gemm_x3_bench.py 2>&1 | sed -e 's/bf16 .*| x3/x3/'
  done
done
