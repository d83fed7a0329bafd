#!/bin/bash
# alternating tools/attn_bench.py runs (split-precision forward lines) of the in-tree library and of the builds given; then the attention parity tests
# of the in-tree library.  Usage (GPU box, repo root): tools/probes/run_attn_ab.sh other.so [...]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
for r in 1 2; do
  for lib in tree "$@"; do
    if [ "$lib" = tree ]; then unset MANIPOSE_HIP_LIB; else export MANIPOSE_HIP_LIB=$PWD/$lib; fi
    echo "== $lib"
    B=79 timeout -k 10 120 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids
  done
done
unset MANIPOSE_HIP_LIB
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "attention or bf16x3_full_size or persistent_kernels_inside" 2>&1 | tail -3
