#!/bin/bash
# builds tools/probes/ln_bwd_probe.hip with each set of defines given as one argument ("" = as shipped) and runs it; on a GPU box from the repo root
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
F="--offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Xclang -target-feature -Xclang -packed-fp32-ops -Iinclude -w"
i=0
for defs in "$@"; do
  i=$((i + 1))
  hipcc $F $defs tools/probes/ln_bwd_probe.hip -o /tmp/ln_bwd_probe_$i 2> >(grep -v "recognized feature" >&2) || { echo "build failed: $defs"; continue; }
  echo "== [$defs]"
  timeout -k 10 60 /tmp/ln_bwd_probe_$i
done
