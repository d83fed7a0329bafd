#!/bin/bash
# alternating bench.py runs of the in-tree library and of the libraries given as arguments (same box): ms/step of each run
set -u
ROUNDS="${ROUNDS:-3}"
for r in $(seq 1 "$ROUNDS"); do
  for which in tree "$@"; do
    if [ "$which" = tree ]; then unset MANIPOSE_HIP_LIB; else export MANIPOSE_HIP_LIB="$PWD/$which"; fi
    timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which', round(d['ms_per_step'],2), 'ms/step')"
  done
done
