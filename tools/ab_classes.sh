#!/bin/bash
# Same-box A/B of kernel-class times with every kernel on ONE queue (isolated kernel durations): in-tree library against the libraries given.
# Usage: tools/ab_classes.sh other.so [more.so ...]   (ROUNDS=n)
set -u
ROUNDS="${ROUNDS:-2}"
for r in $(seq 1 "$ROUNDS"); do
  for which in tree "$@"; do
    if [ "$which" = tree ]; then unset MANIPOSE_HIP_LIB; else export MANIPOSE_HIP_LIB="$PWD/$which"; fi
    timeout -k 10 300 python bench.py --single-queue --steps 6 --warmup 2 --no-cpu-baseline --no-parity --no-extra --no-power 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_classes']
print('$which', round(d['ms_per_step'],2), 'ms/step', ' '.join(f'{n}={v[\"isolated_ms_per_step\"]:.2f}' for n,v in k.items()))"
  done
done
