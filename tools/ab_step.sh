#!/bin/bash
# Same-box A/B of library builds: for each build the three-queue training step (ms per step) and, with every kernel on ONE queue, the isolated
# kernel-class times.  Usage: tools/ab_step.sh tree other.so [more.so ...]   (ROUNDS=n, default 2; "tree" = the in-tree library)
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
ROUNDS="${ROUNDS:-2}"
for r in $(seq 1 "$ROUNDS"); do
  for which in "$@"; do
    if [ "$which" = tree ]; then unset MANIPOSE_HIP_LIB; else export MANIPOSE_HIP_LIB="$PWD/$which"; fi
    timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-extra --no-power 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which', 'step', round(d['ms_per_step'],2), 'ms', round(d['value']), 'poses/s')"
    timeout -k 10 200 python bench.py --single-queue --steps 6 --warmup 2 --no-cpu-baseline --no-parity --no-extra --no-power 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_classes']
print('$which', 'one queue', round(d['ms_per_step'],2), 'ms', ' '.join(f'{n}={v[\"isolated_ms_per_step\"]:.2f}' for n,v in k.items()))"
  done
done
