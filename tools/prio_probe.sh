#!/bin/bash
# Queue-priority probe: the default bench with the side streams (bones net, weight gradients) at normal / low / high queue priority.
# Usage (GPU box, repo root): bash tools/prio_probe.sh [rounds]
for r in $(seq 1 "${1:-2}"); do
  for p in 0 11 10 1 22 20; do
    MANIPOSE_SIDE_PRIORITY=$p timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-parity --no-extra --no-prof 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('priority $p:', round(d['ms_per_step'],2), 'ms/step')"
  done
done
