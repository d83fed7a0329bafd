#!/bin/bash
# A/B of two builds of libmanipose_hip.so on ONE box: alternates bench.py runs of the in-tree library and of $1 (another build),
# prints ms/step of each run.  Usage: tools/ab_bench.sh .ab/prev/manipose_amd/libmanipose_hip.so [rounds]
set -u
OTHER="$1"; ROUNDS="${2:-2}"
for r in $(seq 1 "$ROUNDS"); do
  for which in tree other; do
    if [ "$which" = other ]; then export MANIPOSE_HIP_LIB="$PWD/$OTHER"; else unset MANIPOSE_HIP_LIB; fi
    timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which', round(d['ms_per_step'],2), 'ms/step', round(d['value']), 'poses/s')"
  done
done
