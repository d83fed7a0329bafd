#!/bin/bash
# Same-box A/B of library builds on the training step: tools/ab_bench_f8.sh "<bench flags>" <a.so|base> <b.so|base> ...   (ROUNDS rounds, alternating)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
FLAGS="$1"; shift
ROUNDS="${ROUNDS:-2}"
for r in $(seq 1 "$ROUNDS"); do
  for lib in "$@"; do
    if [ "$lib" = "base" ]; then unset MANIPOSE_HIP_LIB; else export MANIPOSE_HIP_LIB="$PWD/$lib"; fi
    timeout -k 10 200 python bench.py $FLAGS --no-cpu-baseline --no-extra --no-other-configs --no-power --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('round $r $lib [$FLAGS]', round(d['value']), 'poses/s', round(d['ms_per_step'],1), 'ms  parity', '%.2e' % d['parity']['mpjpe_m'], {k: round(v.get('isolated_ms_per_step',0),1) for k,v in d['kernel_classes'].items()})
"
  done
done
