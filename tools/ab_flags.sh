#!/bin/bash
# Same-box A/B of two bench.py configurations of the in-tree library: alternating runs with the flag sets A and B.
# Usage: tools/ab_flags.sh "<flags A>" "<flags B>" [rounds]      e.g.  tools/ab_flags.sh "" "--f16f8 1 --f16-backward"
set -u
A="$1"; B="$2"; ROUNDS="${3:-2}"
for r in $(seq 1 "$ROUNDS"); do
  for v in "$A" "$B"; do
    timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --no-power $v 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('parity') or {}
print('[$v]', round(d['ms_per_step'],2), 'ms/step', round(d['value']), 'poses/s  parity', p.get('mpjpe_m'), ' gemm_fwd', round(d['kernel_classes']['gemm_fwd']['isolated_ms_per_step'],2), 'ln', round(d['kernel_classes']['layernorm']['isolated_ms_per_step'],2))"
  done
done
