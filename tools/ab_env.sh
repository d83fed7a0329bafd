#!/bin/bash
# Same-box A/B of an environment switch of the in-tree library: alternating bench.py runs with VAR=a and VAR=b.  Usage: tools/ab_env.sh VAR a b [rounds]
set -u
VAR="$1"; A="$2"; B="$3"; ROUNDS="${4:-2}"
for r in $(seq 1 "$ROUNDS"); do
  for v in "$A" "$B"; do
    env "$VAR=$v" timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('parity') or {}
print('$VAR=$v', round(d['ms_per_step'],2), 'ms/step', round(d['value']), 'poses/s  parity', p.get('mpjpe_m'), ' gemm_fwd', round(d['kernel_classes']['gemm_fwd']['ms_per_step'],2), 'ln', round(d['kernel_classes']['layernorm']['ms_per_step'],2))"
  done
done
