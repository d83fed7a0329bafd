#!/bin/bash
# poses/s of the default bench at several windows-per-GPU: the persistent GEMM walks ceil(B*4131/256) row panels x N/256 column tiles in
# rounds of 256 workgroups, so batch sizes whose tile counts land just under a multiple of 256 waste no last round.
set -u
for b in "$@"; do
  timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-other-configs --batch "$b" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B', $b, round(d['ms_per_step'],2), 'ms/step', round(d['value']), 'poses/s', 'persist frac', round(d['roofline']['frac'],4), 'parity', d.get('parity',{}).get('mpjpe_m'), 'within_bound', d.get('parity',{}).get('within_bound'), 'workspace GiB', round(d['workspace_gib'],1))"
done
