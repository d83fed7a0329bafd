#!/bin/bash
# Isolated kernel durations inside the real step: every kernel on ONE queue (bench.py --single-queue = mp_model_config::streams 3), rocprofv3 kernel stats.
# Usage (GPU box): tools/single_queue_stats.sh <tag> [extra bench.py flags, e.g. --f16f8 1 --f16-backward]
set -eu
TAG="$1"; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/sq_$TAG; rm -rf "$O"; mkdir -p "$O"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python bench.py --single-queue --steps 3 --warmup 2 --no-cpu-baseline --no-parity --no-extra --no-power --no-prof "$@" > "$O/stats.log" 2>&1
find "$O/stats" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$O/kernel_stats.csv"
rm -rf "$O/stats"
python - "$O/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    print(f"{r['Name'][:96]:96s} {int(r['Calls']):5d} {float(r['TotalDurationNs']) / 5e6:7.2f} ms/step {float(r['AverageNs']) / 1e3:8.1f} us")
print("kernel time per step", sum(float(r['TotalDurationNs']) for r in rows) / 5e6)
PY
