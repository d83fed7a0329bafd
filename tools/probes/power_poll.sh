#!/bin/bash
# Samples rocm-smi (socket power, shader clock) every 0.25 s while a command runs:   tools/probes/power_poll.sh <log> <command ...>
# (which kernels of the step run at the 1400 W cap: round 5, DESIGN section 5 "power")
LOG="$1"; shift
"$@" > "$LOG.cmd" 2>&1 &
PID=$!
: > "$LOG"
while kill -0 $PID 2>/dev/null; do
  /opt/rocm/bin/rocm-smi --showclocks --showpower --csv 2>/dev/null | grep "^card0" | awk -F, '{print $6, $10}' >> "$LOG"
  sleep 0.25
done
wait $PID
python3 - "$LOG" <<'PY'
import re, sys
rows = [l.split() for l in open(sys.argv[1]) if l.strip()]
v = [(int(re.sub(r"\D", "", a)), float(b)) for a, b in rows]
busy = [x for x in v if x[1] > 600]
if busy:
    print(f"{len(v)} samples, {len(busy)} above 600 W: mean power {sum(b for _, b in busy) / len(busy):.0f} W (max {max(b for _, b in busy):.0f}), mean shader clock {sum(a for a, _ in busy) / len(busy):.0f} MHz (min {min(a for a, _ in busy)}, max {max(a for a, _ in busy)})")
PY
