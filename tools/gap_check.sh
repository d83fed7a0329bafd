#!/bin/bash
# One short bench run under rocprofv3 --kernel-trace, then the largest idle gaps of the main queue (tools/gap_list.py) and the per-queue
# view of the step (tools/stream_timeline.py).  Usage (GPU box, repo root): bash tools/gap_check.sh
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/gapchk -- python /root/repo/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-parity --no-extra --no-power > /root/repo/gpurun_out/gapchk.log 2>&1
f=$(find /root/repo/gpurun_out/gapchk -name "*kernel_trace.csv" | head -n 1)
python /root/repo/tools/gap_list.py $f 4 | cut -c1-180
python /root/repo/tools/stream_timeline.py $f | head -5
