#!/bin/bash
# Output-heads A/B on one box: the model-level GPU tests that exercise the heads, then the default bench under rocprofv3 with the row
# kernels (--option heads_mfma=0) and with the matrix-core kernels (=1); prints ms/step and the heads / scores kernel lines of each run.
# Usage (from the repo root, on the GPU box): bash tools/heads_ab.sh
set -e
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "heads or rmcl or manifold or full_size or mix_ste or train or grad or mup or config" > gpurun_out/heads_tests.log 2>&1 || { tail -n 30 gpurun_out/heads_tests.log; exit 1; }
tail -n 3 gpurun_out/heads_tests.log
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/hm$v -- python /root/repo/bench.py --option heads_mfma=$v --steps 4 --warmup 2 --no-cpu-baseline --no-parity --no-extra --no-power > /root/repo/gpurun_out/hm$v.log 2>&1
  tail -n 1 /root/repo/gpurun_out/hm$v.log | cut -c1-200
  f=$(find /root/repo/gpurun_out/hm$v -name "*kernel_stats.csv" | head -n 1)
  grep -i "heads\|scores" $f | cut -c1-160
done
