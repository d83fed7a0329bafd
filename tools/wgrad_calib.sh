#!/bin/bash
# calibration of the FETCH_SIZE correction for direct-to-LDS loads: a weight-gradient GEMM with ONE output tile per split (256 x 256)
# has no operand shared between workgroups, so its fabric reads must equal the algorithmic bytes
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for s in "256 256" "512 256" "256 512"; do
  rm -rf gpurun_out/wg
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/wg -- python tools/wgrad_probe.py 326349 $s 6 > gpurun_out/wg.log 2>&1
  grep "^wgrad" gpurun_out/wg.log
  python tools/pmc_avg.py "$(find gpurun_out/wg -name '*counter_collection.csv' | head -1)" | grep "glds_kernel" | cut -c1-120
done
rm -rf gpurun_out/wg
