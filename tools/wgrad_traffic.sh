#!/bin/bash
# fabric read traffic of the isolated weight-gradient GEMM per Linear shape (FETCH_SIZE x 2, MI355X_MICROARCH.md HBM section)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
M="${1:-326349}"
for s in "1536 512" "512 512" "1024 512" "512 1024"; do
  rm -rf gpurun_out/wg
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/wg -- python tools/wgrad_probe.py $M $s 6 > gpurun_out/wg.log 2>&1
  grep "^wgrad" gpurun_out/wg.log
  python tools/pmc_avg.py "$(find gpurun_out/wg -name '*counter_collection.csv' | head -1)" | grep "glds_kernel"
done
rm -rf gpurun_out/wg
