#!/bin/bash
# fabric reads (FETCH_SIZE x 2) of the forward / dgrad GEMM kernels in tools/gemm_bench.py, for MANIPOSE_GEMM_DEBUG values given as arguments
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  export MANIPOSE_GEMM_DEBUG=$v
  rm -rf gpurun_out/pt
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pt -- python tools/gemm_bench.py 326349 > gpurun_out/pt.log 2>&1
  python tools/pmc_avg.py "$(find gpurun_out/pt -name '*counter_collection.csv' | head -1)" | grep "persist" | cut -c1-150 | sed "s/^/debug=$v /"
done
rm -rf gpurun_out/pt
