#!/bin/bash
# probe: L2 (TCC) request / hit / miss / fabric-read counters of the isolated weight-gradient GEMM
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for s in "256 256" "512 512" "1536 512"; do
  for c in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_READ_sum"; do
    rm -rf gpurun_out/wg
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/wg -- python tools/wgrad_probe.py 326349 $s 4 > gpurun_out/wg.log 2>&1
    f="$(find gpurun_out/wg -name '*counter_collection.csv' | head -1)"
    if [ -n "$f" ]; then python tools/pmc_avg.py "$f" | grep "glds_kernel" | cut -c1-60 | sed "s/^/$s: /"; else tail -3 gpurun_out/wg.log; fi
  done
done
rm -rf gpurun_out/wg
