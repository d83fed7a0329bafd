cd "$GRAFT_REPO_ROOT"
export MANIPOSE_HIP_LIB=$PWD/manipose_amd/libmanipose_hip_diag.so
for s in 0 700 1400 2800 0 1400; do
  echo "== stagger $s ticks (10 ns) per phase group"
  MANIPOSE_GEMM_STAGGER=$s timeout -k 10 120 python tools/gemm_x3_bench.py 2>&1 | grep -v amdgpu.ids | sed -e 's/bf16 .*| x3/x3/'
done
