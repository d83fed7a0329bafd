cd "$GRAFT_REPO_ROOT"
for r in 1 2; do
  for lib in tree .ab/attn_exp.so; do
    if [ "$lib" = tree ]; then unset MANIPOSE_HIP_LIB; else export MANIPOSE_HIP_LIB=$PWD/$lib; fi
    echo "== $lib"
    B=79 timeout -k 10 120 python tools/attn_bench.py 2>&1 | grep "bf16x3"
  done
done
export MANIPOSE_HIP_LIB=$PWD/.ab/attn_exp.so
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "bf16x3_attention_forward or bf16x3_full_size or persistent_kernels_inside" 2>&1 | tail -3
