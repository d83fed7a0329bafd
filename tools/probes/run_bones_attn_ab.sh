cd "$GRAFT_REPO_ROOT"
for r in 1 2; do
  for lib in tree .ab/attn_nofull.so; do
    if [ "$lib" = tree ]; then unset MANIPOSE_HIP_LIB; else export MANIPOSE_HIP_LIB=$PWD/$lib; fi
    echo "== $lib"
    timeout -k 10 120 python tools/bones_bench.py 79 2>&1 | grep -i "attention"
  done
done
