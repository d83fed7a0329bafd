for lib in tree .ab/late1.so .ab/late3.so .ab/late5.so; do
  if [ "$lib" = tree ]; then unset MANIPOSE_HIP_LIB; else export MANIPOSE_HIP_LIB="$PWD/$lib"; fi
  echo "== $lib"; timeout -k 10 200 python tools/dgrad_bench.py 2>&1 | grep -v amdgpu.ids
done
