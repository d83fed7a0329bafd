"""Times the split-K weight-gradient GEMM alone (mp_linear_bwd_bf16 with dx = NULL) at the bench's token count.
python tools/wgrad_bench.py [windows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manipose_amd import _lib
from gemm_bench import timeit
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
B = int(sys.argv[1]) if len(sys.argv) > 1 else 79
M = B * 243 * 17
tot = 0.0
for (N, K, name) in [(1536, 512, "qkv"), (512, 512, "proj"), (1024, 512, "fc1"), (512, 1024, "fc2")]:
    x = torch.randn(M, K, device="cuda").bfloat16(); dy = torch.randn(M, N, device="cuda").bfloat16()
    W = torch.randn(N, K, device="cuda").bfloat16()
    dW, db = torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda")
    slab = torch.empty(int(lib.mp_linear_bwd_slab_floats(N, K)), device="cuda")
    t = timeit(lambda: lib.mp_linear_bwd_bf16(dy.data_ptr(), 0, x.data_ptr(), W.data_ptr(), None, 0, dW.data_ptr(), db.data_ptr(), M, N, K, slab.data_ptr(), slab.numel(), st))
    tot += t
    print(f"{name:5s} N={N} K={K}: wgrad {t * 1e3:7.1f} us  {2.0 * M * N * K / t / 1e9:6.0f} TF/s", flush=True)
print(f"block total {tot * 1e3:.0f} us (dbg={os.environ.get('MANIPOSE_GEMM_DEBUG', '0')})")
