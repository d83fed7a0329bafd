"""Times the bf16 dgrad GEMMs of a block (dx = dy W through the persistent kernel) at the bench's token count: mp_linear_bwd_bf16 with and
without dx, the difference is the dgrad launch.  python tools/dgrad_bench.py [windows]"""
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
    W = torch.randn(N, K, device="cuda").bfloat16(); dx = torch.empty(M, K, device="cuda", dtype=torch.bfloat16)
    dW, db = torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda")
    slab = torch.empty(int(lib.mp_linear_bwd_slab_floats(N, K)), device="cuda")
    run = lambda dxp: lib.mp_linear_bwd_bf16(dy.data_ptr(), 0, x.data_ptr(), W.data_ptr(), dxp, 0, dW.data_ptr(), db.data_ptr(), M, N, K, slab.data_ptr(), slab.numel(), st)
    t_w = timeit(lambda: run(None)); t_b = timeit(lambda: run(dx.data_ptr()))
    t = t_b - t_w
    tot += t
    print(f"{name:5s} dy[{M},{N}] W[{N},{K}]: dgrad {t * 1e3:7.1f} us  {2.0 * M * N * K / t / 1e9:6.0f} TF/s   (wgrad alone {t_w * 1e3:.1f} us)", flush=True)
print(f"block total {tot * 1e3:.0f} us (the fc2 dgrad of the engine also multiplies by gelu': not this entry)")
