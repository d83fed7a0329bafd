"""Isolated split-K weight-gradient GEMM of one Linear shape (for rocprofv3 --pmc / --kernel-trace): python tools/wgrad_probe.py M N K [reps]."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manipose_amd import _lib

lib = _lib.load()
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
st = torch.cuda.current_stream().cuda_stream
x = torch.randn(M, K, device="cuda").bfloat16()
W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
dy = torch.randn(M, N, device="cuda").bfloat16()
dW, db = torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda")
slab = torch.empty(int(lib.mp_linear_bwd_slab_floats(N, K)), device="cuda")
run = lambda: _lib.check(lib.mp_linear_bwd_bf16(dy.data_ptr(), 0, x.data_ptr(), W.data_ptr(), None, 0, dW.data_ptr(), db.data_ptr(), M, N, K,
                                                slab.data_ptr(), slab.numel(), st))
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"wgrad M={M} N={N} K={K}: {ms*1e3:.1f} us (with slab reduce), {2.0*M*N*K/ms/1e9:.1f} TF, algorithmic reads {(N+K)*M*2/1e6:.1f} MB", flush=True)
