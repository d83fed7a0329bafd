"""Probe: does an HBM-bound stream (copy) overlap with the MFMA main loop of the tiled GEMM on the same CUs?
Run with MANIPOSE_GEMM_PERSIST=0 MANIPOSE_GEMM_DEBUG=6 (GEMM = main loop only, no DMA, no epilogue)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manipose_amd import _lib

lib = _lib.load()
M, N, K = 264384, 1536, 512
x = torch.randn(M, K, device="cuda").bfloat16()
W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
b = torch.randn(N, device="cuda")
y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
src = torch.randn(M, 768, device="cuda")          # 812 MB
dst = torch.empty_like(src)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(gemm, copy, n=10):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
    for _ in range(n):
        if gemm:
            with torch.cuda.stream(s1):
                lib.mp_linear_fwd_bf16(x.data_ptr(), W.data_ptr(), b.data_ptr(), y.data_ptr(), None, None, M, N, K, 0, s1.cuda_stream)
        if copy:
            with torch.cuda.stream(s2):
                dst.copy_(src)
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for _ in range(2):
    run(True, True, 3)
print(f"gemm alone {run(True, False):.0f} us   copy alone {run(False, True):.0f} us   both {run(True, True):.0f} us")
