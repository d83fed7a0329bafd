"""Reference point: what the vendor library (torch.matmul -> hipBLASLt / rocBLAS) reaches on the plain GEMM shapes of the path."""
import torch
M = 264384
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
for N, K, name in [(1536, 512, "qkv fwd"), (512, 1536, "qkv dgrad"), (512, 512, "proj"), (1024, 512, "fc1"), (512, 1024, "fc2")]:
    x = torch.randn(M, K, device="cuda").bfloat16(); W = torch.randn(N, K, device="cuda").bfloat16(); b = torch.randn(N, device="cuda").bfloat16()
    ms = t(lambda: torch.nn.functional.linear(x, W, b))
    ms2 = t(lambda: torch.matmul(x, W.t()))
    print(f"{name:10s} M={M} N={N} K={K}: linear+bias {2*M*N*K/ms/1e9:7.1f} TF   matmul {2*M*N*K/ms2/1e9:7.1f} TF")
