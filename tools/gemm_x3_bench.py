"""Micro-benchmark of the split-precision (bf16x3) forward GEMMs through the C ABI next to their bf16 counterparts, the four Linear shapes
of a MixSTE block with the epilogue the engine uses for each.  TFLOP/s of the x3 rows = ISSUED matrix-core flops (6 M N K)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manipose_amd import _lib
from gemm_bench import timeit

lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
M = int(sys.argv[1]) if len(sys.argv) > 1 else 326349
tot = {"bf16": 0.0, "x3": 0.0}
for (N, K, name, epi) in [(1536, 512, "qkv", 0), (512, 512, "proj", 2), (1024, 512, "fc1", 1), (512, 1024, "fc2", 2)]:
    x = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda") / K ** 0.5
    xh, xl = torch.empty_like(x, dtype=torch.bfloat16), torch.empty_like(x, dtype=torch.bfloat16)
    Wh, Wl = torch.empty_like(W, dtype=torch.bfloat16), torch.empty_like(W, dtype=torch.bfloat16)
    lib.mp_split_bf16(x.data_ptr(), xh.data_ptr(), xl.data_ptr(), x.numel(), st)
    lib.mp_split_bf16(W.data_ptr(), Wh.data_ptr(), Wl.data_ptr(), W.numel(), st)
    del x
    b = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda") if epi == 2 else None
    yh, yl = torch.empty(M, N, device="cuda", dtype=torch.bfloat16), torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    y32 = torch.empty(M, N, device="cuda") if epi == 2 else None
    z = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if epi == 1 else None
    p = lambda t: t.data_ptr() if t is not None else None
    t1 = timeit(lambda: lib.mp_linear_fwd_bf16(p(xh), p(Wh), p(b), p(y32) if epi == 2 else p(yh), p(z), p(r), M, N, K, epi, st))
    t3 = timeit(lambda: lib.mp_linear_fwd_bf16x3(p(xh), p(xl), p(Wh), p(Wl), p(b), p(y32) if epi == 2 else p(yh), p(yl), p(z), p(r), M, N, K, epi, st))
    tot["bf16"] += t1; tot["x3"] += t3
    fl = 2.0 * M * N * K
    print(f"{name:5s} M={M} N={N} K={K} epi={epi}: bf16 {t1 * 1e3:7.1f} us {fl / t1 / 1e9:6.1f} TF | x3 {t3 * 1e3:7.1f} us {3 * fl / t3 / 1e9:6.1f} TF issued  (x{t3 / t1:.2f})", flush=True)
print(f"block total: bf16 {tot['bf16'] * 1e3:.0f} us, x3 {tot['x3'] * 1e3:.0f} us; x16 blocks = {tot['bf16'] * 16:.1f} / {tot['x3'] * 16:.1f} ms")
