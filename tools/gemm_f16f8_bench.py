"""Micro-benchmark of the "f16f8" Linear forward (one fp16 + one block-scaled fp8 product per k-tile, mp_linear_fwd_f16f8) next to the shipped
split precision (three bf16 products, mp_linear_fwd_bf16x3) through the C ABI: the four Linear shapes of a MixSTE block, plain bias epilogue,
4 bytes of output per element in both (fp32 / planar bf16).  Operand planes hold random bytes of the right formats (timing only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manipose_amd import _lib
from gemm_bench import timeit

lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
M = int(sys.argv[1]) if len(sys.argv) > 1 else 326349
tot = {"x3": 0.0, "f16f8": 0.0}
for (N, K, name) in [(1536, 512, "qkv"), (512, 512, "proj"), (1024, 512, "fc1"), (512, 1024, "fc2")]:
    x = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda") / K ** 0.5
    xh, xl = torch.empty_like(x, dtype=torch.bfloat16), torch.empty_like(x, dtype=torch.bfloat16)
    Wh, Wl = torch.empty_like(W, dtype=torch.bfloat16), torch.empty_like(W, dtype=torch.bfloat16)
    lib.mp_split_bf16(x.data_ptr(), xh.data_ptr(), xl.data_ptr(), x.numel(), st)
    lib.mp_split_bf16(W.data_ptr(), Wh.data_ptr(), Wl.data_ptr(), W.numel(), st)
    x16, W16 = x.half(), W.half()
    # correction planes: e4m3 bytes of random values of the real magnitudes (lo parts scaled as the format says)
    x8 = torch.cat([((x - x16.float()) * 2.0 ** 11).view(M, K // 4, 4), x16.float().view(M, K // 4, 4)], dim=2).to(torch.float8_e4m3fn).view(torch.uint8).reshape(M, 2 * K).contiguous()
    W8 = torch.cat([(W16.float() * 16).view(N, K // 4, 4), ((W - W16.float()) * 2.0 ** 15).view(N, K // 4, 4)], dim=2).to(torch.float8_e4m3fn).view(torch.uint8).reshape(N, 2 * K).contiguous()
    del x
    b = torch.randn(N, device="cuda")
    yh, yl = torch.empty(M, N, device="cuda", dtype=torch.bfloat16), torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    y32 = torch.empty(M, N, device="cuda")
    p = lambda t: t.data_ptr() if t is not None else None
    t3 = timeit(lambda: lib.mp_linear_fwd_bf16x3(p(xh), p(xl), p(Wh), p(Wl), p(b), p(yh), p(yl), None, None, M, N, K, 0, st))
    t8 = timeit(lambda: lib.mp_linear_fwd_f16f8(p(x16), p(x8), p(W16), p(W8), p(b), p(y32), M, N, K, st))
    tot["x3"] += t3; tot["f16f8"] += t8
    fl = 2.0 * M * N * K
    print(f"{name:5s} M={M} N={N} K={K}: x3 {t3 * 1e3:7.1f} us {fl / t3 / 1e9:6.1f} TF algorithmic | f16f8 {t8 * 1e3:7.1f} us {fl / t8 / 1e9:6.1f} TF algorithmic  (x{t3 / t8:.2f})", flush=True)
print(f"block total: x3 {tot['x3'] * 1e3:.0f} us, f16f8 {tot['f16f8'] * 1e3:.0f} us; x16 blocks = {tot['x3'] * 16:.1f} / {tot['f16f8'] * 16:.1f} ms")
