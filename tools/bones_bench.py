"""Isolated kernels of the bones net (C = 128, 16 tokens per frame, head dim 16) at the bench's token count through the C ABI: where its
~10 ms per step go.  python tools/bones_bench.py [windows]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manipose_amd import _lib
from gemm_bench import timeit

lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
B = int(sys.argv[1]) if len(sys.argv) > 1 else 79
T, J, C, H = 243, 16, 128, 8
M = B * T * J
x = torch.randn(M, C, device="cuda")
g, b = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
y, stats = torch.empty_like(x), torch.empty(M, 2, device="cuda")
dy, dx, dskip = torch.randn_like(x), torch.empty_like(x), torch.randn_like(x)
dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
scratch = torch.empty(1024 * 2 * C + 16, device="cuda")
t = timeit(lambda: lib.mp_layernorm_fwd(x.data_ptr(), g.data_ptr(), b.data_ptr(), 1e-6, y.data_ptr(), stats.data_ptr(), M, C, st))
print(f"ln_fwd fp32 C={C}: {t * 1e3:7.1f} us  {2 * x.numel() * 4 / t / 1e6:6.0f} GB/s")
t = timeit(lambda: lib.mp_layernorm_bwd(dy.data_ptr(), x.data_ptr(), stats.data_ptr(), g.data_ptr(), dskip.data_ptr(), dx.data_ptr(), dg.data_ptr(),
                                        db.data_ptr(), M, C, scratch.data_ptr(), scratch.numel(), st))
print(f"ln_bwd (+skip) C={C}: {t * 1e3:7.1f} us  {4 * x.numel() * 4 / t / 1e6:6.0f} GB/s")
for (N, K, name, epi) in [(384, 128, "qkv", 0), (128, 128, "proj", 2), (256, 128, "fc1", 1), (128, 256, "fc2", 2)]:
    a = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / K ** 0.5
    ah, al = torch.empty_like(a, dtype=torch.bfloat16), torch.empty_like(a, dtype=torch.bfloat16)
    Wh, Wl = torch.empty_like(W, dtype=torch.bfloat16), torch.empty_like(W, dtype=torch.bfloat16)
    lib.mp_split_bf16(a.data_ptr(), ah.data_ptr(), al.data_ptr(), a.numel(), st); lib.mp_split_bf16(W.data_ptr(), Wh.data_ptr(), Wl.data_ptr(), W.numel(), st)
    bias = torch.randn(N, device="cuda"); r = torch.randn(M, N, device="cuda") if epi == 2 else None
    yh, yl = torch.empty(M, N, device="cuda", dtype=torch.bfloat16), torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    y32 = torch.empty(M, N, device="cuda") if epi == 2 else None
    z = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if epi == 1 else None
    p = lambda t_: t_.data_ptr() if t_ is not None else None
    t3 = timeit(lambda: lib.mp_linear_fwd_bf16x3(p(ah), p(al), p(Wh), p(Wl), p(bias), p(y32) if epi == 2 else p(yh), p(yl), p(z), p(r), M, N, K, epi, st))
    byt = 4 * M * K + (8 * M * N if epi == 2 else 4 * M * N + (2 * M * N if epi == 1 else 0))
    dyb = torch.randn(M, N, device="cuda").bfloat16(); dxb = torch.empty(M, K, device="cuda", dtype=torch.bfloat16)
    dW, dbias = torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda")
    slab = torch.empty(int(lib.mp_linear_bwd_slab_floats(N, K)) + 64 * (N * K + N), device="cuda")
    tb = timeit(lambda: lib.mp_linear_bwd_bf16(dyb.data_ptr(), 0, ah.data_ptr(), Wh.data_ptr(), dxb.data_ptr(), 0, dW.data_ptr(), dbias.data_ptr(), M, N, K,
                                               slab.data_ptr(), slab.numel(), st))
    print(f"{name:5s} N={N} K={K}: x3 fwd {t3 * 1e3:7.1f} us ({byt / t3 / 1e6:5.0f} GB/s of algorithmic bytes) | bf16 dgrad+wgrad {tb * 1e3:7.1f} us")
qkv = torch.randn(M, 3 * C, device="cuda").bfloat16(); ql = (torch.randn(M, 3 * C, device="cuda") * 2 ** -8).bfloat16()
out, ol = torch.empty(M, C, device="cuda", dtype=torch.bfloat16), torch.empty(M, C, device="cuda", dtype=torch.bfloat16)
dout = torch.randn(M, C, device="cuda").bfloat16(); dq = torch.empty(M, 3 * C, device="cuda", dtype=torch.bfloat16)
lse = torch.empty(B * J * H * T, device="cuda"); delta = torch.empty_like(lse)
for temporal in (1, 0):
    f3 = timeit(lambda: lib.mp_attention_fwd_bf16x3(qkv.data_ptr(), ql.data_ptr(), out.data_ptr(), ol.data_ptr(), lse.data_ptr(), None, temporal, B, T, J, C, H, st))
    bw = timeit(lambda: lib.mp_attention_bwd_bf16(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), delta.data_ptr(), dq.data_ptr(), temporal, B, T, J, C, H, st))
    print(f"attention {'temporal' if temporal else 'spatial '} D=16: x3 fwd {f3 * 1e3:7.1f} us ({M * C * 16 / f3 / 1e6:5.0f} GB/s) | bf16 bwd {bw * 1e3:7.1f} us ({M * C * 16 / bw / 1e6:5.0f} GB/s)")
