"""Micro-benchmark of the LayerNorm row kernels through the C ABI (one MI355X): achieved HBM GB/s."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manipose_amd import _lib

lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 264384
    C = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    x = torch.randn(M, C, device="cuda")
    g, b = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
    y, stats = torch.empty_like(x), torch.empty(M, 2, device="cuda")
    dy, dx, dskip = torch.randn_like(x), torch.empty_like(x), torch.randn_like(x)
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    scratch = torch.empty(1024 * 2 * C + 16, device="cuda")
    t = timeit(lambda: lib.mp_layernorm_fwd(x.data_ptr(), g.data_ptr(), b.data_ptr(), 1e-6, y.data_ptr(), stats.data_ptr(), M, C, st))
    print(f"ln_fwd  fp32->fp32  M={M} C={C}: {t * 1e3:7.1f} us  {2 * x.numel() * 4 / t / 1e6:7.0f} GB/s")
    t = timeit(lambda: lib.mp_layernorm_bwd(dy.data_ptr(), x.data_ptr(), stats.data_ptr(), g.data_ptr(), None, dx.data_ptr(), dg.data_ptr(),
                                            db.data_ptr(), M, C, scratch.data_ptr(), scratch.numel(), st))
    print(f"ln_bwd  (no skip)   M={M} C={C}: {t * 1e3:7.1f} us  {3 * x.numel() * 4 / t / 1e6:7.0f} GB/s")
    t = timeit(lambda: lib.mp_layernorm_bwd(dy.data_ptr(), x.data_ptr(), stats.data_ptr(), g.data_ptr(), dskip.data_ptr(), dx.data_ptr(),
                                            dg.data_ptr(), db.data_ptr(), M, C, scratch.data_ptr(), scratch.numel(), st))
    print(f"ln_bwd  (+ skip)    M={M} C={C}: {t * 1e3:7.1f} us  {4 * x.numel() * 4 / t / 1e6:7.0f} GB/s")
    t = timeit(lambda: y.copy_(x))
    print(f"torch copy          M={M} C={C}: {t * 1e3:7.1f} us  {2 * x.numel() * 4 / t / 1e6:7.0f} GB/s")


if __name__ == "__main__":
    main()
