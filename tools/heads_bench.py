"""Output heads (K x [LayerNorm -> Linear]) in isolation: row kernels (impl 1) against the fp32 matrix-core kernels (impl 2), forward and
backward, at the bench size; results of the two are compared first.   python tools/heads_bench.py [B] [K] [O] [C]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manipose_amd import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 79
K = int(sys.argv[2]) if len(sys.argv) > 2 else 5
O = int(sys.argv[3]) if len(sys.argv) > 3 else 7
C = int(sys.argv[4]) if len(sys.argv) > 4 else 512
M = B * 243 * 17
lib = _lib.load()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g)
x = r(M, C) * 1.5 + 0.3
gamma, beta, W, b = 1 + 0.1 * r(K, C), 0.1 * r(K, C), r(K, O, C) / C ** 0.5, 0.1 * r(K, O)
dout = r(K, M, O)
st = torch.cuda.current_stream().cuda_stream
scratch = torch.empty(lib.mp_heads_bwd_scratch_floats(K, O, C), device=dev)


def run(impl):
    out, stats, fold = torch.empty(K, M, O, device=dev), torch.empty(M, 2, device=dev), torch.zeros(lib.mp_heads_fold_floats(C), device=dev)
    dx = torch.empty(M, C, device=dev)
    grads = [torch.zeros_like(t) for t in (gamma, beta, W, b)]
    fwd = lambda: _lib.check(lib.mp_heads_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), W.data_ptr(), b.data_ptr(), K, O, out.data_ptr(),
                                              stats.data_ptr(), fold.data_ptr(), M, C, impl, st))
    bwd = lambda: _lib.check(lib.mp_heads_bwd(x.data_ptr(), stats.data_ptr(), fold.data_ptr(), out.data_ptr(), gamma.data_ptr(), beta.data_ptr(), W.data_ptr(),
                                              b.data_ptr(), dout.data_ptr(), dx.data_ptr(), *[t.data_ptr() for t in grads], K, O, M, C, impl,
                                              scratch.data_ptr(), scratch.numel(), st))
    fwd(); bwd()
    torch.cuda.synchronize()
    res = [out.clone(), dx.clone()] + [t.clone() for t in grads]
    times = []
    for f in (fwd, bwd):
        for _ in range(2): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / 10 * 1e3)
    return res, times


r1, t1 = run(1)
r2, t2 = run(2)
for name, a, c in zip(("out", "dx", "dgamma", "dbeta", "dW", "db"), r1, r2):
    print(f"{name:7s} max |row - mfma| {float((a - c).abs().max()):.3e}   scale {float(a.abs().max()):.3e}")
print(f"forward : row kernels {t1[0]:8.1f} us   matrix cores {t2[0]:8.1f} us")
print(f"backward: row kernels {t1[1]:8.1f} us   matrix cores {t2[1]:8.1f} us   (dx + parameter gradients + reduction, one stream)")
print(f"bytes: x {M * C * 4 / 1e6:.0f} MB -> forward floor {M * C * 4 / 8e12 * 1e6:.0f} us, backward floor {3 * M * C * 4 / 8e12 * 1e6:.0f} us at 8 TB/s")
