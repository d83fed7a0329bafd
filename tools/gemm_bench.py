"""Micro-benchmark of the bf16 GEMM kernels through the C ABI (one MI355X): TFLOP/s per shape and variant."""
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
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 66096
    for (N, K, name) in [(1536, 512, "qkv"), (512, 512, "proj"), (1024, 512, "fc1"), (512, 1024, "fc2")]:
        x = torch.randn(M, K, device="cuda").bfloat16()
        W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
        b = torch.randn(N, device="cuda")
        r = torch.randn(M, N, device="cuda")
        dy = torch.randn(M, N, device="cuda").bfloat16()
        dy32 = dy.float()
        y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        y32 = torch.empty(M, N, device="cuda")
        z = torch.empty_like(y)
        dx = torch.empty(M, K, device="cuda")
        dW, db = torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda")
        slab = torch.empty(int(lib.mp_linear_bwd_slab_floats(N, K)), device="cuda")
        fl = 2.0 * M * N * K
        res = {}
        res["fwd_bias"] = timeit(lambda: lib.mp_linear_fwd_bf16(x.data_ptr(), W.data_ptr(), b.data_ptr(), y.data_ptr(), None, None, M, N, K, 0, st))
        res["fwd_gelu"] = timeit(lambda: lib.mp_linear_fwd_bf16(x.data_ptr(), W.data_ptr(), b.data_ptr(), y.data_ptr(), z.data_ptr(), None, M, N, K, 1, st))
        res["fwd_resid"] = timeit(lambda: lib.mp_linear_fwd_bf16(x.data_ptr(), W.data_ptr(), b.data_ptr(), y32.data_ptr(), None, r.data_ptr(), M, N, K, 2, st))
        res["bwd_bf16dy(dgrad+wgrad)"] = timeit(lambda: lib.mp_linear_bwd_bf16(dy.data_ptr(), 0, x.data_ptr(), W.data_ptr(), dx.data_ptr(), 1, dW.data_ptr(), db.data_ptr(), M, N, K, slab.data_ptr(), slab.numel(), st)) / 2
        res["bwd_f32dy(dgrad+wgrad)"] = timeit(lambda: lib.mp_linear_bwd_bf16(dy32.data_ptr(), 1, x.data_ptr(), W.data_ptr(), dx.data_ptr(), 1, dW.data_ptr(), db.data_ptr(), M, N, K, slab.data_ptr(), slab.numel(), st)) / 2
        print(f"{name:5s} M={M} N={N} K={K}: " + "  ".join(f"{k}={fl / (v * 1e-3) / 1e12:6.1f}TF" for k, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
