"""One persistent-GEMM instantiation per process against torch (fp32 reference of the same bf16-rounded operands): used to bring up a new
build of the hand-scheduled k-step (MANIPOSE_HIP_LIB=<variant.so>) kernel by kernel, so that a faulting kernel is named by the run that dies.
    python tools/probes/kasm_smoke.py fwd|dgrad|x3|x3res|x3gelu [M]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from manipose_amd import _lib
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
mode = sys.argv[1]; M = int(sys.argv[2]) if len(sys.argv) > 2 else 66100
torch.manual_seed(0)
N, K = 512, 512
x = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
xh = x.bfloat16(); Wh = W.bfloat16()
if mode == "fwd":
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    _lib.check(lib.mp_linear_fwd_bf16(xh.data_ptr(), Wh.data_ptr(), b.data_ptr(), y.data_ptr(), None, None, M, N, K, 0, st))
    ref = xh.float() @ Wh.float().t() + b
    err = (y.float() - ref).abs().max().item() / ref.abs().max().item()
elif mode == "dgrad":
    dy = torch.randn(M, N, device="cuda").bfloat16(); dx = torch.empty(M, K, device="cuda", dtype=torch.bfloat16)
    dW, db = torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda")
    slab = torch.empty(int(lib.mp_linear_bwd_slab_floats(N, K)), device="cuda")
    _lib.check(lib.mp_linear_bwd_bf16(dy.data_ptr(), 0, xh.data_ptr(), Wh.data_ptr(), dx.data_ptr(), 0, dW.data_ptr(), db.data_ptr(), M, N, K, slab.data_ptr(), slab.numel(), st))
    ref = dy.float() @ Wh.float()
    err = (dx.float() - ref).abs().max().item() / ref.abs().max().item()
else:
    xl, Wl = torch.empty_like(xh), torch.empty_like(Wh)
    lib.mp_split_bf16(x.data_ptr(), xh.data_ptr(), xl.data_ptr(), x.numel(), st); lib.mp_split_bf16(W.data_ptr(), Wh.data_ptr(), Wl.data_ptr(), W.numel(), st)
    epi = {"x3": 0, "x3gelu": 1, "x3res": 2}[mode]
    r = torch.randn(M, N, device="cuda") if epi == 2 else None
    yh, yl = torch.empty(M, N, device="cuda", dtype=torch.bfloat16), torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    y32 = torch.empty(M, N, device="cuda") if epi == 2 else None
    z = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if epi == 1 else None
    p = lambda t: t.data_ptr() if t is not None else None
    _lib.check(lib.mp_linear_fwd_bf16x3(p(xh), p(xl), p(Wh), p(Wl), p(b), p(y32) if epi == 2 else p(yh), p(yl), p(z), p(r), M, N, K, epi, st))
    pre = x.double() @ W.double().t() + b.double()
    ref = pre if epi == 0 else (torch.nn.functional.gelu(pre) if epi == 1 else pre + r.double())
    got = y32.double() if epi == 2 else yh.double() + yl.double()
    err = (got - ref).abs().max().item() / ref.abs().max().item()
torch.cuda.synchronize()
print(f"{mode} M={M}: max error / max |ref| = {err:.2e}", "OK" if err < (1e-2 if mode in ("fwd", "dgrad") else 2e-5) else "WRONG", flush=True)
