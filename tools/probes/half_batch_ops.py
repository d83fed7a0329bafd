"""Which operator makes a token's result depend on its row position?  (round 5: a 158-window batch made of two copies of a 79-window batch gives the
second copy the first copy's poses only up to rounding - tools/probes/half_batch_diag.py finds the first difference behind block 0's MLP branch, in
1 / 16 of the rows.)  Each operator of that branch runs on [X; X] (2 x 326 349 rows) through the C ABI; the two halves of its output are compared."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from manipose_amd import _lib
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
H = 326349; M = 2 * H
def halves(name, t):
    t = t.reshape(M, -1); d = (t[:H].float() - t[H:].float()).abs()
    print(f"{name:44s} values differing {int((d > 0).sum()):10d}  rows {int((d.max(1).values > 0).sum()):7d}  max {float(d.max()):.3e}", flush=True)
torch.manual_seed(0)
for (N, K, name, epi) in [(1024, 512, "fc1 (GELU epilogue)", 1), (512, 1024, "fc2 (residual epilogue)", 2), (512, 512, "proj (residual epilogue)", 2), (1536, 512, "qkv (bias epilogue)", 0)]:
    x1 = torch.randn(H, K, device="cuda"); x = torch.cat([x1, x1], 0).contiguous(); del x1
    W = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
    xh, xl = torch.empty_like(x, dtype=torch.bfloat16), torch.empty_like(x, dtype=torch.bfloat16)
    Wh, Wl = torch.empty_like(W, dtype=torch.bfloat16), torch.empty_like(W, dtype=torch.bfloat16)
    lib.mp_split_bf16(x.data_ptr(), xh.data_ptr(), xl.data_ptr(), x.numel(), st); lib.mp_split_bf16(W.data_ptr(), Wh.data_ptr(), Wl.data_ptr(), W.numel(), st)
    del x
    r = None
    if epi == 2:
        r1 = torch.randn(H, N, device="cuda"); r = torch.cat([r1, r1], 0).contiguous(); del r1
    yh, yl = torch.empty(M, N, device="cuda", dtype=torch.bfloat16), torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    y32 = torch.empty(M, N, device="cuda") if epi == 2 else None
    z = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if epi == 1 else None
    p = lambda t: t.data_ptr() if t is not None else None
    _lib.check(lib.mp_linear_fwd_bf16x3(p(xh), p(xl), p(Wh), p(Wl), p(b), p(y32) if epi == 2 else p(yh), p(yl) if epi != 2 else None, p(z), p(r), M, N, K, epi, st))
    torch.cuda.synchronize()
    if epi == 2: halves(name + " y", y32)
    else:
        halves(name + " hi plane", yh); halves(name + " lo plane", yl)
        if z is not None: halves(name + " gelu'", z)
    del xh, xl, yh, yl, y32, z, r
x1 = torch.randn(H, 512, device="cuda"); x = torch.cat([x1, x1], 0).contiguous(); del x1
g, b = torch.randn(512, device="cuda"), torch.randn(512, device="cuda")
y, stats = torch.empty_like(x), torch.empty(M, 2, device="cuda")
_lib.check(lib.mp_layernorm_fwd(x.data_ptr(), g.data_ptr(), b.data_ptr(), 1e-6, y.data_ptr(), stats.data_ptr(), M, 512, st)); torch.cuda.synchronize()
halves("LayerNorm forward fp32 y", y); halves("LayerNorm forward statistics", stats)
# where in the 256-row tiles do the fc1 rows sit whose lo plane differs?
K, N = 512, 1024
x1 = torch.randn(H, K, device="cuda"); x = torch.cat([x1, x1], 0).contiguous(); del x1
W = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
xh, xl = torch.empty_like(x, dtype=torch.bfloat16), torch.empty_like(x, dtype=torch.bfloat16)
Wh, Wl = torch.empty_like(W, dtype=torch.bfloat16), torch.empty_like(W, dtype=torch.bfloat16)
lib.mp_split_bf16(x.data_ptr(), xh.data_ptr(), xl.data_ptr(), x.numel(), st); lib.mp_split_bf16(W.data_ptr(), Wh.data_ptr(), Wl.data_ptr(), W.numel(), st)
yh, yl, z = (torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(3))
_lib.check(lib.mp_linear_fwd_bf16x3(xh.data_ptr(), xl.data_ptr(), Wh.data_ptr(), Wl.data_ptr(), b.data_ptr(), yh.data_ptr(), yl.data_ptr(), z.data_ptr(), None, M, N, K, 1, st))
torch.cuda.synchronize()
d = (yl[:H].float() - yl[H:].float()).abs()
rows = torch.nonzero(d.max(1).values > 0).flatten().cpu()
cols = torch.nonzero(d.max(0).values > 0).flatten().cpu()
print("rows differing:", len(rows), " first-copy row % 256 histogram (non-zero bins):", {int(k): int(v) for k, v in zip(*torch.unique(rows % 256, return_counts=True))})
print("second-copy row % 256:", sorted(set(int(v) for v in ((rows + H) % 256))))
print("columns differing:", len(cols), "col % 64 values:", sorted(set(int(c) % 64 for c in cols))[:70])
# are the first-copy values or the second-copy values the odd ones?  fp64 reference of the GELU of the product for a few differing rows
if len(rows) == 0: sys.exit(0)
r0 = rows[:64].cuda()
pre = (xh[r0].double() + xl[r0].double()) @ (Wh.double() + Wl.double()).t() + b.double()
want = torch.nn.functional.gelu(pre)
e1 = ((yh[r0].double() + yl[r0].double()) - want).abs().max().item(); e2 = ((yh[r0 + H].double() + yl[r0 + H].double()) - want).abs().max().item()
print(f"error of the first copy {e1:.3e}, of the second copy {e2:.3e} (64 differing rows, against fp64)")
