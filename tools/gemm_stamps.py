"""Where the persistent GEMM's workgroups are in time: every workgroup stamps the start and end of each tile's epilogue (MANIPOSE_GEMM_STAMPS,
10 ns ticks of the constant clock).  Prints, for one launch of the split-precision qkv shape, the epilogue durations and how the epilogues of
the 256 workgroups line up (all at once = a chip-wide burst of stores, or spread over the tile time).  Needs the diagnostics build of the
library (MP_DIAG=1 bash manipose_amd/csrc/build.sh).   [MANIPOSE_GEMM_STAGGER=ticks] python tools/gemm_stamps.py [x3|x3gelu|x3res|bf16|dgrad [N K]]
(x3gelu / x3res: the split-precision forward with the GELU epilogue of fc1 / the fp32 residual epilogue of proj and fc2)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
diag = os.environ.get("MANIPOSE_DIAG_LIB") or os.path.join(ROOT, "manipose_amd", "libmanipose_hip_diag.so")      # MANIPOSE_DIAG_LIB: another diagnostics build (A/B of schedule variants)
assert os.path.exists(diag), "build the diagnostics library first: MP_DIAG=1 bash manipose_amd/csrc/build.sh"
os.environ["MANIPOSE_HIP_LIB"] = diag
stamps = torch.zeros(256 * 64 * 2 + 256 * 8 * 4, dtype=torch.int64, device="cuda")
os.environ["MANIPOSE_GEMM_STAMPS"] = hex(stamps.data_ptr())
from manipose_amd import _lib
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
mode = sys.argv[1] if len(sys.argv) > 1 else "x3"
M, N, K = 326349, 1536, 512
if len(sys.argv) > 3: N, K = int(sys.argv[2]), int(sys.argv[3])      # dgrad: N = reduction (dy columns), K = dx columns
x = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / K ** 0.5
xh, xl = torch.empty_like(x, dtype=torch.bfloat16), torch.empty_like(x, dtype=torch.bfloat16)
Wh, Wl = torch.empty_like(W, dtype=torch.bfloat16), torch.empty_like(W, dtype=torch.bfloat16)
lib.mp_split_bf16(x.data_ptr(), xh.data_ptr(), xl.data_ptr(), x.numel(), st); lib.mp_split_bf16(W.data_ptr(), Wh.data_ptr(), Wl.data_ptr(), W.numel(), st)
b = torch.randn(N, device="cuda")
yh, yl = torch.empty(M, N, device="cuda", dtype=torch.bfloat16), torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
if mode == "x3gelu": zz = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
if mode == "x3res": y32, rr = torch.empty(M, N, device="cuda"), torch.randn(M, N, device="cuda")
if mode == "dgrad":
    # (the call also runs the weight-gradient GEMM, which is not a persistent kernel and writes no stamps: the "launch us" line includes it)
    dy = torch.randn(M, N, device="cuda").bfloat16(); dx = torch.empty(M, K, device="cuda", dtype=torch.bfloat16)
    dW, db = torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda")
    slab = torch.empty(int(lib.mp_linear_bwd_slab_floats(N, K)), device="cuda")
def run():
    if mode == "dgrad":
        _lib.check(lib.mp_linear_bwd_bf16(dy.data_ptr(), 0, xh.data_ptr(), Wh.data_ptr(), dx.data_ptr(), 0, dW.data_ptr(), db.data_ptr(), M, N, K, slab.data_ptr(), slab.numel(), st))
    elif mode == "x3":
        _lib.check(lib.mp_linear_fwd_bf16x3(xh.data_ptr(), xl.data_ptr(), Wh.data_ptr(), Wl.data_ptr(), b.data_ptr(), yh.data_ptr(), yl.data_ptr(), None, None, M, N, K, 0, st))
    elif mode == "x3gelu":
        _lib.check(lib.mp_linear_fwd_bf16x3(xh.data_ptr(), xl.data_ptr(), Wh.data_ptr(), Wl.data_ptr(), b.data_ptr(), yh.data_ptr(), yl.data_ptr(), zz.data_ptr(), None, M, N, K, 1, st))
    elif mode == "x3res":
        _lib.check(lib.mp_linear_fwd_bf16x3(xh.data_ptr(), xl.data_ptr(), Wh.data_ptr(), Wl.data_ptr(), b.data_ptr(), y32.data_ptr(), None, None, rr.data_ptr(), M, N, K, 2, st))
    else:
        _lib.check(lib.mp_linear_fwd_bf16(xh.data_ptr(), Wh.data_ptr(), b.data_ptr(), yh.data_ptr(), None, None, M, N, K, 0, st))
for _ in range(3): run()
torch.cuda.synchronize(); stamps.zero_(); run(); torch.cuda.synchronize()
s = stamps[:256 * 64 * 2].view(256, 64, 2).cpu()
cyc = stamps[256 * 64 * 2:].view(256, 8, 4).cpu().double()
t0 = int(s[s > 0].min())
tiles = int((s[:, :, 1] > 0).all(0).sum())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
# the constant clock's rate from the launch itself: first stamp = end of the first tile's main loop, last stamp = end of the launch
ev_us = e0.elapsed_time(e1) * 1e3
tick_us = ev_us * (tiles - 0.9) / tiles / (int(s.max()) - t0)
print(f"{mode}: {tiles} tiles per workgroup, launch {ev_us:.0f} us (HIP events), clock tick ~{tick_us * 1e3:.1f} ns, stagger {os.environ.get('MANIPOSE_GEMM_STAGGER', '0')} ticks per phase")
start, end = (s[:, :tiles, 0] - t0).double() * tick_us, (s[:, :tiles, 1] - t0).double() * tick_us      # us
dur = end - start
print(f"epilogue duration us: mean {dur.mean():.2f}  min {dur.min():.2f}  max {dur.max():.2f};  tile period us: {(start[:, 1:] - start[:, :-1]).mean():.2f}")
gap = start[:, 1:] - end[:, :-1]
print(f"main loop (previous epilogue end -> this epilogue start) us: mean {gap.mean():.2f}")
for t in (0, 1, 2, tiles // 2, tiles - 2):
    st_t = start[:, t]
    print(f"tile {t:2d}: epilogue starts over the 256 workgroups: min {st_t.min():8.2f} us  max {st_t.max():8.2f}  std {st_t.std():6.2f};  durations mean {dur[:, t].mean():.2f}")
# how many workgroups are inside an epilogue at once, sampled every 0.5 us over the launch
T = float(end.max())
ts = torch.arange(0, T, 0.5, dtype=torch.float64)
busy = ((start.reshape(-1, 1) <= ts) & (end.reshape(-1, 1) > ts)).sum(0)
print(f"workgroups inside an epilogue at a time: mean {busy.double().mean():.1f}, max {int(busy.max())}; fraction of the launch with > 128 of them in it: {(busy > 128).double().mean():.2f}, with none: {(busy == 0).double().mean():.2f}")

# shader-clock totals per wave over the launch: where the main loop's cycles go
tot = cyc.sum(-1, keepdim=True)
fr = (cyc / tot).mean((0, 1))
print(f"per-wave cycle split over the launch (mean of 256 x 8 waves): parked on s_waitcnt (operand DMA) {fr[0]:.3f}, on the barrier {fr[1]:.3f}, MFMA stage {fr[2]:.3f}, "
      f"epilogue {fr[3]:.3f}; cycles per wave {tot.mean():.0f}")
ab = cyc.mean(0)      # absolute shader-clock cycles per wave index (what an A/B of two builds compares: the fractions hide a category that grows)
print("  absolute k-cycles by wave index (wait / barrier / mma / epilogue): " + "  ".join(f"w{i}: {ab[i, 0] / 1e3:.0f}/{ab[i, 1] / 1e3:.0f}/{ab[i, 2] / 1e3:.0f}/{ab[i, 3] / 1e3:.0f}" for i in range(8)))
w = cyc / tot
print("  by wave index (wait / barrier / mma / epilogue): " + "  ".join(f"w{i}: {w[:, i, 0].mean():.2f}/{w[:, i, 1].mean():.2f}/{w[:, i, 2].mean():.2f}/{w[:, i, 3].mean():.2f}" for i in range(8)))
