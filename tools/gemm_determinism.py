"""Run-to-run determinism and persistent-vs-tiled agreement of the Linear GEMM kernels at the bench shapes (M = 79 * 4131 tokens).
python tools/gemm_determinism.py [windows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manipose_amd import _lib

lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 79
NTOK = int(sys.argv[2]) if len(sys.argv) > 2 else 17        # 17: rotations-net shapes, 16: bones-net shapes (C = 128)
LOAD = int(sys.argv[3]) if len(sys.argv) > 3 else 0         # 1: a second stream keeps the chip busy with persistent GEMMs meanwhile
ONLY = sys.argv[4] if len(sys.argv) > 4 else ""   # run only the shapes whose name contains this
M = B * 243 * NTOK
st = lambda: torch.cuda.current_stream().cuda_stream


def split(t):
    hi = torch.empty_like(t, dtype=torch.bfloat16)
    lo = torch.empty_like(t, dtype=torch.bfloat16)
    _lib.check(lib.mp_split_bf16(t.data_ptr(), hi.data_ptr(), lo.data_ptr(), t.numel(), st()))
    return hi, lo


def run_x3(xh, xl, Wh, Wl, b, N, K, epi, r, stats, gam, bet, mode):
    _lib.check(lib.mp_set_option(b"gemm_persist_mode", mode))
    outs = []
    if epi in (0, 1):
        yh = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        yl = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
        z = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16) if epi == 1 else None
        _lib.check(lib.mp_linear_fwd_bf16x3(xh.data_ptr(), xl.data_ptr(), Wh.data_ptr(), Wl.data_ptr(), b.data_ptr(), yh.data_ptr(), yl.data_ptr(),
                                            z.data_ptr() if z is not None else None, None, M, N, K, epi, st()))
        outs = [yh.view(torch.int16), yl.view(torch.int16)] + ([z.view(torch.int16)] if z is not None else [])
    elif epi == 2:
        y = torch.zeros(M, N, device="cuda")
        _lib.check(lib.mp_linear_fwd_bf16x3(xh.data_ptr(), xl.data_ptr(), Wh.data_ptr(), Wl.data_ptr(), b.data_ptr(), y.data_ptr(), None, None, r.data_ptr(),
                                            M, N, K, 2, st()))
        outs = [y.view(torch.int32)]
    else:
        y = torch.zeros(M, N, device="cuda")
        _lib.check(lib.mp_linear_fwd_bf16x3_lnres(xh.data_ptr(), xl.data_ptr(), Wh.data_ptr(), Wl.data_ptr(), b.data_ptr(), y.data_ptr(), r.data_ptr(),
                                                  stats.data_ptr(), gam.data_ptr(), bet.data_ptr(), None, 0, 243, 17, M, N, K, st()))
        outs = [y.view(torch.int32)]
    torch.cuda.synchronize()
    _lib.check(lib.mp_set_option(b"gemm_persist_mode", 1))
    return outs


g = torch.Generator(device="cuda").manual_seed(1)
SHAPES = ((("qkv  bias", 1536, 512, 0), ("fc1  gelu", 1024, 512, 1), ("proj resid", 512, 512, 2), ("fc2  resid", 512, 1024, 2), ("proj lnres", 512, 512, 3)) if NTOK == 17 else
          (("qkv  bias", 384, 128, 0), ("fc1  gelu", 256, 128, 1), ("proj resid", 128, 128, 2), ("fc2  resid", 128, 256, 2), ("proj lnres", 128, 128, 3)))
side = torch.cuda.Stream()
if LOAD:
    Ml = 79 * 243 * 17
    lx = torch.randn(Ml, 512, device="cuda").bfloat16()
    lW = torch.randn(1536, 512, device="cuda").bfloat16()
    lb = torch.randn(1536, device="cuda")
    ly = torch.empty(Ml, 1536, device="cuda", dtype=torch.bfloat16)


def load_burst(n=6):
    if LOAD:
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(n):
                _lib.check(lib.mp_linear_fwd_bf16(lx.data_ptr(), lW.data_ptr(), lb.data_ptr(), ly.data_ptr(), None, None, Ml, 1536, 512, 0, side.cuda_stream))


for name, N, K, epi in SHAPES:
    if ONLY not in name:
        continue
    x = torch.randn(M, K, device="cuda", generator=g)
    W = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=g)
    r = torch.randn(M, N, device="cuda", generator=g)
    stats = torch.stack([r.mean(1), (r.var(1, unbiased=False) + 1e-6).rsqrt()], 1).contiguous()
    gam, bet = torch.rand(N, device="cuda", generator=g) + 0.5, torch.randn(N, device="cuda", generator=g)
    xh, xl = split(x)
    Wh, Wl = split(W)
    ref = run_x3(xh, xl, Wh, Wl, b, N, K, epi, r, stats, gam, bet, 0)
    ref2 = run_x3(xh, xl, Wh, Wl, b, N, K, epi, r, stats, gam, bet, 0)
    runs = []
    for _ in range(8):
        load_burst()
        runs.append(run_x3(xh, xl, Wh, Wl, b, N, K, epi, r, stats, gam, bet, 1))
    runs_t = []
    for _ in range(8):
        load_burst()
        runs_t.append(run_x3(xh, xl, Wh, Wl, b, N, K, epi, r, stats, gam, bet, 0))
    nt = [sum(int((a != c).sum().item()) for a, c in zip(o, ref)) for o in runs_t]
    print(f"{name} N={N} K={K}: tiled-only under load, elements differing from the first tiled run: {nt}", flush=True)
    same_t = all(torch.equal(a, c) for a, c in zip(ref, ref2))
    msg = []
    for i, o in enumerate(runs):
        nd = sum(int((a != c).sum().item()) for a, c in zip(o, ref))
        rows = torch.unique(torch.nonzero((o[0] != ref[0]).any(1)).flatten() // 256) if nd else torch.tensor([])
        msg.append(f"run{i}: {nd} elements differ from tiled" + (f" in {rows.numel()} row panels (first {rows[:6].tolist()})" if nd else ""))
    print(f"{name} N={N} K={K}: tiled twice identical={same_t}; " + "; ".join(msg), flush=True)
