"""Is the split-precision forward GEMM limited by the chip's power management rather than by its instruction schedule?  Round 5 removed 20 % of
the main loop's shader cycles (hand-scheduled k-step, young-wave DMA, fragment pipeline across the steps) and the launch time moved by 9 %: the
shader clock computed from the diagnostics build's cycle counters fell with every step (1.76 -> 1.65 -> 1.54 GHz).  This probe runs ONE shape of
the kernel for a few seconds per operand content - the matrix pipes' power depends on how many operand bits toggle - and samples rocm-smi beside it:
    randn     the bench's operands            zeros     all-zero operands (no toggling: the floor of the data-dependent power)
    ones      constant operands               sparse    randn with 7 of 8 elements zeroed
Same instruction stream in every case; a time that follows the operand content is a time set by power, not by the schedule.
    python tools/probes/power_clock.py [qkv|proj|fc1|fc2][-bwd] [seconds per case]      (-bwd: the bf16 dgrad + weight-gradient kernels of the shape)"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from manipose_amd import _lib

lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
shape = sys.argv[1] if len(sys.argv) > 1 else "qkv"
bwd = shape.endswith("-bwd")          # the bf16 backward of the shape (dgrad + weight-gradient kernels) instead of the split-precision forward
shape = shape.replace("-bwd", "")
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
N, K, epi = {"qkv": (1536, 512, 0), "proj": (512, 512, 2), "fc1": (1024, 512, 1), "fc2": (512, 1024, 2)}[shape]
M = 326349
samples = []


def poll(stop):
    while not stop.is_set():
        try:
            out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=5).stdout
            samples.append(out)
        except Exception as e:      # the probe's timing result does not depend on it
            samples.append(f"rocm-smi failed: {e}")
        stop.wait(0.5)


def summarize(txt):
    import re
    sclk = re.findall(r"sclk[^,\n]*\((\d+)Mhz\)", txt)
    rows = [l for l in txt.splitlines() if l.startswith("card")]
    return (sclk, rows[:1])


for kind in ("randn", "zeros", "ones", "sparse", "randn"):
    torch.manual_seed(0)
    x = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / K ** 0.5
    if kind == "zeros": x.zero_(); W.zero_()
    if kind == "ones": x.fill_(1.0); W.fill_(1.0 / K)
    if kind == "sparse": x *= (torch.rand_like(x) < 0.125); W *= (torch.rand_like(W) < 0.125)
    xh, xl = torch.empty_like(x, dtype=torch.bfloat16), torch.empty_like(x, dtype=torch.bfloat16)
    Wh, Wl = torch.empty_like(W, dtype=torch.bfloat16), torch.empty_like(W, dtype=torch.bfloat16)
    lib.mp_split_bf16(x.data_ptr(), xh.data_ptr(), xl.data_ptr(), x.numel(), st); lib.mp_split_bf16(W.data_ptr(), Wh.data_ptr(), Wl.data_ptr(), W.numel(), st)
    del x
    b = torch.zeros(N, device="cuda") if kind in ("zeros",) else torch.randn(N, device="cuda")
    r = (torch.zeros(M, N, device="cuda") if kind == "zeros" else torch.randn(M, N, device="cuda")) if epi == 2 else None
    yh, yl = torch.empty(M, N, device="cuda", dtype=torch.bfloat16), torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    y32 = torch.empty(M, N, device="cuda") if epi == 2 else None
    z = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if epi == 1 else None
    p = lambda t: t.data_ptr() if t is not None else None
    if bwd:
        dy = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16) if kind == "zeros" else (torch.ones(M, N, device="cuda") if kind == "ones" else torch.randn(M, N, device="cuda")).bfloat16()
        dx = torch.empty(M, K, device="cuda", dtype=torch.bfloat16); dW, db = torch.zeros(N, K, device="cuda"), torch.zeros(N, device="cuda")
        slab = torch.empty(int(lib.mp_linear_bwd_slab_floats(N, K)), device="cuda")
    run = (lambda: lib.mp_linear_bwd_bf16(dy.data_ptr(), 0, xh.data_ptr(), Wh.data_ptr(), dx.data_ptr(), 0, dW.data_ptr(), db.data_ptr(), M, N, K, slab.data_ptr(), slab.numel(), st)) if bwd else lambda: lib.mp_linear_fwd_bf16x3(p(xh), p(xl), p(Wh), p(Wl), p(b), p(y32) if epi == 2 else p(yh), p(yl), p(z), p(r), M, N, K, epi, st)
    for _ in range(5): run()
    torch.cuda.synchronize()
    samples.clear()
    stop = threading.Event(); th = threading.Thread(target=poll, args=(stop,)); th.start()
    t0 = time.time(); n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < secs:
        for _ in range(50): run()
        n += 50
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    stop.set(); th.join()
    us = e0.elapsed_time(e1) * 1e3 / n
    print(f"{shape} {kind:7s}: {us:8.1f} us per launch  ({(4.0 if bwd else 6.0) * M * N * K / us / 1e6:7.1f} TF/s issued)   rocm-smi samples: {[summarize(s) for s in samples[1:4]]}", flush=True)
