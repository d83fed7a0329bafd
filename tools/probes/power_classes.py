"""Which kernel classes of the training step run at the board's power cap?  Each kernel below loops for a few seconds at the bench's token count
(random operands) with rocm-smi sampled beside it: socket power, shader clock, time per launch.  A class below the cap runs at the part's full
2.4 GHz and answers to its schedule; a class at the cap answers to its energy ("power").  Companion of power_clock.py (GEMMs).
    python tools/probes/power_classes.py [seconds per kernel]"""
import os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from manipose_amd import _lib

lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
B, T, J, C, H = 79, 243, 17, 512, 8
M = B * T * J


def measure(name, fn, bytes_per_launch=None):
    samples, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            try:
                out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=5).stdout
                row = [l for l in out.splitlines() if l.startswith("card0")]
                if row:
                    f = row[0].split(",")
                    samples.append((int(re.sub(r"\D", "", f[5])), float(f[9])))
            except Exception:
                return
            stop.wait(0.3)

    for _ in range(3): fn()
    torch.cuda.synchronize()
    th = threading.Thread(target=poll, daemon=True); th.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0, n = time.time(), 0
    e0.record()
    try:
        while time.time() - t0 < secs:
            for _ in range(20): fn()
            n += 20
            torch.cuda.synchronize()
        e1.record(); torch.cuda.synchronize()
    finally:
        stop.set(); th.join(timeout=10)
    us = e0.elapsed_time(e1) * 1e3 / n
    s = samples[2:] if len(samples) > 3 else samples
    pw = sum(w for _, w in s) / len(s) if s else float("nan")
    ck = sum(c for c, _ in s) / len(s) if s else float("nan")
    bw = f"  {bytes_per_launch / us / 1e6:5.2f} TB/s" if bytes_per_launch else ""
    print(f"{name:46s} {us:8.1f} us per launch{bw}   {pw:6.0f} W   {ck:5.0f} MHz   ({len(s)} samples)", flush=True)


qkv = torch.randn(M, 3 * C, device="cuda").bfloat16(); dout = torch.randn(M, C, device="cuda").bfloat16()
ql = (torch.randn(M, 3 * C, device="cuda") * 2 ** -8).bfloat16()
out = torch.empty(M, C, device="cuda", dtype=torch.bfloat16); ol = torch.empty_like(out); dq = torch.empty(M, 3 * C, device="cuda", dtype=torch.bfloat16)
lse = torch.empty(B * J * H * T, device="cuda"); delta = torch.empty_like(lse)
for temporal in (1, 0):
    nm = "temporal" if temporal else "spatial"
    measure(f"attention {nm} forward, split precision", lambda: lib.mp_attention_fwd_bf16x3(qkv.data_ptr(), ql.data_ptr(), out.data_ptr(), ol.data_ptr(), lse.data_ptr(), None, temporal, B, T, J, C, H, st),
            M * C * 2 * 8)
    lib.mp_attention_fwd_bf16(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), temporal, B, T, J, C, H, st)
    measure(f"attention {nm} backward, bf16", lambda: lib.mp_attention_bwd_bf16(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), delta.data_ptr(), dq.data_ptr(), temporal, B, T, J, C, H, st),
            M * C * 2 * 8)
del qkv, ql, dq
x = torch.randn(M, C, device="cuda"); g, b = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
y, stats = torch.empty_like(x), torch.empty(M, 2, device="cuda")
dy, dx, dskip = torch.randn_like(x), torch.empty_like(x), torch.randn_like(x)
dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
scratch = torch.empty(1024 * 2 * C + 16, device="cuda")
measure("LayerNorm forward fp32 -> fp32", lambda: lib.mp_layernorm_fwd(x.data_ptr(), g.data_ptr(), b.data_ptr(), 1e-6, y.data_ptr(), stats.data_ptr(), M, C, st), 2 * x.numel() * 4)
measure("LayerNorm backward (+ skip)", lambda: lib.mp_layernorm_bwd(dy.data_ptr(), x.data_ptr(), stats.data_ptr(), g.data_ptr(), dskip.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), M, C,
                                                                       scratch.data_ptr(), scratch.numel(), st), 4 * x.numel() * 4)
measure("device copy (torch), random / zero data", lambda: y.copy_(x), 2 * x.numel() * 4)
x.zero_()
measure("device copy (torch), zeros", lambda: y.copy_(x), 2 * x.numel() * 4)
