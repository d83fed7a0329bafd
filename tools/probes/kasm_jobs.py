"""Diagnostics build, MANIPOSE_GEMM_DEBUG=132: the plain persistent GEMM records the DMA jobs of every step (workgroups 0-7, wave 0) instead of
issuing them; this prints them next to what they should be.  MANIPOSE_HIP_LIB=<diag .so> python tools/probes/kasm_jobs.py [M]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
stamps = torch.zeros(256 * 64 * 2 + 256 * 8 * 4, dtype=torch.int64, device="cuda")
os.environ["MANIPOSE_GEMM_STAMPS"] = hex(stamps.data_ptr()); os.environ["MANIPOSE_GEMM_DEBUG"] = "132"
from manipose_amd import _lib
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
M = int(sys.argv[1]) if len(sys.argv) > 1 else 66100
N, K = 512, 512
xh = torch.randn(M, K, device="cuda").bfloat16(); Wh = torch.randn(N, K, device="cuda").bfloat16(); b = torch.randn(N, device="cuda")
y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
_lib.check(lib.mp_linear_fwd_bf16(xh.data_ptr(), Wh.data_ptr(), b.data_ptr(), y.data_ptr(), None, None, M, N, K, 0, st))
torch.cuda.synchronize()
r = stamps[:8 * 4 * 32 * 6].view(8, 4, 32, 6).cpu()
A, B = xh.data_ptr(), Wh.data_ptr()
print(f"A = {A:#x} .. {A + M * K * 2:#x}, B = {B:#x}")
for wg in range(8):
    for t in range(4):
        for ks in range(8):
            o = [int(v) for v in r[wg, t, ks]]
            if o[0] == 0 and o[1] == 0: continue
            k, last, fetch, m0, m0n = o[0] & 255, (o[0] >> 8) & 1, (o[0] >> 9) & 1, (o[0] >> 16) & 0xFFFFFF, (o[0] >> 40) & 0xFFFFFF
            a0, a3 = o[3] & 0xFFFFFFFF, (o[3] >> 32) & 0xFFFFFFFF
            lo, hi = o[1] + min(a0, a3), o[1] + max(a0, a3) + 16
            ok = (not fetch) or (A <= lo and hi <= A + M * K * 2)
            print(f"wg {wg} tile {t} ks {k} last {last} fetch {fetch} m0 {m0} m0n {m0n} has_next {o[5] >> 32}: A base - A = {o[1] - A:#x} B base - B = {o[2] - B:#x} aoff0 {a0:#x} aoff3 {a3:#x} lds {o[5] & 0xFFFFFFFF:#x}  {'ok' if ok else 'OUT OF RANGE'}")
