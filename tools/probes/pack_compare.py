"""Round 6: the activation-form f16f8 split (mp_split_f16f8, weight = 0) of the library in MANIPOSE_HIP_LIB (default: in-tree) on a fixed input that
covers the clamped range; writes / compares a reference file: python tools/probes/pack_compare.py write|check <file>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from manipose_amd import _lib
lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(1 << 22, device="cuda", generator=g) * torch.exp(4 * torch.randn(1 << 22, device="cuda", generator=g))
x[::1000] = 447.9; x[1::1000] = 448.0; x[2::1000] = 448.1; x[3::1000] = -70000.0; x[4::1000] = 1e-7; x[5::1000] = 0.0; x[6::1000] = float("inf")
h = torch.empty(x.numel(), device="cuda", dtype=torch.int16); c = torch.empty(2 * x.numel(), device="cuda", dtype=torch.uint8)
_lib.check(lib.mp_split_f16f8(x.data_ptr(), h.data_ptr(), c.data_ptr(), x.numel(), 0, torch.cuda.current_stream().cuda_stream), "split")
torch.cuda.synchronize()
if sys.argv[1] == "write":
    torch.save({"h": h.cpu(), "c": c.cpu()}, sys.argv[2]); print("written", sys.argv[2])
else:
    ref = torch.load(sys.argv[2])
    dh, dc = int((ref["h"] != h.cpu()).sum()), int((ref["c"] != c.cpu()).sum())
    print(f"fp16 plane: {dh} of {h.numel()} differ; correction plane: {dc} of {c.numel()} bytes differ; share of |x| > 448: {float((x.abs() > 448).float().mean()):.4f}")
    sys.exit(1 if dh or dc else 0)
