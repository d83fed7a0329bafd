"""Times the bf16 temporal / spatial attention kernels through the C ABI at the bench shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manipose_amd import _lib
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
if os.environ.get("ATTN2P") is not None:         # bit 0: two-phase split-precision forward, bit 1: two-image backward
    _lib.check(lib.mp_set_option(b"attn_two_phase", int(os.environ["ATTN2P"])))
B, T, J, C, H = int(os.environ.get("B", "32")), int(os.environ.get("T", "243")), int(os.environ.get("J", "17")), 512, 8
M = B * T * J
qkv = torch.randn(M, 3 * C, device="cuda").bfloat16(); dout = torch.randn(M, C, device="cuda").bfloat16()
out = torch.empty(M, C, device="cuda", dtype=torch.bfloat16); dq = torch.empty(M, 3 * C, device="cuda", dtype=torch.bfloat16)
lse = torch.empty(B * J * H * T, device="cuda"); delta = torch.empty_like(lse)
def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for temporal in (1, 0):
    f = t(lambda: lib.mp_attention_fwd_bf16(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), temporal, B, T, J, C, H, st))
    b = t(lambda: lib.mp_attention_bwd_bf16(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), delta.data_ptr(), dq.data_ptr(), temporal, B, T, J, C, H, st))
    byt_f, byt_b = M * C * 2 * 4, M * C * 2 * (3 + 2 + 3)
    print(f"{'temporal' if temporal else 'spatial '} B={B} dbg={os.environ.get('MANIPOSE_ATTN_DEBUG','0')}: fwd {f:7.1f} us ({byt_f / f / 1e6:5.2f} TB/s)  bwd {b:7.1f} us ({byt_b / b / 1e6:5.2f} TB/s)", flush=True)
# split precision (bf16x3): planar hi/lo qkv and output
ql = (torch.randn(M, 3 * C, device="cuda") * 2 ** -8).bfloat16(); ol = torch.empty_like(out)
for temporal in (1, 0):
    f = t(lambda: lib.mp_attention_fwd_bf16x3(qkv.data_ptr(), ql.data_ptr(), out.data_ptr(), ol.data_ptr(), lse.data_ptr(), None, temporal, B, T, J, C, H, st))
    print(f"{'temporal' if temporal else 'spatial '} B={B} bf16x3 fwd {f:7.1f} us ({M * C * 2 * 8 / f / 1e6:5.2f} TB/s of planar bytes)", flush=True)
