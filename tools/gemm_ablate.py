import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manipose_amd import _lib
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
M, N, K = 66096, 1536, 512
x = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
b = torch.randn(N, device="cuda"); y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
def t(n=20):
    for _ in range(3): lib.mp_linear_fwd_bf16(x.data_ptr(), W.data_ptr(), b.data_ptr(), y.data_ptr(), None, None, M, N, K, 0, st)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): lib.mp_linear_fwd_bf16(x.data_ptr(), W.data_ptr(), b.data_ptr(), y.data_ptr(), None, None, M, N, K, 0, st)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
print(f"MANIPOSE_GEMM_DEBUG={os.environ.get('MANIPOSE_GEMM_DEBUG','0')}: {t():.1f} us per qkv-shaped launch (ideal MFMA 42 us)")
