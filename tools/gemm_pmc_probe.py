"""Run a few launches of the forward qkv-shaped bf16 GEMM so a rocprofv3 --pmc pass can attribute stall reasons."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manipose_amd import _lib
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
M, N, K = 66096, 1536, 512
x = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
b = torch.randn(N, device="cuda"); y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(5):
    lib.mp_linear_fwd_bf16(x.data_ptr(), W.data_ptr(), b.data_ptr(), y.data_ptr(), None, None, M, N, K, 0, st)
torch.cuda.synchronize()
