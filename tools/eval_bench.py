"""Evaluation throughput (forward only, flip-TTA batched into the same forward, aggregation + MPJPE + analytics kernels), BASELINE
config #5 shape by default: python tools/eval_bench.py [T] [windows_per_batch] [precision]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "hpe"))
import torch
from _entry import evaluate
from manipose_amd import RMCLManifoldMixSTE, h36m_skeleton

T = int(sys.argv[1]) if len(sys.argv) > 1 else 81
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
prec = sys.argv[3] if len(sys.argv) > 3 else "bf16"
torch.manual_seed(0)
model = RMCLManifoldMixSTE(h36m_skeleton(), num_frame=T, n_hyp=5, drop_path_rate=0.1)
model.precision = prec
model.max_batch_hint = 2 * B
model = model.cuda().eval()
g = torch.Generator(device="cuda").manual_seed(1)
n = 8 * B
X = (0.3 * torch.randn(n, T, 17, 2, device="cuda", generator=g)).clamp(-1, 1)
y = 0.1 * torch.randn(n, T, 17, 3, device="cuda", generator=g)
y[:, :, 0] = 0
evaluate(model, X[:B], y[:B], batch=B, tta=True, analytics=True)
torch.cuda.synchronize()
for tta, ana in ((True, True), (True, False), (False, False)):
    t0 = time.perf_counter()
    res = evaluate(model, X, y, batch=B, tta=tta, analytics=ana)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"T={T} B={B} {prec} tta={tta} analytics={ana}: {n * T / dt:,.0f} poses/s ({1e3 * dt / (n // B):.1f} ms per batch of {B} windows), MPJPE {res['mpjpe']:.1f} mm", flush=True)
