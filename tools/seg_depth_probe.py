import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from manipose_amd import RMCLManifoldMixSTE, h36m_skeleton
from manipose_amd.training import LiftingTrainer
B = 79
for dseg in (2, 1, 2, 1):
    torch.manual_seed(42)
    m = RMCLManifoldMixSTE(h36m_skeleton(), drop_path_rate=0.1, depth_seg=dseg); m.precision = "bf16x3"; m.max_batch_hint = B
    tr = LiftingTrainer(m.cuda().train(), seed=42)
    X = (0.3 * torch.randn(B, 243, 17, 2, device="cuda")).clamp(-1, 1); y = 0.3 * torch.randn(B, 243, 17, 3, device="cuda"); y[:, :, 0] = 0
    for _ in range(3): tr.train_step(X, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8): tr.train_step(X, y)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
    print(f"depth_seg={dseg}: {dt*1e3:.2f} ms/step", flush=True)
    m._engine = None; del tr, m; torch.cuda.empty_cache()
