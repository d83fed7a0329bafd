import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import torch
import manipose_ref as orc
from manipose_amd import RMCLManifoldMixSTE, h36m_skeleton
H, B = int(os.environ.get("H", "79")), 0
B = 2 * H
torch.manual_seed(42)
model = RMCLManifoldMixSTE(h36m_skeleton(), num_frame=243, n_hyp=5, drop_path_rate=0.1)
model.precision = "bf16x3"; model.max_batch_hint = B
model = model.cuda().eval()
X, _ = orc.synthetic_batch(H, 243, seed=11); X = X.cuda()
X2 = torch.cat([X, X], 0).contiguous()
model._ensure_engine(B, X.device)
eng, flat = model._engine, model.flat_parameters()
with torch.no_grad():
    poses, scores = eng.forward(flat, X2, train=False)
torch.cuda.synchronize()
def cmp(name, t, rows_per_window):
    t = t.reshape(B * rows_per_window, -1) if t.numel() % (B * rows_per_window) == 0 else None
    if t is None: print(name, "shape?"); return
    a, b = t[:H * rows_per_window], t[H * rows_per_window:]
    d = (a - b).abs()
    print(f"{name:28s} differ {int((d > 0).sum()):10d} of {d.numel():10d}  max {float(d.max()):.3e}  rows differing {int((d.max(1).values > 0).sum())}", flush=True)
cmp("embedding (99)", eng.peek(99), 243 * 17)
for l in range(16):
    cmp(f"block {l} after attention", eng.peek(100 + 2 * l), 243 * 17)
    cmp(f"block {l} after mlp", eng.peek(101 + 2 * l), 243 * 17)
ho = eng.peek(0).reshape(5, B * 243 * 17, -1)
d = (ho[:, :H * 4131] - ho[:, H * 4131:]).abs(); print("head outputs differ", int((d > 0).sum()), float(d.max()))
cmp("segment lengths", eng.peek(1), 1)
cmp("segments net embedding (299)", eng.peek(299), 243 * 16)
for l in range(4):
    cmp(f"seg block {l} after attention", eng.peek(300 + 2 * l), 243 * 16)
    cmp(f"seg block {l} after mlp", eng.peek(301 + 2 * l), 243 * 16)
