"""Probe: the whole training step captured into one HIP graph (torch.cuda.CUDAGraph) and replayed, against eager launches.
Timing only - the DropPath step counter and Adam's step count are baked into the captured kernel arguments.
python tools/graph_probe.py [batch] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manipose_amd import RMCLManifoldMixSTE, h36m_skeleton
from manipose_amd.training import LiftingTrainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 79
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
torch.manual_seed(42)
model = RMCLManifoldMixSTE(h36m_skeleton(), num_frame=243, n_hyp=5, drop_path_rate=0.1)
model.precision = "bf16"
model.max_batch_hint = B
model = model.cuda().train()
tr = LiftingTrainer(model, lr=4e-5, weight_decay=1e-6, seed=42)
g0 = torch.Generator(device="cuda").manual_seed(1)
X = (0.3 * torch.randn(B, 243, 17, 2, device="cuda", generator=g0)).clamp(-1, 1)
y = 0.3 * torch.randn(B, 243, 17, 3, device="cuda", generator=g0)
y[:, :, 0] = 0
for _ in range(3):
    tr.train_step(X, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    tr.train_step(X, y)
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / steps
print(f"eager : {1e3 * eager:.2f} ms/step", flush=True)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    tr.train_step(X, y)
torch.cuda.synchronize()
for _ in range(2):
    graph.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    graph.replay()
torch.cuda.synchronize()
rep = (time.perf_counter() - t0) / steps
print(f"graph : {1e3 * rep:.2f} ms/step  ({100 * (eager / rep - 1):+.1f} %)", flush=True)
