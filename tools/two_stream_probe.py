"""Probe: do two independent half-batch training streams (two models, two torch streams, complementary kernels interleaving) beat one
full-batch stream?  WGS=<n> (mp_set_option "gemm_persist_wgs") limits the persistent GEMMs' workgroups so that the other stream's kernels find free CUs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manipose_amd import RMCLManifoldMixSTE, h36m_skeleton
from manipose_amd.training import LiftingTrainer
from manipose_amd import _lib
if os.environ.get("WGS"):
    _lib.check(_lib.load().mp_set_option(b"gemm_persist_wgs", int(os.environ["WGS"])))

B = int(os.environ.get("B", "79")); NS = int(os.environ.get("NS", "2")); prec = os.environ.get("PREC", "bf16x3")
def make(b):
    torch.manual_seed(42)
    m = RMCLManifoldMixSTE(h36m_skeleton(), drop_path_rate=0.1); m.precision = prec; m.max_batch_hint = b
    m = m.cuda().train()
    return LiftingTrainer(m, seed=42)
per = [B // NS + (1 if i < B % NS else 0) for i in range(NS)]
trs = [make(b) for b in per]
streams = [torch.cuda.Stream() for _ in range(NS)]
data = []
for b in per:
    X = (0.3 * torch.randn(b, 243, 17, 2, device="cuda")).clamp(-1, 1); y = 0.3 * torch.randn(b, 243, 17, 3, device="cuda"); y[:, :, 0] = 0
    data.append((X, y))
def step():
    for tr, st, (X, y) in zip(trs, streams, data):
        with torch.cuda.stream(st):
            tr.train_step(X, y)
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 8
for _ in range(n): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"streams={NS} per-stream batch={per} wgs={os.environ.get('WGS','all')} prec={prec}: {dt*1e3:.1f} ms/step, {B*243/dt:.0f} poses/s", flush=True)
