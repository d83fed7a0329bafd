"""Achieved HBM GB/s of the window-gather kernel (mp_gather_windows): 17 x (2 + 3) x 4 = 340 B read + 340 B written per frame."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from manipose_amd import h36m_skeleton
from manipose_amd.augmentations import PoseFlip
from manipose_amd.data import PoseSequenceGenerator

g = np.random.default_rng(0)
lens = [20000] * 64
p3 = [g.standard_normal((n, 17, 3), dtype=np.float32) for n in lens]
p2 = [g.standard_normal((n, 17, 2), dtype=np.float32) for n in lens]
gen = PoseSequenceGenerator(p3, p2, None, seq_len=243, random_start=True, drop_last=True, transform=PoseFlip(h36m_skeleton(), 0.5))
B = 4096
t0 = time.perf_counter(); seq, start, flip = gen.draw(torch.randint(0, len(gen), (B,)).tolist()); host_ms = (time.perf_counter() - t0) * 1e3
for _ in range(3): gen.gather(seq, start, flip)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): gen.gather(seq, start, flip)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"mp_gather_windows B={B} T=243: {ms * 1e3:.1f} us  {B * 243 * 680 / ms / 1e6:.0f} GB/s (read + write), {B * 243 / ms / 1e3:.1f} M frames/s; "
      f"host-side draw of the {B} (sequence, start, flip) triples: {host_ms:.1f} ms")
