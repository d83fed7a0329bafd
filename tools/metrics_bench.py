"""Achieved HBM GB/s of the one-pass evaluation-analytics kernel (mp_pose_metrics): algorithmic bytes = 2 x 17 x 3 x 4 = 408 B per
frame (prediction + target read once), against the oracle restatement's wall time for the same table on the host."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import torch
from manipose_amd.metrics import pose_analytics

B, L = int(os.environ.get("B", "4096")), 243
pred = (400 * torch.randn(B, L, 17, 3, device="cuda")); gt = pred + 50 * torch.randn_like(pred)
for _ in range(3): pose_analytics(pred, gt)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): pose_analytics(pred, gt)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"mp_pose_metrics  B={B} L={L}: {ms * 1e3:.1f} us  {B * L * 408 / ms / 1e6:.0f} GB/s algorithmic ({B * L / ms * 1e3 / 1e6:.1f} M frames/s)")
if os.environ.get("CPU", "1") == "1":
    import manipose_ref as orc
    n = 256
    p, g = pred[:n].cpu(), gt[:n].cpu()
    t0 = time.perf_counter()
    jc, gj = p.permute(0, 3, 2, 1), g.permute(0, 3, 2, 1)
    orc.mpjpe_error(p, g); orc.mse_error(p, g); orc.jointwise_error(p, g); orc.sagittal_symmetry(jc, "average", False)
    orc.segments_time_consistency(jc.permute(1, 2, 0, 3).reshape(1, 3, 17, -1), "std"); orc.segments_len_err(jc, gj, "average", False)
    orc.keypoint_3d_pck_auc(p.reshape(-1, 17, 3), g.reshape(-1, 17, 3)); orc.eval_velocity_error(p, g)
    dt = time.perf_counter() - t0
    print(f"CPU oracle, same table, {n} windows: {dt * 1e3:.1f} ms ({n * L / dt / 1e6:.2f} M frames/s, {torch.get_num_threads()} threads)")
