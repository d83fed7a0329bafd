"""Shape sweep of the whole model against the CPU oracle (checker only): unusual window lengths / widths / batch sizes, all three precisions,
forward + fused loss + backward.  python tools/shape_fuzz.py [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import manipose_ref as orc
from manipose_amd import ManifoldMixSTE, RMCLManifoldMixSTE, h36m_skeleton
from manipose_amd.metrics import manifold_training_loss, mpjpe_error, rmcl_training_loss

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cases = [(2, 32, 4, 1, 3), (3, 64, 8, 5, 5), (17, 32, 4, 2, 1), (64, 64, 4, 1, 2), (100, 128, 8, 3, 4), (243, 64, 8, 2, 5), (256, 32, 4, 1, 3),
         (300, 64, 4, 2, 2), (81, 128, 8, 7, 0), (27, 256, 8, 2, 0),
         # head dim 64 with 128 < T <= 256: the two-phase split-precision temporal attention forward; ragged last GEMM tiles
         (200, 256, 4, 2, 3), (255, 128, 2, 3, 2), (130, 64, 1, 1, 2), (243, 512, 8, 1, 2), (241, 128, 2, 5, 1)]
worst = 0.0
for T, C, H, B, K in cases:
    cfg = dict(T=T, J=17, num_bones=16, C_rot=C, depth_rot=2, heads_rot=H, C_seg=32, depth_seg=1, heads_seg=4, n_hyp=K)
    st = orc.make_state(cfg, seed=int(rng.integers(1 << 30)))
    X, y = orc.synthetic_batch(B, T, seed=int(rng.integers(1 << 30)))
    kw = dict(skeleton=h36m_skeleton(), num_frame=T, embed_dim_rot=C, depth_rot=2, num_heads_rot=H, embed_dim_seg=32, depth_seg=1,
              num_heads_seg=4, drop_path_rate=0.0)
    req = {k: v.clone().requires_grad_(True) for k, v in st.items()}
    if K > 0:
        op, os_ = orc.rmcl_manifold_forward(X, req, orc.oracle_cfg(cfg))
        ot, _ = orc.rmcl_training_loss(op, os_, y)
    else:
        op = orc.manifold_forward(X, req, orc.oracle_cfg(cfg))
        ot, _ = orc.manifold_training_loss(op, y)
    ot.backward()
    for prec in ("fp32", "bf16x3", "bf16"):
        model = RMCLManifoldMixSTE(n_hyp=K, **kw) if K > 0 else ManifoldMixSTE(**kw)
        model.load_state_dict(st, strict=True)
        model.precision = prec
        model = model.cuda().eval()
        out = model(X.cuda())
        if K > 0:
            tot, _ = rmcl_training_loss(out[0], out[1], y.cuda())
            poses = out[0]
        else:
            tot, _ = manifold_training_loss(out, y.cuda())
            poses = out
        tot.backward()
        mp = mpjpe_error(poses, op.detach().cuda(), "average").item()
        errs = {k: ((p.grad.cpu() - req[k].grad).abs().max() / (req[k].grad.abs().max() + 1e-12)).item() for k, p in model.named_parameters()}
        wk = max(errs, key=errs.get)
        gerr = errs[wk]
        cos = min(torch.nn.functional.cosine_similarity(p.grad.cpu().reshape(-1), req[k].grad.reshape(-1), dim=0).item()
                  for k, p in model.named_parameters() if req[k].grad.abs().max() > 0)       # (K = 1: the score head has no gradient)
        if prec == "fp32":
            ok = mp <= 1e-4 and gerr <= 5e-3 and abs(tot.item() - ot.item()) <= 1e-4 * abs(ot.item())
        elif prec == "bf16x3":        # forward inside the north-star bound on every shape; bf16 backward
            ok = mp <= 1e-4 and np.isfinite(gerr) and cos > 0.97 and abs(tot.item() - ot.item()) <= 1e-3 * abs(ot.item())
        else:
            ok = mp <= 5e-2 and np.isfinite(gerr) and abs(tot.item() - ot.item()) <= 5e-2 * abs(ot.item())
        worst = max(worst, mp if prec == "fp32" else 0.0)
        print(f"T={T:3d} C={C:3d} H={H} B={B} K={K} {prec}: MPJPE {mp:.2e} m, loss {tot.item():.5f} vs {ot.item():.5f}, worst grad rel {gerr:.2e} ({wk}), min cosine {cos:.4f}  {'ok' if ok else 'FAIL'}",
              flush=True)
        assert ok
print("all shapes ok; worst fp32 MPJPE", worst)
