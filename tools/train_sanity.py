"""Longer training run in all three precisions on a fixed synthetic batch (memorisation): the loss must fall steadily and stay finite.
python tools/train_sanity.py [steps] [batch] [lr] [seeds]
With seeds > 1 every precision is trained from `seeds` different initialisations / DropPath streams and the end state of the scoring
term is printed per run (winner histogram of the K heads, mean winning score): the round-2 review saw score_reg end at 0.005 in
bf16x3 but at 0.0001 in fp32 and bf16 with ONE seed - this tells a precision effect from the spread between basins."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manipose_amd import RMCLManifoldMixSTE, h36m_skeleton
from manipose_amd.training import LiftingTrainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
lr = float(sys.argv[3]) if len(sys.argv) > 3 else 2e-4
seeds = int(sys.argv[4]) if len(sys.argv) > 4 else 1
def run(precision, X, y, seed=0):
    torch.manual_seed(seed)
    model = RMCLManifoldMixSTE(h36m_skeleton(), num_frame=243, n_hyp=5, drop_path_rate=0.1)
    model.precision = precision.split("/")[0]
    model.f16f8 = 3 if precision.endswith("/f16f8") else 0      # "bf16x3/f16f8": the benchmark's operand form since round 6
    model = model.cuda().train()
    tr = LiftingTrainer(model, lr=lr, weight_decay=1e-6, seed=seed)
    hist = []
    for i in range(steps):
        terms = tr.train_step(X, y)
        if i % max(1, steps // 12) == 0 or i == steps - 1:
            v = terms.tolist()
            hist.append(sum(v))
            print(f"{precision} step {i:4d}: total {sum(v):.4f}  wloss {v[0]:.4f} score_reg {v[1]:.4f} vloss {v[2]:.4f} sreg {v[3]:.4f}", flush=True)
    # end state of the scoring path: which head wins (oracle choice) per frame, and how confident the score head is about it
    with torch.no_grad():
        poses, scores = model.eval()(X)
        err = (poses - y[:, None]).norm(dim=-1).mean(-1)            # (B, K, T)
        win = err.argmin(1)
        counts = torch.bincount(win.flatten(), minlength=poses.shape[1]).tolist()
        pw = scores[..., 0].gather(1, win[:, None]).mean().item()
        gap = (err.topk(2, dim=1, largest=False).values.diff(dim=1)).mean().item()
    print(f"{precision} seed {seed}: winners per head {counts}, mean score of the winner {pw:.4f}, mean gap best -> second {gap * 1e3:.3f} mm, "
          f"final score_reg {v[1]:.5f}", flush=True)
    return hist


# targets ON the manifold and learnable: hypothesis 0 of a teacher = the student's own initialisation with perturbed weights
torch.manual_seed(0)
teacher = RMCLManifoldMixSTE(h36m_skeleton(), num_frame=243, n_hyp=5, drop_path_rate=0.0)
gp = torch.Generator().manual_seed(5)
with torch.no_grad():
    for n, p in teacher.named_parameters():
        p.add_(0.05 * p.abs().mean().clamp_min(0.02) * torch.randn(p.shape, generator=gp))
teacher = teacher.cuda().eval()
g = torch.Generator(device="cuda").manual_seed(1)
t = torch.linspace(0, 1, 243, device="cuda")[None, :, None, None]
X = (0.5 * torch.sin(6.28 * (1 + 2 * torch.rand(B, 1, 17, 2, device="cuda", generator=g)) * t + 6.28 * torch.rand(B, 1, 17, 2, device="cuda", generator=g))).contiguous()
with torch.no_grad():
    y = teacher(X)[0][:, 0].contiguous()
del teacher
torch.cuda.empty_cache()
hists = {}
for sd in range(seeds):
    for p in ("fp32", "bf16x3/f16f8", "bf16x3", "bf16"):
        hists[(p, sd)] = run(p, X, y, seed=sd)
for p, hist in hists.items():
    assert all(torch.isfinite(torch.tensor(hist))), f"{p}: non-finite loss"
    assert hist[-1] < 0.6 * hist[0], (p, hist[0], hist[-1])
# the split precision against fp32, same seed (round-5 advisory: not only "the loss falls"): the first step's loss is the forward at equal weights
# (1e-3: DropPath streams are the same, the forward differs by ~1e-5 m); the end state of a winner-take-all trajectory is chaotic in the heads that win
# ("ten-step trajectories"), so the final total is held to a band, not to digits
for (p, sd), hist in hists.items():
    if p.startswith("bf16x3"):
        ref = hists[("fp32", sd)]
        assert abs(hist[0] - ref[0]) <= 1e-3 * abs(ref[0]), (p, sd, hist[0], ref[0])
        assert abs(hist[-1] - ref[-1]) <= 0.25 * abs(ref[-1]), (p, sd, hist[-1], ref[-1])
print("ok:", {p: (round(h[0], 4), round(h[-1], 4)) for p, h in hists.items()})
