"""Bitwise run-to-run reproducibility of a full training step (forward + backward) at full width; names the parameters whose
gradients differ between identical steps, in backward order (heads first), to locate a racy kernel.
python tools/step_determinism.py [batch] [runs] [precision] [wgrad_stream 0|1]   (STEP_SIDE=0: the segments net on the caller's stream)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manipose_amd import RMCLManifoldMixSTE, _lib, h36m_skeleton

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
prec = sys.argv[3] if len(sys.argv) > 3 else "bf16x3"
lib = _lib.load()
torch.manual_seed(42)
model = RMCLManifoldMixSTE(h36m_skeleton(), num_frame=243, n_hyp=5, drop_path_rate=0.1)
model.precision = prec
model.max_batch_hint = B
model.wgrad_stream = not (len(sys.argv) > 4 and sys.argv[4] == "0")
model.side_stream = os.environ.get("STEP_SIDE") != "0"
model = model.cuda().train()
g = torch.Generator(device="cuda").manual_seed(1)
X = (0.3 * torch.randn(B, 243, 17, 2, device="cuda", generator=g)).clamp(-1, 1)
model._ensure_engine(B, X.device)
eng, flat = model._engine, model.flat_parameters()
dp = torch.randn(B, 5, 243, 17, 3, device="cuda", generator=g) * 1e-3
ds = torch.randn(B, 5, 243, 1, device="cuda", generator=g) * 1e-3
layout = list(eng.layout)


def step():
    poses, scores = eng.forward(flat, X, train=True, seed=5, step=9)
    grads = torch.zeros_like(flat)
    eng.backward(flat, grads, dp, ds)
    torch.cuda.synchronize()
    return poses.clone(), grads


p0, g0 = step()
for r in range(1, runs):
    p, gr = step()
    bad = [(n, int((gr[o:o + k].view(torch.int32) != g0[o:o + k].view(torch.int32)).sum()), float((gr[o:o + k] - g0[o:o + k]).abs().max()), float(g0[o:o + k].abs().max()))
           for n, o, k in layout if not torch.equal(gr[o:o + k], g0[o:o + k])]
    print(f"run {r}: poses identical {torch.equal(p, p0)}; {len(bad)} of {len(layout)} gradient tensors differ", flush=True)
    if bad:
        rot = [b for b in bad if b[0].startswith("rotations")]
        seg = [b for b in bad if b[0].startswith("segments")]
        for tag, lst in (("rot", rot), ("seg", seg)):
            names = [b[0] for b in lst]
            print(f"   {tag}: {len(lst)} tensors; e.g. " + "; ".join(f"{n.split('module.')[1]} ({c} values, max diff {d:.2e} of {m:.2e})" for n, c, d, m in lst[:3]))
            if tag == "rot" and r == 1:
                d = {n: (c, dd, mm) for n, c, dd, mm in lst}
                L = max(int(n.split("blocks.")[1].split(".")[0]) for n in names if "blocks." in n)
                for blk in (f"TTEblocks.{L}", f"STEblocks.{L}", f"TTEblocks.{L - 1}"):
                    print(f"      {blk} (backward order): " + ", ".join(
                        f"{q}={'same' if ('rotations_module.' + blk + '.' + q) not in d else '%d/%.1e' % d['rotations_module.' + blk + '.' + q][:2]}"
                        for q in ("mlp.fc2.weight", "mlp.fc2.bias", "mlp.fc1.weight", "mlp.fc1.bias", "norm2.weight", "norm2.bias", "attn.proj.weight", "attn.proj.bias",
                                  "attn.qkv.weight", "attn.qkv.bias", "norm1.weight", "norm1.bias")))
                for q in ("Spatial_norm.weight", "Temporal_norm.weight", "Temporal_pos_embed", "head.0.norm.weight", "head.0.prediction_head.weight"):
                    k = "rotations_module." + q
                    print(f"      {q}: {'same' if k not in d else '%d/%.1e' % d[k][:2]}")
            blocks = sorted({n.split("blocks.")[0][-3:] + n.split("blocks.")[1].split(".")[0] for n in names if "blocks." in n})
            print(f"      blocks touched: {blocks}; heads touched: {any('head' in n for n in names)}; embedding touched: {any('embed' in n or 'proj.' in n and 'joints' in n for n in names)}")
