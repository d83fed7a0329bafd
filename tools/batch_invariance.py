"""Is the eval-mode forward of a window independent of the batch it sits in and of the GEMM kernel family that runs?
The bench's parity block measured 2.1e-5 m at B=79 (persistent split-precision GEMMs) against 0.9e-5 m for the same windows as a
stand-alone batch of 3 (tiled kernels).  This probe runs the bench model's forward on 3 windows (a) alone, (b) embedded in a 79-window
batch with the default kernel choice, (c) the same with the persistent kernels switched off, and prints how the outputs differ.
python tools/batch_invariance.py [precision] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manipose_amd import RMCLManifoldMixSTE, _lib, h36m_skeleton

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 79
lib = _lib.load()
torch.manual_seed(42)
m = RMCLManifoldMixSTE(h36m_skeleton(), num_frame=243, n_hyp=5, drop_path_rate=0.1)
with torch.no_grad():
    for n, p in m.named_parameters():
        if n.endswith("pos_embed"):
            p.normal_(0.0, 0.02)
m.precision = prec
m.max_batch_hint = B
m = m.cuda().eval()
g = torch.Generator().manual_seed(4242)
Xp = (0.3 * torch.randn(3, 243, 17, 2, generator=g)).clamp(-1, 1).cuda()
gg = torch.Generator(device="cuda").manual_seed(42)
X = (0.3 * torch.randn(B, 243, 17, 2, device="cuda", generator=gg)).clamp(-1, 1)
idx = [0, (B - 1) // 2, B - 1]
X[idx] = Xp


def run(x, mode, side=1):
    _lib.check(lib.mp_set_option(b"gemm_persist_mode", mode))
    if (m.side_stream, m.wgrad_stream) != (bool(side), bool(side)):      # the engine's extra streams are per-model state (mp_model_config::streams): rebuild
        m.side_stream = m.wgrad_stream = bool(side)
        m._engine = None
    with torch.no_grad():
        p, s = m(x)
        h = m._engine.peek(0).clone()
        ln = m._engine.peek(1).clone()
    torch.cuda.synchronize()
    _lib.check(lib.mp_set_option(b"gemm_persist_mode", 1))
    return p.clone(), s.clone(), h, ln


def rep(tag, a, b):
    d = (a - b).norm(dim=-1)
    print(f"{tag}: mean {d.mean().item():.3e} m, p999 {torch.quantile(d.flatten().float(), 0.999).item():.3e}, max {d.max().item():.3e}, "
          f"identical elements {(a == b).float().mean().item():.4f}", flush=True)


p_small, _, h_small, l_small = run(Xp, 1)
p_small2, _, h_small2, l_small2 = run(Xp, 1)
rep("poses  stand-alone B=3, first forward of the process vs second", p_small, p_small2)
print("   head outputs identical: %.4f, lengths identical: %.4f" % ((h_small == h_small2).float().mean().item(), (l_small == l_small2).float().mean().item()))
ref = run(X, 0, 0)            # tiled kernels, one stream
rep("poses  B=%d tiled / one stream vs stand-alone B=3 (first)" % B, ref[0][idx], p_small)
rep("poses  B=%d tiled / one stream vs stand-alone B=3 (second)" % B, ref[0][idx], p_small2)
p_small3, _, h_small3, l_small3 = run(Xp, 1)
rep("poses  stand-alone B=3 after the big batch vs second", p_small3, p_small2)
hb = ref[2].view(5, B, 243 * 17, -1)[:, idx]
print("   head outputs of the big tiled run vs stand-alone (second): max %.3e; lengths max %.3e" % ((hb - h_small2.view(5, 3, 243 * 17, -1)).abs().max().item(), (ref[3].view(B, 16)[idx] - l_small2.view(3, 16)).abs().max().item()))
for mode, side in ((0, 0), (0, 1), (1, 0), (1, 1), (1, 1), (1, 1)):
    o = run(X, mode, side)
    d = (o[0] - ref[0]).norm(dim=-1)
    bad_w = (d.flatten(1).max(1).values > 1e-6).sum().item()
    print(f"persist={mode} side_streams={side}: poses vs the tiled one-stream run: mean {d.mean().item():.3e} m max {d.max().item():.3e}, windows touched {bad_w}/{B}; "
          f"rot head outputs differing {(o[2] != ref[2]).float().mean().item():.5f} (max {(o[2] - ref[2]).abs().max().item():.3e}); "
          f"bone lengths differing {(o[3] != ref[3]).float().mean().item():.5f} (max {(o[3] - ref[3]).abs().max().item():.3e})", flush=True)

# layer-by-layer: which block of which net first differs between two runs of the same forward?
print("residual stream, two runs of the default configuration (persistent GEMMs, side stream on):")


def stream_dump():
    with torch.no_grad():
        m(X)
    out = {}
    for base, name, nb in ((300, "seg", 4), (100, "rot", 16)):
        for code in [base - 1] + list(range(base, base + 2 * nb)):
            out[(name, code - base)] = m._engine.peek(code).clone()
    torch.cuda.synchronize()
    return out


a, b_ = stream_dump(), stream_dump()
for key in a:
    ne = (a[key] != b_[key])
    if ne.any():
        rows = torch.nonzero(ne.view(-1, a[key].numel() // (B * 243 * (16 if key[0] == "seg" else 17))).any(1)).flatten()
        print(f"  {key[0]} stream point {key[1]} ({'embedding' if key[1] < 0 else ('x_mid' if key[1] % 2 == 0 else 'x_out') + ' of block ' + str(key[1] // 2)}): "
              f"{int(ne.sum())} elements differ, max {float((a[key] - b_[key]).abs().max()):.3e}, {rows.numel()} rows, first rows {rows[:8].tolist()}, last {rows[-3:].tolist()}", flush=True)
print("  (points not listed are bit-identical)")
key = ("seg", 4)
ne = (a[key] != b_[key]).view(-1, 128)
rows = torch.nonzero(ne.any(1)).flatten()
for r in rows[:12].tolist():
    cols = torch.nonzero(ne[r]).flatten().tolist()
    va, vb = a[key].view(-1, 128)[r, cols[:4]].tolist(), b_[key].view(-1, 128)[r, cols[:4]].tolist()
    print(f"  row {r} (tile row {r % 128}, wave-row {r % 64}): cols {cols}; run A {[round(v, 4) for v in va]} run B {[round(v, 4) for v in vb]}")
