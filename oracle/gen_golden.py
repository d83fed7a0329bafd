"""Generate golden vectors by running the *reference itself* (read-only at /root/reference).

Run only in the build container (the reference never travels to the GPU box):

    python oracle/gen_golden.py

Outputs small ``.npz`` fixtures (data only: inputs, weights, expected outputs) under
``tests/golden/``. Third-party packages missing offline are replaced by import stand-ins
that are NOT reference code:
  * ``timm.models.layers.DropPath`` (timm==0.9.16, requirements_frozen.txt:6): stand-in with the
    published semantics (per-row-of-dim-0 Bernoulli(keep)/keep in train mode, identity in eval);
    it also records the masks it drew so the oracle can be fed identical masks.
  * ``mup.MuReadout`` (mup==1.0.0): subclass of nn.Linear, never instantiated with mup=False.
``rotation_tools.normalize_vector`` hard-codes ``.cuda()`` (rotation_tools.py:10-13); it is
replaced at run time by the same arithmetic on the input's device.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import manipose_ref as orc  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
RECORDED_MASKS = []


class DropPath(nn.Module):
    def __init__(self, drop_prob=0.0, scale_by_keep=True):
        super().__init__()
        self.drop_prob, self.scale_by_keep = drop_prob, scale_by_keep

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1 - self.drop_prob
        m = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        if keep > 0 and self.scale_by_keep:
            m.div_(keep)
        RECORDED_MASKS.append(m.reshape(-1).clone())
        return x * m


def import_reference():
    tl = types.ModuleType("timm.models.layers")
    tl.DropPath = DropPath
    sys.modules.update({"timm": types.ModuleType("timm"), "timm.models": types.ModuleType("timm.models"),
                        "timm.models.layers": tl})
    mup = types.ModuleType("mup")
    mup.MuReadout = type("MuReadout", (nn.Linear,), {})
    sys.modules["mup"] = mup
    sys.path.insert(0, "/root/reference/hpe")
    import mh_so3_hpe.architectures.utils.rotation_tools as rt

    def _nv(v):
        mag = torch.max(torch.sqrt(v.pow(2).sum(1)), torch.tensor([1e-8], dtype=v.dtype, device=v.device))
        return v / mag.view(-1, 1).expand(-1, v.shape[1])
    rt.normalize_vector = _nv
    from mh_so3_hpe.architectures import RMCLManifoldMixSTE, ManifoldMixSTE, MixSTE
    import mh_so3_hpe.metrics as M
    from mh_so3_hpe.data.skeleton import Skeleton
    from mh_so3_hpe.data.h36m_lifting import T_POSE_OPERATORS
    from mh_so3_hpe.architectures.pose_decoder import PoseDecoder
    sk = Skeleton(parents=list(orc.H36M_PARENTS), joints_left=list(orc.H36M_JOINTS_LEFT),
                  joints_right=list(orc.H36M_JOINTS_RIGHT), t_pose_operators=T_POSE_OPERATORS)
    return dict(RMCL=RMCLManifoldMixSTE, Manifold=ManifoldMixSTE, MixSTE=MixSTE, M=M, sk=sk,
                PoseDecoder=PoseDecoder)


def build_ref_model(ref, cfg, drop_path_rate):
    kw = dict(skeleton=ref["sk"], num_frame=cfg["T"], num_joints=cfg["J"], num_bones=cfg["num_bones"],
              in_chans=2, rot_rep_dim=cfg.get("rot_dim", 6), embed_dim_rot=cfg["C_rot"], depth_rot=cfg["depth_rot"],
              num_heads_rot=cfg["heads_rot"], embed_dim_seg=cfg["C_seg"], depth_seg=cfg["depth_seg"],
              num_heads_seg=cfg["heads_seg"], drop_path_rate=drop_path_rate)
    if cfg["n_hyp"] > 0:
        return ref["RMCL"](n_hyp=cfg["n_hyp"], **kw)
    return ref["Manifold"](**kw)


def ref_losses(ref, poses, scores, y, squared=False):
    """The four default loss terms exactly as main_h36m_lifting.py:101-209 assembles them (squared = train.sq_loss)."""
    M = ref["M"]
    w = M.STANDARD_H36M_WEIGHTS
    wl = M.wta_l2_loss_and_activate_head(hypothesis=poses, y=y, weights=w, squared=squared)[0].mean()
    sr = M.wta_with_scoring_loss(hypothesis=poses, scores=scores, y=y, beta=0.1, weights=w, squared=squared)[1]
    vl = 2.0 * M.mean_velocity_error(predicted=poses, target=y, squared=squared, axis=2)
    sg = 0.5 * M.smoothness_regularization(prediction=poses, weights=w, axis=2)
    return wl, sr, vl, sg


def np_state(st):
    return {"w::" + k: v.detach().numpy() for k, v in st.items()}


def gen_model_fixture(ref, name, cfg, B, seed, drop_path_rate=0.0, train=False):
    torch.manual_seed(seed)
    st = orc.make_state(cfg, seed=seed)
    model = build_ref_model(ref, cfg, drop_path_rate)
    missing = model.load_state_dict(st, strict=True)   # also proves the key/shape layout
    assert not missing.missing_keys and not missing.unexpected_keys
    model.train(train)
    X, y = orc.synthetic_batch(B, cfg["T"], cfg["J"], seed=seed + 1)
    RECORDED_MASKS.clear()
    out = {"cfg": np.array([cfg[k] for k in ("T", "J", "num_bones", "C_rot", "depth_rot", "heads_rot",
                                              "C_seg", "depth_seg", "heads_seg", "n_hyp")], dtype=np.int64),
           "X": X.numpy(), "y": y.numpy(), "drop_path_rate": np.float64(drop_path_rate), "rot_dim": np.int64(cfg.get("rot_dim", 6))}
    if cfg["n_hyp"] > 0:
        rot, sc_only = model.rotations_module(X)
        RECORDED_MASKS.clear()
        bl = model.segments_module(X)
        RECORDED_MASKS.clear()
        torch.manual_seed(seed + 7)
        poses, scores = model(X)
        masks = [m.numpy() for m in RECORDED_MASKS]
        wl, sr, vl, sg = ref_losses(ref, poses, scores, y)
        total = wl + sr + vl + sg
        model.zero_grad()
        total.backward()
        if not train:
            out["rot6d"] = rot.detach().numpy()
            out["bones"] = bl.detach().numpy()
        out.update(poses=poses.detach().numpy(), scores=scores.detach().numpy(),
                   loss_terms=np.array([wl.item(), sr.item(), vl.item(), sg.item()], dtype=np.float64),
                   loss_total=np.float64(total.item()))
        e, idx = ref["M"].wta_l2_loss_and_activate_head(hypothesis=poses.detach(), y=y,
                                                        weights=ref["M"].STANDARD_H36M_WEIGHTS)
        out["wta_idx"] = idx.numpy()
        out["wta_val"] = e.numpy()
        # eval-time aggregation + the parity metric (rmcl_manifold_mix_ste.py:141-185, mean_joint_errors.py:31-36)
        with torch.no_grad():
            agg = model.aggregate(poses, scores, mode="weighted_ave")
            best = model.aggregate(poses, scores, mode="best_score")
            oe, op = model.aggregate(poses, mode="oracle", ground_truth=y)
            out.update(agg_weighted=agg.numpy(), agg_best=best.numpy(), agg_oracle=op.numpy(),
                       agg_oracle_err=oe.numpy(),
                       mpjpe_weighted=np.float64(ref["M"].mpjpe_error(agg, y, mode="average").item()))
    else:
        pred = model(X)
        masks = [m.numpy() for m in RECORDED_MASKS]
        M = ref["M"]
        w = M.STANDARD_H36M_WEIGHTS
        wl = M.weighted_mpjpe_loss(pred, y, weights=w)
        vl = 2.0 * M.mean_velocity_error(predicted=pred, target=y, squared=False, axis=1)
        sg = 0.5 * M.smoothness_regularization(prediction=pred, weights=w, axis=1)
        total = wl + vl + sg
        model.zero_grad()
        total.backward()
        out.update(poses=pred.detach().numpy(),
                   loss_terms=np.array([wl.item(), vl.item(), sg.item()], dtype=np.float64),
                   loss_total=np.float64(total.item()))
    for i, m in enumerate(masks):
        out[f"mask::{i:03d}"] = m
    out["n_masks"] = np.int64(len(masks))
    out.update(np_state(st))
    for k, p in model.named_parameters():
        out["g::" + k] = p.grad.detach().numpy()
    # one Adam step (main_h36m_lifting.py:234-238) on the same grads
    opt = torch.optim.Adam(model.parameters(), lr=4e-5, weight_decay=1e-6)
    opt.step()
    for k in ("rotations_module.STEblocks.0.attn.qkv.weight", "segments_module.head.1.weight",
              "rotations_module.Spatial_norm.bias"):
        out["adam1::" + k] = dict(model.named_parameters())[k].detach().numpy()
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1e3:.0f} kB, loss={total.item():.6f}, masks={len(masks)}")


def gen_mixste_fixture(ref):
    """model.arch=mixste (main_h36m_lifting.py:617-628): the bare MixSTE regressor with the single-hypothesis loss of make_loss."""
    torch.manual_seed(31)
    T, C, depth, heads = 9, 32, 2, 4
    model = ref["MixSTE"](num_frame=T, num_joints=17, in_chans=2, out_dim=3, embed_dim=C, depth=depth, num_heads=heads, drop_path_rate=0.0)
    g = torch.Generator().manual_seed(32)
    with torch.no_grad():
        for k, p in model.named_parameters():       # perturb the LN affines / zero-initialised position embeddings too
            p.add_(0.05 * torch.randn(p.shape, generator=g))
    model.eval()
    X, y = orc.synthetic_batch(2, T, 17, seed=33)
    out = {"cfg_mixste": np.array([T, C, depth, heads]), "X": X.numpy(), "y": y.numpy()}
    out.update({"w::" + k: v.detach().numpy().copy() for k, v in model.state_dict().items()})
    pred = model(X)
    M = ref["M"]
    w = M.STANDARD_H36M_WEIGHTS
    wl = M.weighted_mpjpe_loss(pred, y, weights=w)
    vl = 2.0 * M.mean_velocity_error(predicted=pred, target=y, squared=False, axis=1)
    sg = 0.5 * M.smoothness_regularization(prediction=pred, weights=w, axis=1)
    total = wl + vl + sg
    total.backward()
    out.update(poses=pred.detach().numpy(), loss_terms=np.array([wl.item(), vl.item(), sg.item()]), loss_total=np.float64(total.item()))
    for k, p in model.named_parameters():
        out["g::" + k] = p.grad.detach().numpy()
    # the same step with train.rigid_seg_reg = 0.7 (make_loss :170-177): value, d term / d prediction, parameter gradients of the total
    model.zero_grad()
    pred = model(X)
    pred.retain_grad()
    rigid = 0.7 * M.segments_time_consistency(pred.permute(0, 3, 2, 1), skeleton=ref["sk"], mode="sum")
    rigid.backward(retain_graph=True)
    out["rigid_term"], out["rigid_g_pred"] = np.float64(rigid.item()), pred.grad.detach().numpy().copy()
    model.zero_grad()
    pred = model(X)
    tot = (M.weighted_mpjpe_loss(pred, y, weights=w) + 2.0 * M.mean_velocity_error(predicted=pred, target=y, squared=False, axis=1)
           + 0.5 * M.smoothness_regularization(prediction=pred, weights=w, axis=1)
           + 0.7 * M.segments_time_consistency(pred.permute(0, 3, 2, 1), skeleton=ref["sk"], mode="sum"))
    tot.backward()
    out["rigid_total"] = np.float64(tot.item())
    for k, p in model.named_parameters():
        out["g_rigid::" + k] = p.grad.detach().numpy()
    np.savez_compressed(os.path.join(OUT, "mixste_tiny.npz"), **out)
    print("mixste_tiny: ok, loss", total.item(), "keys", len(model.state_dict()))


def gen_decoder_fixture(ref):
    torch.manual_seed(5)
    dec = ref["PoseDecoder"](skeleton=ref["sk"], rot_rep_dim=6)
    B, L = 3, 7
    rot = torch.randn(B * L, 17, 6)
    rot[0, 3] = 0.0                       # degenerate: both vectors zero -> 1e-8 clamp path
    rot[1, 5, 3:] = rot[1, 5, :3] * 2.0   # degenerate: colinear -> zero cross product
    rot[2, 0, :] = torch.tensor([1., 0, 0, 0, 1, 0])
    bl = torch.randn(B, 16, 1) * 0.3
    rot.requires_grad_(True)
    bl.requires_grad_(True)
    poses = dec(rotations_repr=rot, bones_lengths_repr=bl, root_positions=torch.zeros(B * L, 3))
    gp = torch.randn_like(poses)
    (poses * gp).sum().backward()
    # known answer from SURVEY section 4: identity rotations -> T-pose
    ident = torch.tensor([1., 0, 0, 0, 1, 0]).repeat(1, 17, 1)
    lens = torch.tensor([.2, .5, .5, .2, .5, .5, .2, .2, .2, .2, .2, .4, .4, .2, .4, .4]).view(1, 16, 1)
    tp = dec(rotations_repr=ident, bones_lengths_repr=lens, root_positions=torch.zeros(1, 3))
    np.savez_compressed(os.path.join(OUT, "decoder.npz"), rot6d=rot.detach().numpy(), bones=bl.detach().numpy(),
                        poses=poses.detach().numpy(), gpos=gp.numpy(), g_rot6d=rot.grad.numpy(),
                        g_bones=bl.grad.numpy(), tpose_lens=lens.numpy(), tpose=tp.detach().numpy())
    print("decoder: ok")
    gen_rot4_decoder_fixture(ref)


def gen_rot4_decoder_fixture(ref):
    """model.rot_dim=4: the 4-D representation of rotation_tools.py:60-116 through the reference's PoseDecoder."""
    torch.manual_seed(15)
    dec = ref["PoseDecoder"](skeleton=ref["sk"], rot_rep_dim=4)
    B, L = 3, 7
    rot = torch.randn(B * L, 17, 4)
    rot[2, 0, :] = torch.tensor([0., 1, 1, 0])      # c1 = 0, s1 = 1, c2 = 1, s2 = 0 -> the identity
    bl = torch.randn(B, 16, 1) * 0.3
    rot.requires_grad_(True)
    bl.requires_grad_(True)
    poses = dec(rotations_repr=rot, bones_lengths_repr=bl, root_positions=torch.zeros(B * L, 3))
    gp = torch.randn_like(poses)
    (poses * gp).sum().backward()
    ident = torch.tensor([0., 1, 1, 0]).repeat(1, 17, 1)
    lens = torch.tensor([.2, .5, .5, .2, .5, .5, .2, .2, .2, .2, .2, .4, .4, .2, .4, .4]).view(1, 16, 1)
    tp = dec(rotations_repr=ident, bones_lengths_repr=lens, root_positions=torch.zeros(1, 3))
    np.savez_compressed(os.path.join(OUT, "decoder_rot4.npz"), rot4d=rot.detach().numpy(), bones=bl.detach().numpy(),
                        poses=poses.detach().numpy(), gpos=gp.numpy(), g_rot4d=rot.grad.numpy(), g_bones=bl.grad.numpy(),
                        tpose_lens=lens.numpy(), tpose=tp.detach().numpy())
    print("decoder_rot4: ok")


def gen_loss_fixture(ref):
    torch.manual_seed(9)
    B, H, L, J = 2, 5, 9, 17
    poses = (0.3 * torch.randn(B, H, L, J, 3)).requires_grad_(True)
    logits = torch.randn(B, H, L, 1)
    scores = logits.softmax(dim=1).requires_grad_(True)
    y = 0.3 * torch.randn(B, L, J, 3)
    wl, sr, vl, sg = ref_losses(ref, poses, scores, y)
    total = wl + sr + vl + sg
    total.backward()
    np.savez_compressed(os.path.join(OUT, "loss.npz"), poses=poses.detach().numpy(), scores=scores.detach().numpy(),
                        y=y.numpy(), loss_terms=np.array([wl.item(), sr.item(), vl.item(), sg.item()]),
                        g_poses=poses.grad.numpy(), g_scores=scores.grad.numpy())
    print("loss: ok")
    gen_sq_loss_fixture(ref)


def gen_sq_loss_fixture(ref):
    """train.sq_loss=True: the squared variants of the WTA / velocity terms, multi-hypothesis and single."""
    M = ref["M"]
    torch.manual_seed(19)
    B, H, L, J = 2, 5, 9, 17
    poses = (0.3 * torch.randn(B, H, L, J, 3)).requires_grad_(True)
    scores = torch.randn(B, H, L, 1).softmax(dim=1).requires_grad_(True)
    y = 0.3 * torch.randn(B, L, J, 3)
    wl, sr, vl, sg = ref_losses(ref, poses, scores, y, squared=True)
    (wl + sr + vl + sg).backward()
    out = dict(poses=poses.detach().numpy(), scores=scores.detach().numpy(), y=y.numpy(),
               loss_terms=np.array([wl.item(), sr.item(), vl.item(), sg.item()]), g_poses=poses.grad.numpy(), g_scores=scores.grad.numpy())
    val, idx = M.wta_l2_loss_and_activate_head(hypothesis=poses.detach(), y=y, weights=M.STANDARD_H36M_WEIGHTS, squared=True)
    out["wta_vals"], out["wta_idx"] = val.numpy(), idx.numpy()
    for nm, w in (("w", M.STANDARD_H36M_WEIGHTS), ("nw", None)):      # single hypothesis (make_loss :113-127): weighted_mse_loss
        p1 = poses.detach()[:, 0].clone().requires_grad_(True)
        a = M.weighted_mse_loss(p1, y, weights=w)
        b = 2.0 * M.mean_velocity_error(predicted=p1, target=y, squared=True, axis=1)
        c = 0.5 * M.smoothness_regularization(prediction=p1, weights=w, axis=1)
        (a + b + c).backward()
        out[f"single_{nm}.terms"], out[f"single_{nm}.g"] = np.array([a.item(), b.item(), c.item()]), p1.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "loss_sq.npz"), **out)
    print("loss_sq: ok", out["loss_terms"])


def gen_metrics_fixture(ref):
    """Evaluation analytics: outputs of the reference's own metric functions on seeded pose sequences in
    millimetres: a 'free' prediction (noisy bones) and a 'rigid' one (constant bone lengths, the manifold case where a naive
    variance cancels)."""
    import mh_so3_hpe.metrics.regularizations as R
    import mh_so3_hpe.metrics.mean_joint_errors as E
    import mh_so3_hpe.metrics.pck as P
    from mh_so3_hpe.metrics.losses import mean_velocity_error
    sk = ref["sk"]
    g = torch.Generator().manual_seed(77)
    B, L, J = 3, 301, 17
    gt = 400.0 * torch.randn(B, L, J, 3, generator=g)
    gt[:, :, 0] = 0
    pred = gt + 60.0 * torch.randn(B, L, J, 3, generator=g)
    # rigid sequence: fixed bone lengths, random directions (walk down the tree)
    lens = 200.0 + 100.0 * torch.rand(B, 16, generator=g)
    rigid = torch.zeros(B, L, J, 3)
    for j in range(1, J):
        d = torch.randn(B, L, 3, generator=g)
        d = d / d.norm(dim=-1, keepdim=True)
        rigid[:, :, j] = rigid[:, :, orc.H36M_PARENTS[j]] + lens[:, j - 1, None, None] * d
    out = dict(pred=pred.numpy(), gt=gt.numpy(), rigid=rigid.numpy(), rigid_lens=lens.numpy())
    for nm, x in (("pred", pred), ("rigid", rigid)):
        jc = x.permute(0, 3, 2, 1)                            # (B, 3, J, L), the layout the reference passes
        for mode in ("average", "sum", "std", "min", "max"):
            out[f"{nm}.stc.{mode}"] = R.segments_time_consistency(jc, sk, mode).numpy()
        for mode in ("average", "sum", "std"):
            out[f"{nm}.stc_per_bone.{mode}"] = R.segments_time_consistency_per_bone(jc, sk, mode).numpy()
        # the evaluation call flattens the batch into the time axis (main_h36m_lifting.py:950-959)
        out[f"{nm}.stc_flat.std"] = R.segments_time_consistency(jc.permute(1, 2, 0, 3).reshape(1, 3, J, -1), sk, "std").numpy()
        for sq in (False, True):
            for mode in ("average", "sum"):
                out[f"{nm}.sym.{mode}.{int(sq)}"] = R.sagittal_symmetry(jc, sk, mode, squared=sq).numpy()
                out[f"{nm}.sym_per_bone.{mode}.{int(sq)}"] = R.sagittal_symmetry_per_bone(jc, sk, mode, squared=sq).numpy()
        gj = gt.permute(0, 3, 2, 1)
        for signed in (False, True):
            for mode in ("average", "sum"):
                out[f"{nm}.len_err.{mode}.{int(signed)}"] = E.segments_len_err(jc, gj, sk, mode, signed=signed).numpy()
        for mode in ("average", "sum"):
            out[f"{nm}.mpjpe.{mode}"] = E.mpjpe_error(x, gt, mode).numpy()
            out[f"{nm}.mse.{mode}"] = E.mse_error(x, gt, mode).numpy()
            out[f"{nm}.jw_err.{mode}"] = E.jointwise_error(x, gt, mode).numpy()
            out[f"{nm}.jw_mse.{mode}"] = E.jointwise_mse(x, gt, mode).numpy()
        out[f"{nm}.vel"] = mean_velocity_error(predicted=x, target=gt, squared=False, axis=1).numpy()
        out[f"{nm}.vel_sq"] = mean_velocity_error(predicted=x, target=gt, squared=True, axis=1).numpy()
        xf, gf = x.reshape(-1, J, 3).numpy(), gt.reshape(-1, J, 3).numpy()
        mask = (torch.rand(B * L, J, generator=torch.Generator().manual_seed(5)) > 0.2).numpy()
        out["mask"] = mask
        for al in ("none", "scale", "procrustes"):
            out[f"{nm}.pck.{al}"] = np.float64(P.keypoint_3d_pck(xf, gf, mask=None, alignment=al, threshold=150))
            out[f"{nm}.auc.{al}"] = np.float64(P.keypoint_3d_auc(xf, gf, mask=None, alignment=al))
        out[f"{nm}.pck.procrustes.masked"] = np.float64(P.keypoint_3d_pck(xf, gf, mask=mask, alignment="procrustes", threshold=150))
        out[f"{nm}.p_mpjpe"] = np.float64(E.p_mpjpe(x.clone(), gt.clone()))
        out[f"{nm}.pck.masked"] = np.float64(P.keypoint_3d_pck(xf, gf, mask=mask, alignment="none", threshold=150))
        out[f"{nm}.auc.masked"] = np.float64(P.keypoint_3d_auc(xf, gf, mask=mask, alignment="none"))
        out[f"{nm}.pck.thr80"] = np.float64(P.keypoint_3d_pck(xf, gf, mask=None, alignment="none", threshold=80))
    np.savez_compressed(os.path.join(OUT, "metrics.npz"), **out)
    print("metrics: ok", len(out), "arrays")


def gen_windows_fixture(ref):
    """Input pipeline: windows produced by the reference's PoseSequenceGenerator (+ PoseFlip) on seeded sequences
    of uneven lengths.  The source arrays are float64 so that the reference's in-place flip (which aliases float32 dataset arrays)
    cannot modify them between items."""
    from mh_so3_hpe.data.generators import PoseSequenceGenerator
    from mh_so3_hpe.augmentations.transforms import PoseFlip
    g = np.random.default_rng(123)
    lens = [61, 28, 100, 35, 54]      # every sequence longer than the window (the reference's random start needs len > seq_len)
    p3 = [g.normal(size=(n, 17, 3)) for n in lens]
    p2 = [g.normal(size=(n, 17, 2)) for n in lens]
    out = {"lens": np.array(lens)}
    for i, (a, b) in enumerate(zip(p3, p2)):
        out[f"p3.{i}"], out[f"p2.{i}"] = a.astype(np.float32), b.astype(np.float32)
    p3 = [out[f"p3.{i}"].astype(np.float64) for i in range(len(lens))]      # exactly representable float32 values
    p2 = [out[f"p2.{i}"].astype(np.float64) for i in range(len(lens))]
    import contextlib, io
    cases = {"strided_drop": dict(random_start=False, drop_last=True, flip=False),
             "strided_pad": dict(random_start=False, drop_last=False, flip=False),
             "random_flip": dict(random_start=True, drop_last=True, flip=True),
             "strided_pad_flip": dict(random_start=False, drop_last=False, flip=True)}
    for nm, c in cases.items():
        with contextlib.redirect_stdout(io.StringIO()):
            gen = PoseSequenceGenerator(p3, p2, None, seq_len=27, random_start=c["random_start"], drop_last=c["drop_last"],
                                        miss_type="no_miss", transform=PoseFlip(ref["sk"], 0.5) if c["flip"] else None)
        torch.manual_seed(2024)
        xs, ys = [], []
        for i in range(len(gen)):
            x, y = gen[i]
            xs.append(x.clone().numpy()); ys.append(y.clone().numpy())
        out[f"{nm}.X"], out[f"{nm}.y"] = np.stack(xs), np.stack(ys)
        out[f"{nm}.len"] = np.int64(len(gen))
    # occlusion patterns (generators.py:160-216): numpy's global RNG
    for mt in ("random", "random_left_arm_right_leg", "structured_joint", "structured_frame", "noisy", "all"):
        with contextlib.redirect_stdout(io.StringIO()):
            gen = PoseSequenceGenerator(p3, p2, None, seq_len=27, random_start=True, drop_last=True, miss_type=mt, miss_rate=0.3,
                                        noise_sigma=0.05, transform=PoseFlip(ref["sk"], 0.5))
        torch.manual_seed(77)
        np.random.seed(99)
        xs, ys = [], []
        for i in range(len(gen)):
            x, y = gen[i]
            xs.append(x.clone().numpy()); ys.append(y.clone().numpy())
        out[f"miss.{mt}.X"], out[f"miss.{mt}.y"] = np.stack(xs), np.stack(ys)
    np.savez_compressed(os.path.join(OUT, "windows.npz"), **out)
    print("windows: ok", {k: v.shape for k, v in out.items() if k.endswith(".X")})


def _write_raw_dataset_files(d, raw):
    """The on-disk formats of the reference's loaders (h36m_lifting.py:620, utils.py:13-14, dataset_3dhp.py:153,185): npz files whose
    single entry is a pickled dict."""
    np.savez_compressed(os.path.join(d, "data_3d_h36m.npz"), positions_3d=raw["h36m3"])
    np.savez_compressed(os.path.join(d, "data_2d_h36m_gt.npz"), positions_2d=raw["h36m2"], metadata={"num_joints": 17})
    np.savez_compressed(os.path.join(d, "data_train_3dhp.npz"), data=raw["hp_train"])
    np.savez_compressed(os.path.join(d, "data_test_3dhp.npz"), data=raw["hp_test"])


def gen_datasets_fixture(ref):
    """Dataset ingest (on-disk formats): tiny synthetic files in the reference's formats, read by the reference's own
    Human36mDataset / read_3d_data / create_2d_data / fetch and Dataset3DHP; the raw arrays and what the reference made of them are the
    fixture.  Also dumps the Human3.6M camera calibration (dataset facts: the public calibration every VideoPose3D-derived loader
    carries) to manipose_amd/data/h36m_cameras.json."""
    import contextlib, copy, io, json, tempfile
    from types import SimpleNamespace as NS
    import mh_so3_hpe.data.h36m_lifting as H
    from mh_so3_hpe.data.utils import create_2d_data, read_3d_data, fetch
    from mh_so3_hpe.data.dataset_3dhp import Dataset3DHP
    g = np.random.default_rng(7)
    acts = {"S1": {"Walking": 33, "Eating 1": 29}, "S9": {"Walking 1": 31, "Photo": 28}, "S11": {"Walking": 30, "SittingDown 2": 35}}
    raw = {"h36m3": {s: {a: (0.5 * g.normal(size=(n, 32, 3)) + [0.2, -0.1, 1.0]).astype(np.float32) for a, n in d.items()}
                     for s, d in acts.items()},
           "h36m2": {s: {a: [g.uniform(0, 1000, size=(n, 17, 2)).astype(np.float32) for _ in range(4)] for a, n in d.items()}
                     for s, d in acts.items()}}
    hp_train, hp_test = {}, {}
    for seq, n in (("S1 Seq1", 31), ("S2 Seq1", 29)):
        hp_train[seq] = [{str(c): {"data_3d": (600 * g.normal(size=(n, 17, 3)) + [0, 0, 3500]).astype(np.float32),
                                   "data_2d": g.uniform(0, 2048, size=(n, 17, 2)).astype(np.float32)} for c in (0, 2, 7)}]
    for seq, n in (("TS1", 34), ("TS5", 30)):
        valid = (g.uniform(size=n) > 0.2).astype(np.float32)
        hp_test[seq] = {"data_3d": (600 * g.normal(size=(n, 17, 3)) + [0, 0, 3500]).astype(np.float32),
                        "data_2d": g.uniform(0, 1920 if seq == "TS5" else 2048, size=(n, 17, 2)).astype(np.float32), "valid": valid}
    raw["hp_train"], raw["hp_test"] = hp_train, hp_test
    out = {}
    for s, d in raw["h36m3"].items():
        for a, v in d.items():
            out[f"raw.h36m3|{s}|{a}"] = v
            for c, k in enumerate(raw["h36m2"][s][a]):
                out[f"raw.h36m2|{s}|{a}|{c}"] = k
    for seq, lst in hp_train.items():
        for c, dd in lst[0].items():
            out[f"raw.hptrain|{seq}|{c}|3d"], out[f"raw.hptrain|{seq}|{c}|2d"] = dd["data_3d"], dd["data_2d"]
    for seq, dd in hp_test.items():
        out[f"raw.hptest|{seq}|3d"], out[f"raw.hptest|{seq}|2d"], out[f"raw.hptest|{seq}|valid"] = dd["data_3d"], dd["data_2d"], dd["valid"]
    with tempfile.TemporaryDirectory() as d:
        _write_raw_dataset_files(d, copy.deepcopy(raw))
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            ds = H.Human36mDataset(os.path.join(d, "data_3d_h36m.npz"), n_joints=17)
            ds = read_3d_data(ds)
            kp = create_2d_data(os.path.join(d, "data_2d_h36m_gt.npz"), ds)
        sk = ds.skeleton
        out["h36m.parents"] = np.array(sk.parents)
        out["h36m.joints_left"], out["h36m.joints_right"] = np.array(sk.joints_left), np.array(sk.joints_right)
        for nm, subjects, filt, stride in (("all", ["S1", "S9"], None, 1), ("walk_s2", ["S9", "S11"], ["walking"], 2),
                                           ("s11_sit", ["S11"], ["sittingdown"], 1)):
            p3, p2, actions, cams = fetch(subjects, ds, kp, filt, stride)
            out[f"h36m.{nm}.n"] = np.int64(len(p3))
            for i in range(len(p3)):
                out[f"h36m.{nm}.p3.{i}"], out[f"h36m.{nm}.p2.{i}"] = np.asarray(p3[i]), np.asarray(p2[i])
                out[f"h36m.{nm}.cam.{i}"] = np.asarray(cams[i][0])
            out[f"h36m.{nm}.actions"] = np.array([a[0] for a in actions])
        cfg = NS(data=NS(dataset="3dhp", keypoints="gt", actions="*", downsample=1, seq_len=27, pad=0, out_all=True),
                 train=NS(flip_aug=True, batch_size=2, batch_size_test=2, tta=True))
        for nm, tr in (("train", True), ("test", False)):
            with contextlib.redirect_stdout(io.StringIO()):
                hp = Dataset3DHP(config=cfg, root_path=d + "/", train=tr)
            out[f"hp.{nm}.n"] = np.int64(len(hp.poses))
            for i in range(len(hp.poses)):
                out[f"hp.{nm}.p3.{i}"], out[f"hp.{nm}.p2.{i}"] = np.asarray(hp.poses[i]), np.asarray(hp.poses_2d[i])
    np.savez_compressed(os.path.join(OUT, "datasets.npz"), **out)
    calib = {"intrinsic": [{k: v for k, v in c.items() if k != "azimuth"} for c in H.h36m_cameras_intrinsic_params],
             "extrinsic": {s: [dict(c) for c in cams] for s, cams in H.h36m_cameras_extrinsic_params.items()}}
    with open(os.path.join(os.path.dirname(HERE), "manipose_amd", "data", "h36m_cameras.json"), "w") as f:
        json.dump(calib, f, indent=0, separators=(",", ":"))
    print("datasets: ok", {k: v.shape for k, v in out.items() if k.endswith(".p3.0")},
          {k: (v.dtype, v.shape) for k, v in out.items() if k.endswith(".p2.0") or k.endswith(".cam.0")})


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = import_reference()
    if len(sys.argv) > 1 and sys.argv[1] == "windows":       # regenerate the input-pipeline fixture only
        gen_windows_fixture(ref)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "mixste":        # regenerate the bare-MixSTE fixture only
        gen_mixste_fixture(ref)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "rot4":          # regenerate the 4-D rotation fixtures only
        gen_rot4_decoder_fixture(ref)
        gen_model_fixture(ref, "rmcl_tiny_rot4", dict(T=9, J=17, num_bones=16, C_rot=32, depth_rot=2, heads_rot=4, C_seg=16, depth_seg=1,
                                                      heads_seg=4, n_hyp=3, rot_dim=4), B=2, seed=21)
        gen_model_fixture(ref, "manifold_k1_rot4", dict(T=9, J=17, num_bones=16, C_rot=32, depth_rot=2, heads_rot=4, C_seg=16, depth_seg=1,
                                                        heads_seg=4, n_hyp=0, rot_dim=4), B=2, seed=23)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "loss_sq":       # regenerate the squared-loss fixture only
        gen_sq_loss_fixture(ref)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "datasets":      # regenerate the dataset-ingest fixture only
        gen_datasets_fixture(ref)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "metrics":       # regenerate the analytics fixture only
        gen_metrics_fixture(ref)
        return
    tiny = dict(T=9, J=17, num_bones=16, C_rot=32, depth_rot=2, heads_rot=4, C_seg=16, depth_seg=1, heads_seg=4, n_hyp=3)
    small = dict(T=27, J=17, num_bones=16, C_rot=64, depth_rot=2, heads_rot=8, C_seg=32, depth_seg=2, heads_seg=8, n_hyp=5)
    k1 = dict(T=27, J=17, num_bones=16, C_rot=64, depth_rot=2, heads_rot=8, C_seg=32, depth_seg=1, heads_seg=8, n_hyp=0)
    gen_model_fixture(ref, "rmcl_tiny", tiny, B=2, seed=11)
    gen_model_fixture(ref, "rmcl_small", small, B=2, seed=12)
    gen_model_fixture(ref, "manifold_k1", k1, B=2, seed=13)
    gen_model_fixture(ref, "rmcl_tiny_droppath", tiny, B=3, seed=14, drop_path_rate=0.5, train=True)
    gen_model_fixture(ref, "rmcl_tiny_rot4", dict(tiny, rot_dim=4), B=2, seed=21)
    gen_model_fixture(ref, "manifold_k1_rot4", dict(tiny, n_hyp=0, rot_dim=4), B=2, seed=23)
    gen_mixste_fixture(ref)
    gen_decoder_fixture(ref)
    gen_loss_fixture(ref)
    gen_metrics_fixture(ref)
    gen_windows_fixture(ref)
    gen_datasets_fixture(ref)
    # default initialisation under seed 42 (the product's constructors must consume the RNG identically)
    torch.manual_seed(42)
    m0 = build_ref_model(ref, small, 0.1)
    np.savez_compressed(os.path.join(OUT, "init_seed42_small.npz"), **{k: v.numpy() for k, v in m0.state_dict().items()})
    # parameter-count known answers
    counts = {}
    for nm, T, K in (("rmcl_T243_K5", 243, 5), ("rmcl_T81_K5", 81, 5), ("manifold_T27", 27, 0)):
        cfg = dict(orc.FULL_CFG, T=T, n_hyp=K)
        m = build_ref_model(ref, cfg, 0.1)
        counts[nm] = sum(p.numel() for p in m.parameters())
        if nm == "rmcl_T243_K5":
            keys = sorted(f"{k}|{'x'.join(map(str, v.shape))}" for k, v in m.state_dict().items())
            with open(os.path.join(OUT, "state_dict_keys_T243_K5.txt"), "w") as f:
                f.write("\n".join(keys) + "\n")
    np.savez(os.path.join(OUT, "param_counts.npz"), **{k: np.int64(v) for k, v in counts.items()})
    print(counts)


if __name__ == "__main__":
    main()
