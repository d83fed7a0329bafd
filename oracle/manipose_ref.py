"""CPU oracle for the ManiPose lifting hot path (TEST INFRASTRUCTURE ONLY).

This file is a plain torch-CPU fp32 restatement of the reference algorithm
(cedricrommel/manipose @ 2025-01-17). It is the *checker* for the HIP path:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it. The product package ``manipose_amd`` never imports it and
has no CPU fallback.

Parity pin: the functions below are checked against golden vectors generated
by importing the reference itself in the build container
(``oracle/gen_golden.py`` -> ``tests/golden/*.npz``, see
``tests/test_oracle_golden.py``). The reference holds no tests/fixtures of its
own for this path, so those generated vectors plus the
closed-form known answers of SURVEY.md section 4 are the pin.

Every function takes a flat ``state`` dict using the reference's state-dict
key names so that fixtures, oracle and product share
one weight format. All citations are ``path:line`` under ``/root/reference``.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# ---------------------------------------------------------------------------
# Skeleton tables (hpe/mh_so3_hpe/data/skeleton.py:87-120; 17-joint tree literal
# hpe/mh_so3_hpe/data/dataset_3dhp.py:132-138; T-pose operators
# hpe/mh_so3_hpe/data/h36m_lifting.py:40-57)
# ---------------------------------------------------------------------------
H36M_PARENTS = [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 9, 8, 11, 12, 8, 14, 15]
H36M_JOINTS_LEFT = [4, 5, 6, 11, 12, 13]
H36M_JOINTS_RIGHT = [1, 2, 3, 14, 15, 16]
# unit translation taking a joint's parent to the joint in T-pose, index = joint
T_POSE_OPERATORS = {
    1: (1.0, 0.0, 0.0), 2: (0.0, -1.0, 0.0), 3: (0.0, -1.0, 0.0),
    4: (-1.0, 0.0, 0.0), 5: (0.0, -1.0, 0.0), 6: (0.0, -1.0, 0.0),
    7: (0.0, 1.0, 0.0), 8: (0.0, 1.0, 0.0), 9: (0.0, 1.0, 0.0),
    10: (0.0, 1.0, 0.0), 11: (-1.0, 0.0, 0.0), 12: (-1.0, 0.0, 0.0),
    13: (-1.0, 0.0, 0.0), 14: (1.0, 0.0, 0.0), 15: (1.0, 0.0, 0.0),
    16: (1.0, 0.0, 0.0),
}
# hpe/mh_so3_hpe/metrics/losses.py:6-8
STANDARD_H36M_WEIGHTS = [1, 1, 2.5, 2.5, 1, 2.5, 2.5, 1, 1, 1, 1.5, 1.5, 4, 4, 1.5, 4, 4]


def has_children(parents):
    """skeleton.py:88-91."""
    out = [False] * len(parents)
    for p in parents:
        if p != -1:
            out[p] = True
    return out


# ---------------------------------------------------------------------------
# Transformer backbone (hpe/mh_so3_hpe/architectures/mix_ste.py)
# ---------------------------------------------------------------------------
def attention(x: Tensor, st: Dict[str, Tensor], pre: str, num_heads: int,
              scale: Optional[float] = None) -> Tensor:
    """Attention.forward, mix_ste.py:255-282 (comb=False branch)."""
    B, N, C = x.shape
    d = C // num_heads
    scale = scale if scale is not None else d ** -0.5          # mix_ste.py:243-244
    qkv = F.linear(x, st[pre + "qkv.weight"], st[pre + "qkv.bias"])
    qkv = qkv.reshape(B, N, 3, num_heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)) * scale
    attn = attn.softmax(dim=-1)
    x = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(x, st[pre + "proj.weight"], st[pre + "proj.bias"])


def mlp(x: Tensor, st: Dict[str, Tensor], pre: str) -> Tensor:
    """Mlp.forward, mix_ste.py:216-222 (GELU = exact erf form, drop p=0)."""
    x = F.linear(x, st[pre + "fc1.weight"], st[pre + "fc1.bias"])
    x = F.gelu(x)
    return F.linear(x, st[pre + "fc2.weight"], st[pre + "fc2.bias"])


def block(x: Tensor, st: Dict[str, Tensor], pre: str, num_heads: int,
          mask_attn: Optional[Tensor] = None, mask_mlp: Optional[Tensor] = None, scale: Optional[float] = None, rs: float = 1.0) -> Tensor:
    """Block.forward, mix_ste.py:352-368; rs = residual_scale (:330: 1, or 1 / sqrt(depth) with mup), scale = the attention's
    softmax scale (:243-244: head_dim ** -0.5, or 1 / head_dim with mup).

    ``mask_*`` are the DropPath multipliers of timm.models.layers.DropPath
    (third party, timm==0.9.16, not vendored in the reference; parity unpinned
    for train-mode randomness): shape (x.shape[0],), values 0 or 1/keep_prob.
    ``None`` = identity (eval mode or drop_prob == 0).
    """
    C = x.shape[-1]
    a = attention(F.layer_norm(x, (C,), st[pre + "norm1.weight"], st[pre + "norm1.bias"], 1e-6),
                  st, pre + "attn.", num_heads, scale)
    if mask_attn is not None:
        a = a * mask_attn.view(-1, 1, 1)
    x = x * rs + a
    m = mlp(F.layer_norm(x, (C,), st[pre + "norm2.weight"], st[pre + "norm2.bias"], 1e-6),
            st, pre + "mlp.")
    if mask_mlp is not None:
        m = m * mask_mlp.view(-1, 1, 1)
    return x * rs + m


def mixste_backbone(x: Tensor, st: Dict[str, Tensor], pre: str, depth: int,
                    num_heads: int, masks: Optional[Dict[str, Tensor]] = None,
                    embed: bool = True, scale: Optional[float] = None, rs: float = 1.0) -> Tensor:
    """MixSTE.STE_forward / TTE_foward / ST_foward, mix_ste.py:128-173.

    x: (B, L, J, Cin) -> (B, L, J, C). ``masks`` maps
    ``f"{pre}STEblocks.{i}.attn"`` / ``.mlp`` (and TTEblocks) to DropPath
    multipliers. ``embed=False`` is BonesMixSTE's Identity patch embedding
    (manifold_mix_ste.py:133).
    """
    masks = masks or {}
    B, L, J, _ = x.shape

    def snorm(z):
        C = z.shape[-1]
        return F.layer_norm(z, (C,), st[pre + "Spatial_norm.weight"], st[pre + "Spatial_norm.bias"], 1e-6)

    def tnorm(z):
        C = z.shape[-1]
        return F.layer_norm(z, (C,), st[pre + "Temporal_norm.weight"], st[pre + "Temporal_norm.bias"], 1e-6)

    def blk(z, name):
        return block(z, st, pre + name + ".", num_heads,
                     masks.get(pre + name + ".attn"), masks.get(pre + name + ".mlp"), scale, rs)

    # STE_forward :128-145
    x = x.reshape(B * L, J, -1)
    if embed:
        x = F.linear(x, st[pre + "Spatial_patch_to_embedding.weight"],
                     st[pre + "Spatial_patch_to_embedding.bias"])
    x = x + st[pre + "Spatial_pos_embed"]
    x = blk(x, "STEblocks.0")
    x = snorm(x)
    C = x.shape[-1]
    x = x.reshape(B, L, J, C).permute(0, 2, 1, 3).reshape(B * J, L, C)
    # TTE_foward :147-155
    x = x + st[pre + "Temporal_pos_embed"]
    x = blk(x, "TTEblocks.0")
    x = tnorm(x)
    x = x.reshape(B, J, L, C).permute(0, 2, 1, 3)                    # (B L J C)
    # ST_foward :157-173
    for i in range(1, depth):
        x = x.reshape(B * L, J, C)
        x = blk(x, f"STEblocks.{i}")
        x = snorm(x)
        x = x.reshape(B, L, J, C).permute(0, 2, 1, 3).reshape(B * J, L, C)
        x = blk(x, f"TTEblocks.{i}")
        x = tnorm(x)
        x = x.reshape(B, J, L, C).permute(0, 2, 1, 3)
    return x.contiguous()


def mup_scales(embed_dim: int, num_heads: int, depth: int, readout: float = 1.0) -> dict:
    """Scales of a MixSTE built with mup=True: attention 1 / head_dim (mix_ste.py:243), residual 1 / sqrt(depth) (:330), and the
    input multiplier output_mult / width_mult of its mup.MuReadout head (:118-121; third party, mup==1.0.0: parity unpinned)."""
    return dict(scale=1.0 / (embed_dim // num_heads), rs=depth ** -0.5, readout=readout)


def mixste_forward(x: Tensor, st, pre: str, depth: int, num_heads: int, masks=None, scale=None, rs: float = 1.0, readout: float = 1.0) -> Tensor:
    """MixSTE.forward, mix_ste.py:175-191: backbone + head Sequential(LN eps 1e-5, Linear | MuReadout)."""
    h = mixste_backbone(x, st, pre, depth, num_heads, masks, scale=scale, rs=rs)
    C = h.shape[-1]
    h = F.layer_norm(h, (C,), st[pre + "head.0.weight"], st[pre + "head.0.bias"], 1e-5)
    return F.linear(h * readout, st[pre + "head.1.weight"], st[pre + "head.1.bias"])


def rmcl_rot_forward(x: Tensor, st, pre: str, depth: int, num_heads: int, n_hyp: int,
                     masks=None, readout: float = 1.0) -> Tuple[Tensor, Tensor]:
    """RMCLRotMixSTE.forward (rmcl_manifold_mix_ste.py:239-264) with MCLHead.forward (:291-298).  The backbone is never built with mup
    (:208-223 does not forward the flag); with mup the prediction heads are MuReadouts (input multiplier `readout`; the score head's
    fan-in is the joint count, not a width: multiplier 1)."""
    h = mixste_backbone(x, st, pre, depth, num_heads, masks)
    C = h.shape[-1]
    preds, logits = [], []
    for k in range(n_hyp):
        hp = f"{pre}head.{k}."
        z = F.layer_norm(h, (C,), st[hp + "norm.weight"], st[hp + "norm.bias"], 1e-5)
        z = F.linear(z * readout, st[hp + "prediction_head.weight"], st[hp + "prediction_head.bias"])
        preds.append(z[..., :-1])                                                  # (B, L, J, 6)
        logits.append(F.linear(z[..., -1], st[hp + "score_head.weight"], st[hp + "score_head.bias"]))
    hyp = torch.stack(preds, dim=1)                                                # (B, H, L, J, 6)
    logit = torch.stack(logits, dim=1)                                             # (B, H, L, 1)
    return hyp, logit.softmax(dim=1)


def bones_forward(x: Tensor, st, pre: str, depth: int, num_heads: int, num_bones: int,
                  masks=None, scale=None, rs: float = 1.0, readout: float = 1.0) -> Tensor:
    """BonesMixSTE.forward, manifold_mix_ste.py:139-154. x (B,L,J,2) -> (B,S,1)."""
    B, L, _, _ = x.shape
    z = F.linear(x.reshape(B * L, -1), st[pre + "joints_to_segments_proj.weight"],
                 st[pre + "joints_to_segments_proj.bias"])
    z = z.reshape(B, L, num_bones, -1)
    h = mixste_backbone(z, st, pre, depth, num_heads, masks, embed=False, scale=scale, rs=rs)
    C = h.shape[-1]
    h = F.layer_norm(h, (C,), st[pre + "head.0.weight"], st[pre + "head.0.bias"], 1e-5)
    h = F.linear(h * readout, st[pre + "head.1.weight"], st[pre + "head.1.bias"])  # (B, L, S, 1)
    return h.mean(dim=1)


# ---------------------------------------------------------------------------
# Decoder (pose_decoder.py, utils/rotation_tools.py, utils/forward_kinematics.py)
# ---------------------------------------------------------------------------
def normalize_vector(v: Tensor) -> Tensor:
    """rotation_tools.py:6-17 (same arithmetic; the reference's hard-coded .cuda() dropped)."""
    mag = torch.sqrt(v.pow(2).sum(1))
    mag = torch.max(mag, torch.tensor([1e-8], dtype=v.dtype, device=v.device))
    return v / mag.view(-1, 1)


def cross_product(u: Tensor, v: Tensor) -> Tensor:
    """rotation_tools.py:21-32."""
    i = u[:, 1] * v[:, 2] - u[:, 2] * v[:, 1]
    j = u[:, 2] * v[:, 0] - u[:, 0] * v[:, 2]
    k = u[:, 0] * v[:, 1] - u[:, 1] * v[:, 0]
    return torch.stack((i, j, k), dim=1)


def rotation_from_ortho6d(p: Tensor) -> Tensor:
    """compute_rotation_matrix_from_ortho6d, rotation_tools.py:35-57. (N,6)->(N,3,3), x|y|z as columns."""
    x = normalize_vector(p[:, 0:3])
    z = normalize_vector(cross_product(x, p[:, 3:6]))
    y = cross_product(z, x)
    return torch.stack((x, y, z), dim=2)


def rotation_from_ortho4d(p: Tensor) -> Tensor:
    """compute_rotation_matrix_from_ortho4d, rotation_tools.py:60-116 (model.rot_dim=4). (N,4)->(N,3,3) = R_theta @ R_phi, each a
    planar rotation built from a normalised 2-vector: theta_y = (c1, s1, 0), theta_x = theta_y x e_z; phi_y = (0, c2, s2),
    phi_z = e_x x phi_y; columns [theta_x | theta_y | e_z] and [e_x | phi_y | phi_z]."""
    n = p.shape[0]
    cs_theta, cs_phi = normalize_vector(p[:, 0:2]), normalize_vector(p[:, 2:4])
    zero = torch.zeros(n, 1, dtype=p.dtype)
    theta_y = torch.cat([cs_theta, zero], dim=1)
    theta_z = torch.tensor([0.0, 0.0, 1.0], dtype=p.dtype).expand(n, -1)
    theta_x = cross_product(theta_y, theta_z)
    phi_y = torch.cat([zero, cs_phi], dim=1)
    phi_x = torch.tensor([1.0, 0.0, 0.0], dtype=p.dtype).expand(n, -1)
    phi_z = cross_product(phi_x, phi_y)
    return torch.stack((theta_x, theta_y, theta_z), dim=2).bmm(torch.stack((phi_x, phi_y, phi_z), dim=2))


def build_t_pose(bones_length: Tensor, parents=H36M_PARENTS) -> Tensor:
    """PoseDecoder.build_t_pose_from_bone_lengths, pose_decoder.py:98-120. (N,S,1)->(N,J,3)."""
    N = bones_length.shape[0]
    J = len(parents)
    cols = [torch.zeros(N, 3, dtype=bones_length.dtype)]
    for b in range(J - 1):
        op = torch.tensor(T_POSE_OPERATORS[b + 1], dtype=bones_length.dtype)
        cols.append(cols[parents[b + 1]] + op * bones_length[:, b])
    return torch.stack(cols, dim=1)


def forward_kinematics(t_pose: Tensor, rotations: Tensor, root: Tensor, parents=H36M_PARENTS) -> Tensor:
    """forward_kinematics.py:6-48. t_pose (N,J,3), rotations (N,J,3,3), root (N,3) -> (N,J,3)."""
    pos, rot = [], []
    for j in range(len(parents)):
        p = parents[j]
        if p == -1:
            pos.append(root)
            rot.append(rotations[:, 0])
        else:
            off = (t_pose[:, j] - t_pose[:, p]).view(-1, 3, 1)
            rw = rot[p].matmul(rotations[:, j])
            pos.append(rw.matmul(off).view(-1, 3) + pos[p])
            rot.append(rw)       # leaf rotations are simply unused (:41-46)
    return torch.stack(pos, dim=1)


def pose_decoder(rot6d: Tensor, bones_lengths: Tensor, parents=H36M_PARENTS) -> Tensor:
    """PoseDecoder.forward, pose_decoder.py:32-55. rot6d (N,J,6) or (N,J,4) (:69-82); bones_lengths (B,S,1), N % B == 0."""
    N, J, D = rot6d.shape
    B = bones_lengths.shape[0]
    assert N % B == 0
    L = N // B
    bl = torch.stack([bones_lengths] * L, dim=1).reshape(N, -1, 1)                 # :85-96
    assert D in (4, 6), f"Unsupported rotations representation dimension: {D}"
    R = (rotation_from_ortho6d if D == 6 else rotation_from_ortho4d)(rot6d.reshape(-1, D)).reshape(N, J, 3, 3)   # :57-83
    return forward_kinematics(build_t_pose(bl, parents), R, torch.zeros(N, 3, dtype=rot6d.dtype), parents)


# ---------------------------------------------------------------------------
# Whole models
# ---------------------------------------------------------------------------
def rmcl_manifold_forward(x: Tensor, st, cfg: dict, masks=None) -> Tuple[Tensor, Tensor]:
    """RMCLManifoldMixSTE.forward, rmcl_manifold_mix_ste.py:83-106.

    cfg keys: depth_rot, heads_rot, depth_seg, heads_seg, n_hyp, num_bones.
    Returns poses (B,H,L,J,3), scores (B,H,L,1).
    """
    B, L, J, _ = x.shape
    H = cfg["n_hyp"]
    # cfg["mup"] (optional) = {"rot": mup_scales(...), "seg": mup_scales(...)}: model.mup=True (only `readout` of "rot" applies here)
    mup = cfg.get("mup") or {}
    rot, scores = rmcl_rot_forward(x, st, "rotations_module.", cfg["depth_rot"], cfg["heads_rot"], H, masks,
                                   readout=mup.get("rot", {}).get("readout", 1.0))
    bl = bones_forward(x, st, "segments_module.", cfg["depth_seg"], cfg["heads_seg"], cfg["num_bones"], masks, **mup.get("seg", {}))
    poses = pose_decoder(rot.reshape(B * H * L, J, -1), bl)
    return poses.reshape(B, H, L, J, 3), scores


def manifold_forward(x: Tensor, st, cfg: dict, masks=None) -> Tensor:
    """ManifoldMixSTE.forward, manifold_mix_ste.py:75-88 (single hypothesis). -> (B,L,J,3)."""
    B, L, J, _ = x.shape
    mup = cfg.get("mup") or {}
    rot = mixste_forward(x, st, "rotations_module.", cfg["depth_rot"], cfg["heads_rot"], masks, **mup.get("rot", {}))
    bl = bones_forward(x, st, "segments_module.", cfg["depth_seg"], cfg["heads_seg"], cfg["num_bones"], masks, **mup.get("seg", {}))
    return pose_decoder(rot.reshape(B * L, J, -1), bl).reshape(B, L, J, 3)


# ---------------------------------------------------------------------------
# Losses (hpe/mh_so3_hpe/metrics/losses.py, regularizations.py:160-174,
# loss assembly hpe/main_h36m_lifting.py:101-209)
# ---------------------------------------------------------------------------
def weighted_mpjpe_per_hyp(hyp: Tensor, y: Tensor, weights: Optional[Tensor]) -> Tensor:
    """_l2_loss_per_hyp (losses.py:104-123) -> weighted_mpjpe_loss(dims=[3]) (:14-43). -> (B,H,L)."""
    w = torch.ones(y.shape[-2]) if weights is None else weights
    return (w[None, None, :] * torch.norm(hyp - y[:, None], p=2, dim=-1)).mean(dim=3)


def weighted_mse_per_hyp(hyp: Tensor, y: Tensor, weights: Optional[Tensor]) -> Tensor:
    """_l2_loss_per_hyp(squared=True) (losses.py:110-116) -> weighted_mse_loss(dims=[4, 3]) (:46-72): mean over the coordinates, then
    over the joints.  -> (B,H,L).  (With weights=None the reference returns a scalar F.mse_loss here and the WTA min fails; the
    squared WTA loss exists only with train.w_loss=True.  Unit weights stand in for None below.)"""
    w = torch.ones(y.shape[-2]) if weights is None else weights
    return (w[None, None, :, None] * (hyp - y[:, None]) ** 2).mean(dim=4).mean(dim=3)


def wta_l2_loss_and_activate_head(hyp, y, weights=None, squared=False):
    """losses.py:126-138."""
    return torch.min(weighted_mse_per_hyp(hyp, y, weights) if squared else weighted_mpjpe_per_hyp(hyp, y, weights), dim=1)


def wta_with_scoring_loss(hyp, scores, y, beta, weights=None, squared=False):
    """losses.py:141-170. Returns (wta.mean() + beta*bce, beta*bce)."""
    wta, idx = wta_l2_loss_and_activate_head(hyp, y, weights, squared)
    B, H, L = hyp.shape[:3]
    gt = F.one_hot(idx, H).permute(0, 2, 1).to(scores.dtype)                       # (B, H, L)
    bce = F.binary_cross_entropy(scores.view(B, H, L), gt)
    return wta.mean() + beta * bce, beta * bce


def mean_velocity_error(pred, target, axis, squared=False):
    """losses.py:75-101."""
    if pred.dim() > target.dim():
        target = target.unsqueeze(1).expand_as(pred)
    dv = torch.diff(pred, dim=axis) - torch.diff(target, dim=axis)
    return (dv ** 2).mean() if squared else torch.norm(dv, dim=-1).mean()


def smoothness_regularization(pred, weights, axis):
    """regularizations.py:160-174."""
    v = torch.diff(pred, dim=axis)
    w = torch.ones(v.shape[-2]) if weights is None else weights
    return (w[None, None, :, None] * v ** 2).mean()


def weighted_mpjpe_loss(pred, target, weights=None):
    """losses.py:14-43 with dims=None (single-hypothesis wloss)."""
    w = torch.ones(target.shape[-2]) if weights is None else weights
    return (w[None, None, :] * torch.norm(pred - target, p=2, dim=-1)).mean()


def weighted_mse_loss(pred, target, weights=None):
    """losses.py:46-72 with dims=None (single-hypothesis wloss under train.sq_loss)."""
    w = torch.ones(target.shape[-2]) if weights is None else weights
    return (w[None, None, :, None] * (pred - target) ** 2).mean()


DEFAULT_TRAIN_CFG = dict(w_loss=True, vel_loss=2.0, smooth_reg=0.5, rmcl_score_reg=0.1)  # conf/config.yaml:32-38


def rmcl_training_loss(poses, scores, y, train_cfg=DEFAULT_TRAIN_CFG):
    """make_loss + compute_and_acc_loss for the rmcl model, main_h36m_lifting.py:101-209.

    Returns (total, dict of the four terms). time axis = 2.
    """
    w = torch.tensor(STANDARD_H36M_WEIGHTS, dtype=poses.dtype) if train_cfg["w_loss"] else None
    terms = {}
    sq = bool(train_cfg.get("sq_loss", False))                 # conf/config.yaml:31, forwarded as `squared=` by make_loss
    terms["wloss"] = wta_l2_loss_and_activate_head(poses, y, w, sq)[0].mean()
    terms["score_reg"] = wta_with_scoring_loss(poses, scores, y, train_cfg["rmcl_score_reg"], w, sq)[1]
    if train_cfg["vel_loss"] > 0:
        terms["vloss"] = train_cfg["vel_loss"] * mean_velocity_error(poses, y, axis=2, squared=sq)
    if train_cfg["smooth_reg"] > 0:
        terms["sreg"] = train_cfg["smooth_reg"] * smoothness_regularization(poses, w, axis=2)
    total = sum(terms.values())
    return total, terms


def manifold_training_loss(pred, y, train_cfg=DEFAULT_TRAIN_CFG):
    """make_loss for single-hypothesis models (time axis 1), main_h36m_lifting.py:113-127,152-169."""
    w = torch.tensor(STANDARD_H36M_WEIGHTS, dtype=pred.dtype) if train_cfg["w_loss"] else None
    sq = bool(train_cfg.get("sq_loss", False))
    terms = {"wloss": weighted_mse_loss(pred, y, w) if sq else weighted_mpjpe_loss(pred, y, w)}
    if train_cfg["vel_loss"] > 0:
        terms["vloss"] = train_cfg["vel_loss"] * mean_velocity_error(pred, y, axis=1, squared=sq)
    if train_cfg["smooth_reg"] > 0:
        terms["sreg"] = train_cfg["smooth_reg"] * smoothness_regularization(pred, w, axis=1)
    if train_cfg.get("rigid_seg_reg", 0) > 0:                 # main_h36m_lifting.py:170-177
        terms["rigid_seg_reg"] = train_cfg["rigid_seg_reg"] * segments_time_consistency(pred.permute(0, 3, 2, 1), "sum")
    return sum(terms.values()), terms


# ---------------------------------------------------------------------------
# Eval aggregation + parity metric (rmcl_manifold_mix_ste.py:108-185,
# metrics/mean_joint_errors.py:8-36)
# ---------------------------------------------------------------------------
def mpjpe_error(pred: Tensor, gt: Tensor, mode: str = "average") -> Tensor:
    """mean_joint_errors.py:31-36."""
    d = torch.norm(gt.reshape(-1, 3) - pred.reshape(-1, 3), 2, 1)
    return d.mean() if mode == "average" else d.sum()


def poses_from_hyp_idx(hyp: Tensor, idx: Tensor) -> Tensor:
    """rmcl_manifold_mix_ste.py:121-139. hyp (B,H,L,J,3), idx (B,L) -> (B,L,J,3)."""
    B, H, L, J, _ = hyp.shape
    g = idx[:, None, :, None, None].expand(B, 1, L, J, 3)
    return hyp.gather(1, g)[:, 0]


def aggregate(hyp, scores=None, mode="weighted_ave", ground_truth=None):
    """RMCLManifoldMixSTE.aggregate, rmcl_manifold_mix_ste.py:141-185."""
    if mode == "best_score":
        return poses_from_hyp_idx(hyp, torch.argmax(scores, dim=1)[..., 0])
    if mode == "weighted_ave":
        return torch.sum(hyp * scores.unsqueeze(-1), dim=1)
    if mode == "oracle":
        e, idx = wta_l2_loss_and_activate_head(hyp, ground_truth, None)
        return e, poses_from_hyp_idx(hyp, idx)
    raise ValueError(f"Only best_score and weighted_ave modes are implemented.Got {mode}.")


# ---------------------------------------------------------------------------
# Input pipeline: PoseSequenceGenerator with miss_type "no_miss" (data/generators.py:44-219) and the
# PoseFlip transform (augmentations/transforms.py:7-28, functional.py:7-31).  Draws from the torch CPU RNG exactly like the
# reference: per item one torch.randint (random start), then one torch.rand (flip decision).
# ---------------------------------------------------------------------------
def window_tables(lengths, seq_len: int, drop_last: bool):
    """generators.py:87-104 -> (_map_index_to_pose, _map_index_to_frame)."""
    to_pose, to_frame = [], []
    for i, n in enumerate(lengths):
        size = n // seq_len
        if not drop_last and n % seq_len > 0:
            size += 1
        to_pose += [i] * size
        to_frame += [k * seq_len for k in range(size)]
    return to_pose, to_frame


def flip_pose(pose: Tensor, joints_left=H36M_JOINTS_LEFT, joints_right=H36M_JOINTS_RIGHT) -> Tensor:
    """functional.py:7-31 (out of place): horizontal coordinate negated, left and right joints swapped."""
    out = pose.clone()
    out[..., 0] *= -1
    l, r = list(joints_left), list(joints_right)
    out[..., l + r, :] = out[..., r + l, :].clone()
    return out


MISS_RATES = {"no_miss": 0.2, "random": 0.2, "random_left_arm_right_leg": 0.4, "structured_joint": 0.4, "structured_frame": 0.2}


def occlusion_tables(seq_len: int, J: int, miss_type: str, miss_rate: float = 0.2, noise_sigma: float = 5):
    """generators.py:160-216: (mask (L,J), noise (L,J,2) or None) drawn from numpy's global RNG with the reference's calls."""
    import math
    import numpy as np
    shape = (seq_len, J)
    if miss_type == "all":
        miss_type = np.random.choice(list(MISS_RATES.keys()))
        miss_rate = MISS_RATES[miss_type]
    mask, noise = np.ones(shape), None
    if miss_type == "random":
        mask = np.zeros(shape)
        mask[np.random.uniform(0.0, 1.0, size=shape) > miss_rate] = 1.0
    elif miss_type == "random_left_arm_right_leg":
        rand = np.random.choice(seq_len, size=math.floor(miss_rate * seq_len), replace=False).tolist()
        for i in [1, 2, 3, 11, 12, 13]:
            mask[rand, i] = 0.0
    elif miss_type == "structured_joint":
        occl = int(seq_len * miss_rate)
        r = np.random.choice(seq_len - occl, size=1, replace=False)
        mask[r[0]: r[0] + occl, [1, 2, 3]] = 0.0
    elif miss_type == "structured_frame":
        occl = int(seq_len * miss_rate)
        r = np.random.choice(seq_len - occl, size=1, replace=False)
        mask[r[0]: r[0] + occl] = 0.0
    elif miss_type == "noisy":
        noise = np.random.normal(0, noise_sigma, size=(seq_len, J, 2))
    elif miss_type != "no_miss":
        raise ValueError(f"Unexpected miss_type: {miss_type}")
    return mask, noise


def sequence_window(poses_3d, poses_2d, index: int, seq_len: int, random_start: bool, drop_last: bool, flip_probability=None,
                    miss_type: str = "no_miss", miss_rate: float = 0.2, noise_sigma: float = 5):
    """generators.py:106-219: (pose_2d (L,J,2) after occlusion, pose_3d (L,J,3)) of dataset item `index`."""
    to_pose, to_frame = window_tables([p.shape[0] for p in poses_3d], seq_len, drop_last)
    p3 = torch.as_tensor(poses_3d[to_pose[index]]).float()
    p2 = torch.as_tensor(poses_2d[to_pose[index]]).float()
    if random_start:
        start = torch.randint(low=0, high=p3.shape[0] - seq_len, size=(1,)).item()
    else:
        start = to_frame[index]
    idx = torch.clamp(torch.arange(start, start + seq_len), max=p3.shape[0] - 1)       # replicate padding of a short last window
    w2, w3 = p2[idx], p3[idx]
    if flip_probability is not None and torch.rand(1).item() <= flip_probability:
        w2, w3 = flip_pose(w2), flip_pose(w3)
    mask, noise = occlusion_tables(seq_len, w2.shape[1], miss_type, miss_rate, noise_sigma)
    if noise is not None:
        w2 = w2.double() + torch.from_numpy(noise)         # `float32 tensor += float64 ndarray` yields a float64 tensor in the reference
    return w2 * torch.from_numpy(mask[..., None]).float(), w3


# ---------------------------------------------------------------------------
# Dataset ingest (on-disk formats): what the reference's loaders make of the raw npz dictionaries before
# PoseSequenceGenerator sees them.  numpy float32 like the reference; `calib` is the Human3.6M camera calibration as
# {"intrinsic": [4 dicts], "extrinsic": {subject: [4 dicts]}} (data/h36m_lifting.py:122-584).
# ---------------------------------------------------------------------------
import numpy as np  # noqa: E402

H36M_KEPT_JOINTS_17 = tuple(j for j in range(32) if j not in (4, 5, 9, 10, 11, 16, 20, 21, 22, 23, 24, 28, 29, 30, 31))  # h36m_lifting.py:651-653
MAP_H36M_TO_MPI_JOINTS = (14, 8, 9, 10, 11, 12, 13, 15, 1, 16, 0, 5, 6, 7, 2, 3, 4)                                     # dataset_3dhp.py:55-73


def normalize_screen_coordinates(X, w, h):
    """camera.py:9-14 ([0, w] -> [-1, 1], aspect ratio kept).  The list operand makes the subtraction a float64 one."""
    return X / w * 2 - [1, h / w]


def world_to_camera(X, R, t):
    """camera.py:24-28 with quaternion.py:6-31: rotate X - t by the conjugate of the unit quaternion R (w first)."""
    q = np.asarray(R, dtype=np.float32)
    w, qv = q[0], -q[1:]
    v = (X - np.asarray(t, dtype=np.float32)).astype(np.float32)
    uv = np.cross(np.broadcast_to(qv, v.shape), v).astype(np.float32)
    uuv = np.cross(np.broadcast_to(qv, v.shape), uv).astype(np.float32)
    return (v + 2 * (w * uv + uuv)).astype(np.float32)


def h36m_cameras(calib):
    """h36m_lifting.py:591-618: per subject, the four cameras with normalised centre / focal length, translation in metres and the
    9-vector of intrinsics."""
    out = {}
    for subject, cams in calib["extrinsic"].items():
        out[subject] = []
        for i, ext in enumerate(cams):
            intr = calib["intrinsic"][i]
            cam = {"res_w": intr["res_w"], "res_h": intr["res_h"]}
            center = normalize_screen_coordinates(np.array(intr["center"], dtype="float32"), w=intr["res_w"], h=intr["res_h"]).astype("float32")
            focal = np.array(intr["focal_length"], dtype="float32") / intr["res_w"] * 2.0
            cam["intrinsic"] = np.concatenate((focal, center, np.array(intr["radial_distortion"], dtype="float32"),
                                               np.array(intr["tangential_distortion"], dtype="float32")))
            if "orientation" in ext:
                cam["orientation"] = np.array(ext["orientation"], dtype="float32")
                cam["translation"] = np.array(ext["translation"], dtype="float32") / 1000
            out[subject].append(cam)
    return out


def h36m_sequences(positions_3d, positions_2d, calib, subjects, action_filter=None, stride=1):
    """Human36mDataset (h36m_lifting.py:587-660, 17 joints) -> read_3d_data (utils.py:29-58) -> create_2d_data (utils.py:9-26) ->
    fetch (utils.py:61-127): one (3-D, 2-D, action, camera vector) entry per subject x action x camera."""
    cams = h36m_cameras(calib)
    p3, p2, actions, cam_out = [], [], [], []
    for subject in subjects:
        for action in positions_2d[subject].keys():
            if action_filter is not None and not any(action.lower().split(" ")[0] == a for a in action_filter):
                continue
            world = np.asarray(positions_3d[subject][action])[:, list(H36M_KEPT_JOINTS_17)]
            for i, kps in enumerate(positions_2d[subject][action]):
                cam = cams[subject][i]
                k = np.array(kps, dtype=np.float32, copy=True)
                k[..., :2] = normalize_screen_coordinates(k[..., :2], w=cam["res_w"], h=cam["res_h"])
                p2.append(k[::stride])
                actions.append(action.split(" ")[0])
                cam_out.append(np.concatenate([cam["intrinsic"], cam["orientation"], cam["translation"], np.array([i])]))
            for cam in cams[subject]:
                c = world_to_camera(world, cam["orientation"], cam["translation"])
                c = c - c[:, :1]
                p3.append(c[::stride])
    return p3, p2, actions, cam_out


def hp3d_sequences(data, train: bool):
    """Dataset3DHP.prepare_data (dataset_3dhp.py:146-229): root-relative (joint 14), H36M joint order, metres; 2-D normalised to the
    2048^2 (TS5 / TS6: 1920x1080) frame; test sequences keep their valid frames only."""
    p3, p2 = [], []
    m = list(MAP_H36M_TO_MPI_JOINTS)
    if train:
        for seq in data.keys():
            for cam in data[seq][0].keys():
                d3 = np.array(data[seq][0][cam]["data_3d"], copy=True)
                d3 = d3 - d3[:, 14:15]
                p3.append(d3[:, m] / 1000)
                d2 = np.array(data[seq][0][cam]["data_2d"], copy=True)
                d2[..., :2] = normalize_screen_coordinates(d2[..., :2], w=2048, h=2048)
                p2.append(d2[:, m])
        return p3, p2
    for seq in data.keys():
        valid = np.asarray(data[seq]["valid"]).astype(bool)
        d3 = np.array(data[seq]["data_3d"], copy=True)
        d3 = d3 - d3[:, 14:15]
        p3.append(d3[valid][:, m] / 1000)
        w, h = (1920, 1080) if seq in ("TS5", "TS6") else (2048, 2048)
        d2 = np.array(data[seq]["data_2d"], copy=True)
        d2[..., :2] = normalize_screen_coordinates(d2[..., :2], w=w, h=h)
        p2.append(d2[valid][:, m])
    return p3, p2


# ---------------------------------------------------------------------------
# Evaluation analytics: skeleton-consistency metrics (metrics/regularizations.py:8-157,
# metrics/utils.py:4-20), the remaining error metrics (metrics/mean_joint_errors.py:39-130), evaluation velocity
# error (metrics/losses.py:75-101) and 3DPCK / AUC (metrics/pck.py:92-199).  joints_coords is (B, 3, J, L).
# ---------------------------------------------------------------------------
H36M_BONES = tuple((j, p) for j, p in enumerate(H36M_PARENTS) if p >= 0)          # data/skeleton.py:101-103
H36M_BONES_LEFT = tuple(j - 1 for j in H36M_JOINTS_LEFT)                          # data/skeleton.py:110-120 (bone k <-> joint k+1)
H36M_BONES_RIGHT = tuple(j - 1 for j in H36M_JOINTS_RIGHT)


def measure_bones_length(joints_coords: Tensor, bones=H36M_BONES) -> Tensor:
    """metrics/utils.py:4-20 -> (B, num_bones, L)."""
    return torch.stack([(joints_coords[:, :, j, :] - joints_coords[:, :, p, :]).pow(2).sum(1).sqrt() for j, p in bones], dim=1)


def segments_time_stat(joints_coords: Tensor, mode: str) -> Tensor:
    """regularizations.py:8-35: unbiased variance (std for mode 'std') over time of every bone length -> (B, num_bones)."""
    bl = measure_bones_length(joints_coords)
    return torch.std(bl, dim=2) if mode == "std" else torch.var(bl, dim=2)


def segments_time_consistency(joints_coords: Tensor, mode: str) -> Tensor:
    """regularizations.py:38-50."""
    agg = {"average": torch.mean, "sum": torch.sum, "std": torch.mean, "min": torch.min, "max": torch.max}[mode]
    return agg(segments_time_stat(joints_coords, mode))


def segments_time_consistency_per_bone(joints_coords: Tensor, mode: str) -> Tensor:
    """regularizations.py:53-64 (modes with a tensor result)."""
    agg = {"average": torch.mean, "sum": torch.sum, "std": torch.mean}[mode]
    return agg(segments_time_stat(joints_coords, mode), dim=0)


def sagittal_diff(joints_coords: Tensor, squared: bool) -> Tensor:
    """regularizations.py:96-122 -> (B, num_pairs, L)."""
    bl = measure_bones_length(joints_coords)
    d = (bl[:, list(H36M_BONES_LEFT), :] - bl[:, list(H36M_BONES_RIGHT), :]).abs()
    return d ** 2.0 if squared else d


def sagittal_symmetry(joints_coords: Tensor, mode: str, squared: bool = True) -> Tensor:
    """regularizations.py:125-138."""
    d = sagittal_diff(joints_coords, squared)
    return d.mean() if mode == "average" else d.sum()


def sagittal_symmetry_per_bone(joints_coords: Tensor, mode: str, squared: bool = True) -> Tensor:
    """regularizations.py:141-157."""
    d = sagittal_diff(joints_coords, squared).permute(0, 2, 1).reshape(-1, len(H36M_BONES_LEFT))
    return d.mean(0) if mode == "average" else d.sum(0)


def segments_len_err(batch_imp: Tensor, batch_gt: Tensor, mode: str, signed: bool = True) -> Tensor:
    """mean_joint_errors.py:83-130: gt - predicted bone lengths over (frame, bone)."""
    diff = measure_bones_length(batch_gt) - measure_bones_length(batch_imp)
    if not signed:
        diff = diff.abs()
    return diff.mean() if mode == "average" else diff.sum()


def mse_error(pred: Tensor, gt: Tensor, mode: str = "average") -> Tensor:
    """mean_joint_errors.py:39-44."""
    d = (gt.reshape(-1, 3) - pred.reshape(-1, 3)).pow(2).sum(1)
    return d.mean() if mode == "average" else d.sum()


def jointwise_error(pred: Tensor, gt: Tensor, mode: str = "average", squared: bool = False) -> Tensor:
    """mean_joint_errors.py:47-80 (jointwise_error / jointwise_mse) -> (J,)."""
    J = gt.shape[-2]
    d = gt.reshape(-1, J, 3) - pred.reshape(-1, J, 3)
    e = d.pow(2).sum(2) if squared else torch.norm(d, 2, 2)
    return e.mean(0) if mode == "average" else e.sum(0)


def eval_velocity_error(pred: Tensor, target: Tensor, axis: int = 1, squared: bool = False) -> Tensor:
    """metrics/losses.py:75-101 for equal shapes (the evaluation call, main_h36m_lifting.py:968-973)."""
    dv = torch.diff(pred, dim=axis) - torch.diff(target, dim=axis)
    return (dv ** 2).mean() if squared else torch.norm(dv, dim=pred.dim() - 1).mean()


def procrustes_align(pred: Tensor, gt: Tensor) -> Tensor:
    """Similarity transform of p_mpjpe (mean_joint_errors.py:148-189) == compute_similarity_transform (pck.py:5-60), batched SVD in
    float64: pred, gt (N, J, 3) -> pred aligned onto gt."""
    X, Y = gt.double(), pred.double()
    muX, muY = X.mean(1, keepdim=True), Y.mean(1, keepdim=True)
    X0, Y0 = X - muX, Y - muY
    nX, nY = X0.pow(2).sum((1, 2), keepdim=True).sqrt(), Y0.pow(2).sum((1, 2), keepdim=True).sqrt()
    X0, Y0 = X0 / nX, Y0 / nY
    H = X0.transpose(1, 2) @ Y0
    U, s, Vt = torch.linalg.svd(H)
    V = Vt.transpose(1, 2)
    R = V @ U.transpose(1, 2)
    sign = torch.sign(torch.linalg.det(R))
    V = V.clone(); s = s.clone()
    V[:, :, -1] *= sign[:, None]
    s[:, -1] *= sign
    R = V @ U.transpose(1, 2)
    a = s.sum(1)[:, None, None] * nX / nY
    t = muX - a * (muY @ R)
    return a * (Y @ R) + t


def p_mpjpe(pred: Tensor, gt: Tensor) -> Tensor:
    """mean_joint_errors.py:148-189."""
    J = gt.shape[-2]
    al = procrustes_align(pred.reshape(-1, J, 3), gt.reshape(-1, J, 3))
    return torch.norm(al - gt.reshape(-1, J, 3).double(), dim=-1).mean()


def keypoint_3d_pck_auc(pred: Tensor, gt: Tensor, mask: Optional[Tensor] = None, alignment: str = "none", threshold: float = 150.0):
    """metrics/pck.py:92-199: (pck, auc) in percent; pred / gt (N, K, 3); alignment 'none', 'scale' or 'procrustes'."""
    pred, gt = pred.double(), gt.double()
    if alignment == "procrustes":
        pred = procrustes_align(pred, gt)
    if alignment == "scale":
        f = (pred * gt).sum((1, 2)) / (pred * pred).sum((1, 2))
        pred = pred * f[:, None, None]
    err = torch.norm(pred - gt, dim=-1)
    m = torch.ones_like(err, dtype=torch.bool) if mask is None else mask.bool()
    pck = (err < threshold)[m].double().mean() * 100
    ths = torch.linspace(0.0, 150.0, 31, dtype=torch.float64)
    auc = torch.stack([(err < t)[m].double().mean() for t in ths]).mean() * 100
    return pck, auc


# ---------------------------------------------------------------------------
# Optimizer (torch.optim.Adam with L2 weight decay, main_h36m_lifting.py:234-238)
# ---------------------------------------------------------------------------
def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr=4e-5, beta1=0.9,
              beta2=0.999, eps=1e-8, weight_decay=1e-6):
    """One torch.optim.Adam (non-AMSGrad, L2 decay) update, restated; returns (p, m, v)."""
    g = g + weight_decay * p
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    return p - (lr / bc1) * (m / denom), m, v


# ---------------------------------------------------------------------------
# Parameter construction (shapes/keys of SURVEY.md section 8b) and synthetic data
# 
# ---------------------------------------------------------------------------
def _linear_init(out_f, in_f, gen):
    """nn.Linear default init (kaiming_uniform(a=sqrt(5)) -> U(-1/sqrt(in), 1/sqrt(in)) for both)."""
    bound = 1.0 / math.sqrt(in_f)
    w = (torch.rand(out_f, in_f, generator=gen) * 2 - 1) * bound
    b = (torch.rand(out_f, generator=gen) * 2 - 1) * bound
    return w, b


def _mixste_state(st, pre, gen, T, J, C, depth, in_chans, embed=True, pos_std=0.02):
    if embed:
        st[pre + "Spatial_patch_to_embedding.weight"], st[pre + "Spatial_patch_to_embedding.bias"] = \
            _linear_init(C, in_chans, gen)
    st[pre + "Spatial_pos_embed"] = pos_std * torch.randn(1, J, C, generator=gen)
    st[pre + "Temporal_pos_embed"] = pos_std * torch.randn(1, T, C, generator=gen)
    for kind in ("STEblocks", "TTEblocks"):
        for i in range(depth):
            bp = f"{pre}{kind}.{i}."
            for n in ("norm1", "norm2"):
                st[bp + n + ".weight"] = 1 + 0.1 * torch.randn(C, generator=gen)
                st[bp + n + ".bias"] = 0.1 * torch.randn(C, generator=gen)
            st[bp + "attn.qkv.weight"], st[bp + "attn.qkv.bias"] = _linear_init(3 * C, C, gen)
            st[bp + "attn.proj.weight"], st[bp + "attn.proj.bias"] = _linear_init(C, C, gen)
            st[bp + "mlp.fc1.weight"], st[bp + "mlp.fc1.bias"] = _linear_init(2 * C, C, gen)
            st[bp + "mlp.fc2.weight"], st[bp + "mlp.fc2.bias"] = _linear_init(C, 2 * C, gen)
    for n in ("Spatial_norm", "Temporal_norm"):
        st[pre + n + ".weight"] = 1 + 0.1 * torch.randn(C, generator=gen)
        st[pre + n + ".bias"] = 0.1 * torch.randn(C, generator=gen)


def make_state(cfg: dict, seed: int = 0) -> Dict[str, Tensor]:
    """Random weights with the reference's key layout; LN affine and pos-embeds are perturbed
    away from their (1, 0, 0) defaults so that every parameter is exercised by parity tests.

    cfg: T, J, num_bones, C_rot, depth_rot, heads_rot, C_seg, depth_seg, heads_seg, n_hyp [, rot_dim = 6]
    (n_hyp == 0 -> ManifoldMixSTE layout with a plain MixSTE head of out_dim rot_dim).
    """
    D = cfg.get("rot_dim", 6)
    gen = torch.Generator().manual_seed(seed)
    st: Dict[str, Tensor] = {}
    T, J, S = cfg["T"], cfg["J"], cfg["num_bones"]
    C, Cs = cfg["C_rot"], cfg["C_seg"]
    rp, sp = "rotations_module.", "segments_module."
    _mixste_state(st, rp, gen, T, J, C, cfg["depth_rot"], 2)
    if cfg["n_hyp"] > 0:
        for k in range(cfg["n_hyp"]):
            hp = f"{rp}head.{k}."
            st[hp + "norm.weight"] = 1 + 0.1 * torch.randn(C, generator=gen)
            st[hp + "norm.bias"] = 0.1 * torch.randn(C, generator=gen)
            st[hp + "prediction_head.weight"], st[hp + "prediction_head.bias"] = _linear_init(D + 1, C, gen)
            st[hp + "score_head.weight"], st[hp + "score_head.bias"] = _linear_init(1, J, gen)
    else:
        st[rp + "head.0.weight"] = 1 + 0.1 * torch.randn(C, generator=gen)
        st[rp + "head.0.bias"] = 0.1 * torch.randn(C, generator=gen)
        st[rp + "head.1.weight"], st[rp + "head.1.bias"] = _linear_init(D, C, gen)
    _mixste_state(st, sp, gen, T, S, Cs, cfg["depth_seg"], 2, embed=False)
    st[sp + "head.0.weight"] = 1 + 0.1 * torch.randn(Cs, generator=gen)
    st[sp + "head.0.bias"] = 0.1 * torch.randn(Cs, generator=gen)
    st[sp + "head.1.weight"], st[sp + "head.1.bias"] = _linear_init(1, Cs, gen)
    st[sp + "joints_to_segments_proj.weight"], st[sp + "joints_to_segments_proj.bias"] = \
        _linear_init(S * Cs, J * 2, gen)
    return st


def oracle_cfg(cfg: dict) -> dict:
    return dict(depth_rot=cfg["depth_rot"], heads_rot=cfg["heads_rot"], depth_seg=cfg["depth_seg"],
                heads_seg=cfg["heads_seg"], n_hyp=cfg["n_hyp"], num_bones=cfg["num_bones"])


def synthetic_batch(B: int, T: int, J: int = 17, seed: int = 42):
    """X = clamp(0.3 randn, -1, 1) screen coords; y = 0.3 randn metres, root 0."""
    gen = torch.Generator().manual_seed(seed)
    X = (0.3 * torch.randn(B, T, J, 2, generator=gen)).clamp(-1, 1)
    y = 0.3 * torch.randn(B, T, J, 3, generator=gen)
    y[:, :, 0] = 0
    return X, y


FULL_CFG = dict(T=243, J=17, num_bones=16, C_rot=512, depth_rot=8, heads_rot=8,
                C_seg=128, depth_seg=2, heads_seg=8, n_hyp=5)
