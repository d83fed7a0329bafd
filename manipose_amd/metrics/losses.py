"""Training losses of the lifting path on the fused HIP loss kernel (csrc/wta_loss.hip).

Reference functions restated at their call signatures (hpe/mh_so3_hpe/metrics/losses.py, regularizations.py:160-174):
``wta_l2_loss_and_activate_head``, ``wta_with_scoring_loss``, ``mean_velocity_error``, ``smoothness_regularization``,
``weighted_mpjpe_loss``; plus the one-launch ``rmcl_training_loss`` / ``manifold_training_loss`` that evaluate the whole
default loss of ``make_loss`` (hpe/main_h36m_lifting.py:101-209) and its gradient in a single pass.
All inputs must live on a ROCm device; there is no CPU implementation here.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from .. import _lib

STANDARD_H36M_WEIGHTS = torch.Tensor([1, 1, 2.5, 2.5, 1, 2.5, 2.5, 1, 1, 1, 1.5, 1.5, 4, 4, 1.5, 4, 4])


def _uses_h36m_weights(weights):
    """mp_loss_config.w_loss for a `weights` argument of the reference's loss functions: 0 (None), 1 (STANDARD_H36M_WEIGHTS), or the
    tuple (2, 17 floats) for any other per-joint weights (passed to the kernel in mp_loss_config.joint_weights)."""
    if weights is None:
        return 0
    w = torch.as_tensor(weights, dtype=torch.float32).detach().cpu().reshape(-1)
    if w.numel() != 17:
        raise ValueError(f"expected 17 per-joint weights, got {w.numel()}")
    if torch.equal(w, STANDARD_H36M_WEIGHTS):
        return 1
    return (2, tuple(float(v) for v in w))


def _loss_config(beta, vel_w, smooth_w, use_w, squared):
    cfg = _lib.LossConfig(rmcl_score_reg=beta, vel_loss=vel_w, smooth_reg=smooth_w, w_loss=use_w[0] if isinstance(use_w, tuple) else int(use_w),
                          sq_loss=int(squared))
    if isinstance(use_w, tuple):
        for j, v in enumerate(use_w[1]):
            cfg.joint_weights[j] = v
    return cfg


def _scratch(B: int, T: int, device) -> torch.Tensor:
    return torch.empty(4 * ((B * T + 47) // 48) + 8, dtype=torch.float32, device=device)      # >= 4 ceil(B T / 48): the loss kernel's short-lived workgroups


class _FusedLoss(torch.autograd.Function):
    """terms = (wloss, score_reg, vloss, sreg) [rmcl] or (wloss, vloss, sreg) [single]; grads produced in the same launch."""

    @staticmethod
    def forward(ctx, poses, scores, y, beta, vel_w, smooth_w, use_w, squared=0):
        lib = _lib.load()
        poses = poses.contiguous().float()
        y = y.contiguous().float()
        cfg = _loss_config(beta, vel_w, smooth_w, use_w, squared)
        dev = poses.device
        need_grad = poses.requires_grad or (scores is not None and scores.requires_grad)
        if scores is not None:
            B, K, T = poses.shape[:3]
            scores = scores.contiguous().float()
            terms = torch.empty(4, dtype=torch.float32, device=dev)
            d_poses = torch.empty_like(poses) if need_grad else None
            d_scores = torch.empty_like(scores) if need_grad else None
            argmin = torch.empty(B, T, dtype=torch.int32, device=dev)
            sc = _scratch(B, T, dev)
            import ctypes as C
            _lib.check(lib.mp_wta_loss(_lib.ptr(poses), _lib.ptr(scores), _lib.ptr(y), C.byref(cfg), _lib.ptr(terms),
                                       _lib.ptr(argmin), _lib.ptr(d_poses), _lib.ptr(d_scores), B, K, T, _lib.ptr(sc),
                                       sc.numel(), _lib.stream_ptr()), "mp_wta_loss")
            ctx.mark_non_differentiable(argmin)
            ctx.save_for_backward(d_poses, d_scores)
            ctx.n_terms = 4
            return terms, argmin
        B, T = poses.shape[:2]
        terms = torch.empty(3, dtype=torch.float32, device=dev)
        d_poses = torch.empty_like(poses) if need_grad else None
        sc = _scratch(B, T, dev)
        import ctypes as C
        _lib.check(lib.mp_single_loss(_lib.ptr(poses), _lib.ptr(y), C.byref(cfg), _lib.ptr(terms), _lib.ptr(d_poses), B, T,
                                      _lib.ptr(sc), sc.numel(), _lib.stream_ptr()), "mp_single_loss")
        ctx.save_for_backward(d_poses, None)
        ctx.n_terms = 3
        return terms, torch.empty(0, dtype=torch.int32, device=dev)

    @staticmethod
    def backward(ctx, g_terms, _g_argmin):
        d_poses, d_scores = ctx.saved_tensors
        # the kernel's gradient is that of sum(terms); a caller weighting the terms differently is not supported
        if d_poses is None:
            return (None,) * 8
        # (no check of that here: it would force a host sync in the training step)
        g = g_terms
        gp = d_poses * g[0]
        gs = d_scores * g[0] if d_scores is not None else None
        return gp, gs, None, None, None, None, None, None


def rmcl_training_loss(poses, scores, y, w_loss: bool = True, vel_loss: float = 2.0, smooth_reg: float = 0.5,
                       rmcl_score_reg: float = 0.1, sq_loss: bool = False) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    """Whole default multi-hypothesis loss (make_loss with conf/config.yaml:31-38) in ONE kernel launch, through the registered custom
    operator ``torch.ops.manipose.wta_loss`` (manipose_amd/ops.py; value and gradient come out of the same launch)."""
    from .. import ops  # noqa: F401  (registers torch.ops.manipose.*)
    if not poses.is_cuda:
        raise RuntimeError("manipose_amd: HIP kernels need tensors on a ROCm device (got a CPU tensor); there is no CPU fallback")
    terms = torch.ops.manipose.wta_loss(poses, scores, y, float(rmcl_score_reg), float(vel_loss), float(smooth_reg), int(w_loss), int(sq_loss))[0]
    names = ("wloss", "score_reg", "vloss", "sreg")
    return terms.sum(), {n: terms[i] for i, n in enumerate(names)}


def manifold_training_loss(pred, y, w_loss: bool = True, vel_loss: float = 2.0, smooth_reg: float = 0.5, sq_loss: bool = False):
    terms, _ = _FusedLoss.apply(pred, None, y, 0.0, vel_loss, smooth_reg, int(w_loss), int(sq_loss))
    names = ("wloss", "vloss", "sreg")
    return terms.sum(), {n: terms[i] for i, n in enumerate(names)}


# ---- reference-named single-term functions (each one launch of the same kernel with the other weights at 0) ----
def wta_l2_loss_and_activate_head(hypothesis, y, weights=None, squared: bool = False):
    """losses.py:126-138 -> (per-frame WTA error (B, L), winner index (B, L)); not differentiable here (eval/oracle use);
    the differentiable WTA term is ``rmcl_training_loss``/``wta_with_scoring_loss``."""
    lib = _lib.load()
    B, K, T = hypothesis.shape[:3]
    hyp = hypothesis.detach().contiguous().float()
    gt = y.detach().contiguous().float()
    # per-frame values: winner poses through the aggregation kernel, then per-frame weighted error
    use_w = _uses_h36m_weights(weights)
    import ctypes as C
    cfg = _loss_config(0.0, 0.0, 0.0, use_w, squared)
    terms = torch.empty(4, dtype=torch.float32, device=hyp.device)
    argmin = torch.empty(B, T, dtype=torch.int32, device=hyp.device)
    dummy = torch.full((B, K, T, 1), 1.0 / K, dtype=torch.float32, device=hyp.device)
    sc = _scratch(B, T, hyp.device)
    _lib.check(lib.mp_wta_loss(_lib.ptr(hyp), _lib.ptr(dummy), _lib.ptr(gt), C.byref(cfg), _lib.ptr(terms), _lib.ptr(argmin),
                               None, None, B, K, T, _lib.ptr(sc), sc.numel(), _lib.stream_ptr()), "mp_wta_loss")
    idx = argmin.long()
    win = hyp.gather(1, idx[:, None, :, None, None].expand(B, 1, T, 17, 3))[:, 0]
    w = (torch.tensor(use_w[1]) if isinstance(use_w, tuple) else (STANDARD_H36M_WEIGHTS if use_w else torch.ones(17))).to(hyp.device)
    if squared:         # losses.py:110-116: mean over the coordinates, then over the joints
        vals = (w[:, None] * (win - gt) ** 2).mean(dim=-1).mean(dim=-1)
    else:
        vals = (w * (win - gt).norm(dim=-1)).mean(dim=-1)
    return vals, idx


def wta_with_scoring_loss(hypothesis, scores, y, beta: float, weights=None, squared: bool = False):
    """losses.py:141-170 -> (wta.mean() + beta*bce, beta*bce); beta == 0 returns the WTA mean alone."""
    terms, _ = _FusedLoss.apply(hypothesis, scores, y, float(beta), 0.0, 0.0, _uses_h36m_weights(weights), int(squared))
    if beta == 0:
        return terms[0]
    return terms[0] + terms[1], terms[1]


def mean_velocity_error(predicted, target, axis: int = 1, squared: bool = False):
    """losses.py:75-101 (time axis 2 for (B,H,L,J,3) predictions, 1 for (B,L,J,3))."""
    if predicted.dim() == 5:
        assert axis == 2, "time axis of (B,H,L,J,3) hypotheses is 2"
        dummy = torch.full(predicted.shape[:3] + (1,), 1.0 / predicted.shape[1], device=predicted.device)
        terms, _ = _FusedLoss.apply(predicted, dummy, target, 0.0, 1.0, 0.0, 0, int(squared))
        return terms[2]
    assert axis == 1 and predicted.shape == target.shape
    terms, _ = _FusedLoss.apply(predicted, None, target, 0.0, 1.0, 0.0, 0, int(squared))
    return terms[1]


def smoothness_regularization(prediction, weights=None, axis: int = 1):
    """regularizations.py:160-174."""
    use_w = _uses_h36m_weights(weights)
    zeros = torch.zeros(prediction.shape[:1] + prediction.shape[-3:], device=prediction.device) if prediction.dim() == 5 \
        else torch.zeros_like(prediction)
    if prediction.dim() == 5:
        assert axis == 2
        dummy = torch.full(prediction.shape[:3] + (1,), 1.0 / prediction.shape[1], device=prediction.device)
        terms, _ = _FusedLoss.apply(prediction, dummy, zeros, 0.0, 0.0, 1.0, use_w)
        return terms[3]
    assert axis == 1
    terms, _ = _FusedLoss.apply(prediction, None, zeros, 0.0, 0.0, 1.0, use_w)
    return terms[2]


def weighted_mse_loss(prediction, target, weights=None, dims=None):
    """losses.py:46-72 with dims=None on (B, L, J, 3) tensors (single-hypothesis wloss under train.sq_loss)."""
    if dims is not None:
        raise NotImplementedError("manipose_amd: dims != None is served by wta_l2_loss_and_activate_head(squared=True)")
    terms, _ = _FusedLoss.apply(prediction, None, target, 0.0, 0.0, 0.0, _uses_h36m_weights(weights), 1)
    return terms[0]


def weighted_mpjpe_loss(prediction, target, weights=None, dims=None):
    """losses.py:14-43 with dims=None on (B, L, J, 3) tensors (single-hypothesis wloss)."""
    if dims is not None:
        raise NotImplementedError("manipose_amd: dims != None is served by wta_l2_loss_and_activate_head")
    terms, _ = _FusedLoss.apply(prediction, None, target, 0.0, 0.0, 0.0, _uses_h36m_weights(weights))
    return terms[0]
