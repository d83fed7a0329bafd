"""One-pass evaluation analytics on the GPU (C ABI ``mp_pose_metrics``): the running sums behind the reference's
consistency / error metrics, computed by one HIP kernel over the frames.  The reference-named functions in
``regularizations.py``, ``mean_joint_errors.py`` and ``pck.py`` are thin views of this result.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import torch

from .. import _lib
from ..data.skeleton import assert_h36m

NB, NP, NJ, NS = 16, 6, 17, 12
_OB, _OP, _OJ = NS, NS + 4 * NB, NS + 4 * NB + 2 * NP


@dataclass
class PoseAnalytics:
    """Per-batch-item sums (see include/manipose_hip.h, mp_pose_metrics) plus the frame-0 bone lengths."""
    rows: torch.Tensor          # (B, row_floats)
    len0: torch.Tensor          # (B, 16)
    B: int
    L: int

    def scalar(self, i: int) -> torch.Tensor:
        return self.rows[:, i].double().sum()

    @property
    def per_bone(self) -> torch.Tensor:           # (B, 16, 4): sum d, sum d^2, sum |gt - pred|, sum (gt - pred)
        return self.rows[:, _OB:_OP].view(self.B, NB, 4)

    @property
    def per_pair(self) -> torch.Tensor:           # (B, 6, 2): sum |l - r|, sum (l - r)^2
        return self.rows[:, _OP:_OJ].view(self.B, NP, 2)

    @property
    def per_joint(self) -> torch.Tensor:          # (B, 17, 2): sum ||e||, sum ||e||^2
        return self.rows[:, _OJ:_OJ + 2 * NJ].view(self.B, NJ, 2)

    def bone_variance(self, unbiased: bool = True) -> torch.Tensor:
        """Variance over time of every bone length, (B, 16): from the shifted sums (exact up to rounding)."""
        s1, s2 = self.per_bone[..., 0].double(), self.per_bone[..., 1].double()
        n = float(self.L)
        var = (s2 - s1 * s1 / n) / (n - 1.0 if unbiased else n)
        return var.clamp_min(0.0).float()


def _dptr(t: Optional[torch.Tensor]):
    """Device address of a (possibly strided) float32 tensor: the kernel addresses elements through explicit strides."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("manipose_amd: HIP kernels need tensors on a ROCm device (got a CPU tensor); there is no CPU fallback")
    return C.c_void_p(t.data_ptr())


def _strides(t: torch.Tensor, order) -> "C.Array":
    st = t.stride()
    return (C.c_int64 * 4)(*[int(st[i]) for i in order])


def pose_analytics(pred: torch.Tensor, gt: Optional[torch.Tensor] = None, layout: str = "BLJC", mask: Optional[torch.Tensor] = None,
                   pred_scale: float = 1.0, gt_scale: float = 1.0, pck_threshold: float = 150.0, auc_max: float = 150.0,
                   auc_steps: int = 31, scale_align: bool = False, skeleton=None) -> PoseAnalytics:
    """``layout``: "BLJC" = (B, L, J, 3) or "BCJL" = the reference's permuted (B, 3, J, L); any strides, no copy."""
    if skeleton is not None:
        assert_h36m(skeleton)
    if pred.device.type != "cuda":
        raise RuntimeError("manipose_amd: pose analytics run on the ROCm device only (HIP kernel, no CPU fallback)")
    order = {"BLJC": (0, 1, 2, 3), "BCJL": (0, 3, 2, 1)}[layout]
    if pred.dim() != 4 or pred.dtype != torch.float32:
        raise AssertionError(f"expected a 4-D float32 tensor in layout {layout}, got {tuple(pred.shape)} {pred.dtype}")
    B, L, J, Cc = (pred.shape[i] for i in order)
    if Cc != 3 or J != NJ:
        raise AssertionError(f"expected 3 coordinates of {NJ} joints, got J={J}, C={Cc}")
    if gt is not None and (gt.shape != pred.shape or gt.dtype != torch.float32 or gt.device != pred.device):
        raise AssertionError("prediction and target must have the same shape, dtype and device")
    lib = _lib.load()
    nv = int(lib.mp_pose_metrics_row_floats())
    rows = torch.empty(B, nv, device=pred.device)
    len0 = torch.empty(B, NB, device=pred.device)
    chunks = (L + 127) // 128
    scratch = torch.empty(B * chunks * nv, device=pred.device)
    m = None
    if mask is not None:
        m = mask.to(device=pred.device, dtype=torch.uint8).contiguous()
        if m.numel() != B * L * J:
            raise AssertionError("mask must have B*L*J entries")
    with torch.cuda.device(pred.device):
        _lib.check(lib.mp_pose_metrics(_dptr(pred), _strides(pred, order), _dptr(gt), _strides(gt, order) if gt is not None else None,
                                       _lib.ptr(m), B, L, J, float(pred_scale), float(gt_scale), float(pck_threshold), float(auc_max),
                                       int(auc_steps), int(bool(scale_align)), _lib.ptr(rows), _lib.ptr(len0), _lib.ptr(scratch),
                                       scratch.numel(), _lib.stream_ptr()), "mp_pose_metrics")
    return PoseAnalytics(rows, len0, B, L)


class AnalyticsAccumulator:
    """Adds up ``pose_analytics`` results over evaluation batches and reports the reference's analytics table
    (main_h36m_lifting.py:933-990, main_3dhp.py:860-910): all frames of all windows form ONE sequence, as in the reference's
    ``(1, 3, J, B*L)`` reshape for the time consistency.  Only (B, 16)-sized float64 bookkeeping happens here."""

    def __init__(self):
        self.scal = None            # (12,) float64
        self.pairs = None           # (6, 2)
        self.joints = None          # (17, 2)
        self.bone_err = None        # (16, 2): sum |gt - pred|, sum (gt - pred)
        self.n = 0.0                # frames
        self.ref = None             # (16,) common shift of the variance sums
        self.s1 = None              # (16,) sum (len - ref)
        self.s2 = None              # (16,) sum (len - ref)^2
        self.proc = None            # (5,) Procrustes sums (procrustes_sums)

    def add(self, a: PoseAnalytics) -> None:
        rows = a.rows.double()
        pb = a.per_bone.double()
        len0 = a.len0.double()
        if self.scal is None:
            z = lambda *s: torch.zeros(*s, dtype=torch.float64, device=rows.device)
            self.scal, self.pairs, self.joints, self.bone_err = z(NS), z(NP, 2), z(NJ, 2), z(NB, 2)
            self.ref, self.s1, self.s2 = len0[0].clone(), z(NB), z(NB)
        self.scal += rows[:, :NS].sum(0)
        self.pairs += a.per_pair.double().sum(0)
        self.joints += a.per_joint.double().sum(0)
        self.bone_err += pb[..., 2:4].sum(0)
        # re-centre every item's shifted sums on the common reference: sum (x-c) = S1 + n (s-c), sum (x-c)^2 = S2 + 2 (s-c) S1 + n (s-c)^2
        d = len0 - self.ref[None]
        n = float(a.L)
        self.s1 += (pb[..., 0] + n * d).sum(0)
        self.s2 += (pb[..., 1] + 2.0 * d * pb[..., 0] + n * d * d).sum(0)
        self.n += n * a.B

    def add_procrustes(self, sums: torch.Tensor) -> None:
        self.proc = sums.clone() if self.proc is None else self.proc + sums

    def all_reduce(self, group=None) -> None:
        """Merge the accumulators of all ranks (evaluation sharded over windows, SURVEY 8e: "a final sum-reduce of (sum of per-joint
        error, count)"): one SUM all-reduce of the packed sums, one all-gather of the shifted variance sums, which are re-centred on
        rank 0's reference before they are added.  Every rank ends with the merged accumulator."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return
        if self.scal is None:
            raise RuntimeError("AnalyticsAccumulator.all_reduce: every rank must have added at least one batch")
        world = dist.get_world_size(group)
        has_proc = self.proc is not None
        n_t = torch.tensor([self.n], dtype=torch.float64, device=self.scal.device)
        parts = [self.scal, self.pairs.reshape(-1), self.joints.reshape(-1), self.bone_err.reshape(-1), n_t] + ([self.proc] if has_proc else [])
        flat = torch.cat([p.reshape(-1).double() for p in parts])
        var = torch.cat([self.ref, self.s1, self.s2, n_t])                       # (3 * 16 + 1,)
        gathered = [torch.empty_like(var) for _ in range(world)]
        dist.all_gather(gathered, var, group=group)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        o = 0
        for name, shape in (("scal", (NS,)), ("pairs", (NP, 2)), ("joints", (NJ, 2)), ("bone_err", (NB, 2))):
            k = 1
            for d in shape:
                k *= d
            setattr(self, name, flat[o:o + k].reshape(shape).clone())
            o += k
        self.n = float(flat[o].item()); o += 1
        if has_proc:
            self.proc = flat[o:o + 5].clone()
        ref0 = gathered[0][:NB]
        s1, s2 = torch.zeros_like(self.s1), torch.zeros_like(self.s2)
        for g in gathered:        # sum (x - c0) = S1 + n (c - c0);  sum (x - c0)^2 = S2 + 2 (c - c0) S1 + n (c - c0)^2
            c, a1, a2, n = g[:NB], g[NB:2 * NB], g[2 * NB:3 * NB], g[3 * NB]
            d = c - ref0
            s1 += a1 + n * d
            s2 += a2 + 2.0 * d * a1 + n * d * d
        self.ref, self.s1, self.s2 = ref0.clone(), s1, s2

    def report(self) -> dict:
        s, n = self.scal, self.n
        nj = n * NJ
        var = ((self.s2 - self.s1 * self.s1 / n) / (n - 1.0)).clamp_min(0.0)
        mpjpe = (s[0] / nj).item()
        mse = (s[1] / nj).item()
        out = {"mpjpe": mpjpe, "mse": mse, "err_var": mse - mpjpe ** 2,
               "mpsse": (s[2] / (n * NP)).item(),                      # sagittal_symmetry(mode="average", squared=False)
               "mpsce": var.sqrt().mean().item(),                       # segments_time_consistency(flattened, mode="std")
               "seg_len_err": (s[4] / (n * NB)).item(),                 # segments_len_err(mode="average", signed=False)
               "pck": (100.0 * s[6] / s[8]).item() if s[8] > 0 else float("nan"),
               "auc": (100.0 * s[7] / (31.0 * s[8])).item() if s[8] > 0 else float("nan"),
               "jointwise_err": (self.joints[:, 0] / n).tolist(),
               "mpsce_per_bone": var.sqrt().tolist(), "mpsse_per_pair": (self.pairs[:, 0] / n).tolist()}
        if self.proc is not None:                # Protocol #2 and the Procrustes-aligned 3DPCK / AUC
            out["p_mpjpe"] = (self.proc[0] / (self.proc[4] * NJ)).item()
            out["pck_procrustes"] = (100.0 * self.proc[1] / self.proc[3]).item()
            out["auc_procrustes"] = (100.0 * self.proc[2] / (31.0 * self.proc[3])).item()
        return out


def procrustes_sums(pred: torch.Tensor, gt: torch.Tensor, mask: Optional[torch.Tensor] = None, pred_scale: float = 1.0,
                    gt_scale: float = 1.0, pck_threshold: float = 150.0, auc_max: float = 150.0, auc_steps: int = 31) -> torch.Tensor:
    """(5,) float64: sum of Procrustes-aligned per-joint errors, PCK count, AUC count sum, visible joints, frames (mp_procrustes_errors).
    pred / gt: (..., 17, 3) on the device."""
    if pred.device.type != "cuda":
        raise RuntimeError("manipose_amd: Procrustes errors run on the ROCm device only (HIP kernel, no CPU fallback)")
    assert pred.shape == gt.shape and pred.shape[-1] == 3 and pred.shape[-2] == NJ
    p = pred.detach().float().reshape(-1, NJ, 3).contiguous()
    g = gt.detach().to(p.device).float().reshape(-1, NJ, 3).contiguous()
    n = p.shape[0]
    m = None
    if mask is not None:
        m = mask.to(device=p.device, dtype=torch.uint8).contiguous()
        assert m.numel() == n * NJ
    lib = _lib.load()
    out = torch.empty(5, device=p.device)
    scratch = torch.empty(5 * ((n + 255) // 256), device=p.device)
    with torch.cuda.device(p.device):
        _lib.check(lib.mp_procrustes_errors(_lib.ptr(p), _lib.ptr(g), _lib.ptr(m), n, NJ, float(pred_scale), float(gt_scale),
                                            float(pck_threshold), float(auc_max), int(auc_steps), _lib.ptr(out), _lib.ptr(scratch),
                                            scratch.numel(), _lib.stream_ptr()), "mp_procrustes_errors")
    return out.double()
