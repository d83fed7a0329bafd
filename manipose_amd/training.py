"""Native training step of the lifting path: engine forward -> fused WTA loss (+grad) -> engine backward ->
gradient all-reduce (RCCL, one flat buffer) -> fused Adam.  No autograd graph, no host synchronisation inside a step.

Counterpart of the hot loop of ``train()`` in the reference (hpe/main_h36m_lifting.py:294-311); loss assembly as
``make_loss`` / ``compute_and_acc_loss`` (:101-209) with the defaults of hpe/conf/config.yaml:32-38.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
import torch.distributed as dist

from . import _lib
from .distributed import allreduce_gradients, allreduce_gradients_bucketed
from .optim import FusedAdam


class LiftingTrainer:
    def __init__(self, model, lr: float = 4e-5, weight_decay: float = 1e-6, w_loss: bool = True, vel_loss: float = 2.0,
                 smooth_reg: float = 0.5, rmcl_score_reg: float = 0.1, seed: int = 42, process_group=None,
                 sq_loss: bool = False, rigid_seg_reg: float = 0.0, grad_buckets: bool = False, health_interval: int = 25,
                 on_saturation: str = "raise"):
        """health_interval / on_saturation: a model with f16_backward carries some gradient operands as fp16 of S x value; stores that hit
        the +-65504 clamp (or met a non-finite value, written as 0) are counted on the device (mp_model_grad_health).  The trainer sums the
        two counters of every backward on the device (no synchronisation) and reads the sums every `health_interval` steps: a non-zero sum
        raises RuntimeError ("raise": the steps since the last check trained on clamped operands - they HAVE been applied; health_interval=1
        stops on the first) or warns ("warn").  With more than one rank the sums are all-reduced first, so every rank decides alike."""
        self.model = model
        self.lib = _lib.load()
        self.opt = FusedAdam(model, lr=lr, weight_decay=weight_decay)
        self.loss_cfg = _lib.LossConfig(rmcl_score_reg=rmcl_score_reg, vel_loss=vel_loss, smooth_reg=smooth_reg,
                                        w_loss=int(w_loss), sq_loss=int(sq_loss))
        self.rigid_seg_reg = float(rigid_seg_reg)      # main_h36m_lifting.py:170-177; single-hypothesis models only (4th loss term)
        if self.rigid_seg_reg > 0 and model._arch == "rmcl_manifold":
            raise NotImplementedError("train.rigid_seg_reg > 0 with the multi-hypothesis model: the reference's term permutes a 4-D "
                                      "(B, L, J, 3) prediction and fails on (B, H, L, J, 3) hypotheses")
        self.step_no = 0
        # grad_buckets: overlap the gradient exchange with the backward, one all-reduce per layer of the rotations net (~25 MB) on a
        # communication stream as the backward finishes it, instead of one 138 MB all-reduce behind it.  Off by default: the single
        # collective is ~1 ms of a ~175 ms step, and no multi-GPU box was available to measure whether RCCL kernels resident during the
        # backward cost the persistent GEMMs more than the overlap saves.
        self.grad_buckets = bool(grad_buckets)
        self._comm_stream = None
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        # every rank draws its own DropPath masks (the reference draws independent per-sample masks over the whole batch)
        self.seed = seed + (dist.get_rank(process_group) if self.world > 1 else 0)
        # time_exchange: record device events around the backward and the gradient exchange of every step (bench.py at N > 1:
        # "how long is the backward with / without collective kernels resident, how much of the exchange is exposed"); read with exchange_times()
        self.time_exchange = False
        self._xev = []
        self.flat_grads: Optional[torch.Tensor] = None
        if on_saturation not in ("raise", "warn"):
            raise ValueError("on_saturation must be 'raise' or 'warn'")
        self.health_interval, self.on_saturation = max(1, int(health_interval)), on_saturation
        self._health = self._health_sum = None            # device: this backward's four health values; running sum of (clamped, non-finite)
        self.saturation_events = 0                        # gradient elements found clamped / non-finite so far ("warn" mode keeps counting)
        self.rmcl = model._arch == "rmcl_manifold"
        self._bufs = {}

    def _buffers(self, B: int, T: int, device):
        key = (B, T)
        if key not in self._bufs:
            K = self.model._engine.K
            self._bufs = {key: dict(
                d_poses=torch.empty(B, K, T, 17, 3, device=device),
                d_scores=torch.empty(B, K, T, 1, device=device) if self.rmcl else None,
                terms=torch.zeros(4, device=device),
                scratch=torch.empty(4 * ((B * T + 47) // 48) + 8, device=device),      # (>= 4 ceil(B T / 48): the loss kernel's short-lived workgroups)
                rigid=torch.empty(B, device=device))}
        return self._bufs[key]

    def train_step(self, X: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        """One optimisation step on a per-rank batch; returns the device tensor of loss terms
        (wloss, score_reg, vloss, sreg) [3 terms for the single-hypothesis model]; nothing is synchronised."""
        m = self.model
        B, T = X.shape[0], X.shape[1]
        m._ensure_engine(B, X.device)
        eng = m._engine
        X = X.contiguous().float()
        y = y.contiguous().float()
        self.step_no += 1
        train = m.training
        poses, scores = eng.forward(m._flat, X, train=train, seed=self.seed, step=self.step_no)
        bf = self._buffers(B, T, X.device)
        st = _lib.stream_ptr()
        if self.rmcl:
            _lib.check(self.lib.mp_wta_loss(_lib.ptr(poses), _lib.ptr(scores), _lib.ptr(y), C.byref(self.loss_cfg),
                                            _lib.ptr(bf["terms"]), None, _lib.ptr(bf["d_poses"]), _lib.ptr(bf["d_scores"]),
                                            B, eng.K, T, _lib.ptr(bf["scratch"]), bf["scratch"].numel(), st), "mp_wta_loss")
        else:
            _lib.check(self.lib.mp_single_loss(_lib.ptr(poses), _lib.ptr(y), C.byref(self.loss_cfg), _lib.ptr(bf["terms"]),
                                               _lib.ptr(bf["d_poses"]), B, T, _lib.ptr(bf["scratch"]), bf["scratch"].numel(),
                                               st), "mp_single_loss")
            if self.rigid_seg_reg > 0:
                _lib.check(self.lib.mp_rigid_segments_loss(_lib.ptr(poses), self.rigid_seg_reg, C.c_void_p(bf["terms"].data_ptr() + 12),
                                                           _lib.ptr(bf["d_poses"]), B, T, _lib.ptr(bf["rigid"]), B, st), "mp_rigid_segments_loss")
        if self.flat_grads is None or self.flat_grads.shape != m._flat.shape:
            self.flat_grads = torch.empty_like(m._flat)
        self.flat_grads.zero_()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if self.time_exchange else None
        if ev:
            ev[0].record()
        eng.backward(m._flat, self.flat_grads, bf["d_poses"], bf["d_scores"])
        if ev:
            ev[1].record()
        if getattr(m, "f16_backward", False):
            self._check_grad_health(eng, X.device)
        if self.world > 1:
            # one collective per step over the single flat gradient buffer (137.8 MB fp32 at full size); RCCL picks the
            # all-links algorithm over the xGMI mesh.  Averaging is folded into the Adam kernel (grad_scale).
            if self.grad_buckets:
                if self._comm_stream is None:
                    self._comm_stream = torch.cuda.Stream()
                allreduce_gradients_bucketed(self.flat_grads, eng, self._comm_stream, self.pg)
            else:
                allreduce_gradients(self.flat_grads, self.pg)
        if ev:
            ev[2].record()             # the current stream has the reduced gradients from here on
            self._xev.append(ev)
        self.opt.step(self.flat_grads, grad_scale=1.0 / self.world)
        return bf["terms"]

    def _check_grad_health(self, eng, device) -> None:
        """f16_backward models: add this backward's saturation counters to a device-side sum (asynchronous), look at the sum every
        health_interval steps (one small device-to-host copy then)."""
        if self._health is None:
            self._health = torch.zeros(4, device=device)
            self._health_sum = torch.zeros(2, device=device)
        eng.grad_health_async(self._health)
        self._health_sum += self._health[1:3]
        if self.step_no % self.health_interval == 0:
            if self.world > 1:
                # every rank must take the same decision BEFORE the gradient collective: a rank that raised alone would leave the others
                # waiting in the all-reduce until the RCCL timeout
                dist.all_reduce(self._health_sum, group=self.pg)
            clamped, bad = (int(v) for v in self._health_sum.tolist())
            self._health_sum.zero_()
            if clamped or bad:
                self.saturation_events += clamped + bad
                msg = (f"manipose_amd: the fp16 gradient operands of the last {self.health_interval} steps saturated ({clamped} elements clamped at "
                       f"+-65504, {bad} non-finite): those steps trained on clamped gradients; rebuild the model with f16_backward=False")
                if self.on_saturation == "raise":
                    raise RuntimeError(msg)
                import warnings
                warnings.warn(msg)

    def exchange_times(self):
        """Mean device times (ms) over the steps recorded since the last call (time_exchange): `backward_ms` = the backward on the caller's
        stream (with grad_buckets the collectives of finished layers run beside it), `exposed_exchange_ms` = what the caller's stream then
        still waits for before the optimizer step.  Synchronises."""
        if not self._xev:
            return None
        torch.cuda.synchronize()
        b = sum(e[0].elapsed_time(e[1]) for e in self._xev) / len(self._xev)
        x = sum(e[1].elapsed_time(e[2]) for e in self._xev) / len(self._xev)
        n = len(self._xev)
        self._xev = []
        return {"backward_ms": b, "exposed_exchange_ms": x, "steps": n}

    @torch.no_grad()
    def eval_loss(self, X: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        m = self.model
        B, T = X.shape[0], X.shape[1]
        m._ensure_engine(B, X.device)
        eng = m._engine
        poses, scores = eng.forward(m._flat, X.contiguous().float(), train=False, infer=True)
        bf = self._buffers(B, T, X.device)
        st = _lib.stream_ptr()
        y = y.contiguous().float()
        if self.rmcl:
            _lib.check(self.lib.mp_wta_loss(_lib.ptr(poses), _lib.ptr(scores), _lib.ptr(y), C.byref(self.loss_cfg),
                                            _lib.ptr(bf["terms"]), None, None, None, B, eng.K, T, _lib.ptr(bf["scratch"]),
                                            bf["scratch"].numel(), st), "mp_wta_loss")
        else:
            _lib.check(self.lib.mp_single_loss(_lib.ptr(poses), _lib.ptr(y), C.byref(self.loss_cfg), _lib.ptr(bf["terms"]),
                                               None, B, T, _lib.ptr(bf["scratch"]), bf["scratch"].numel(), st), "mp_single_loss")
            if self.rigid_seg_reg > 0:
                _lib.check(self.lib.mp_rigid_segments_loss(_lib.ptr(poses), self.rigid_seg_reg, C.c_void_p(bf["terms"].data_ptr() + 12),
                                                           None, B, T, _lib.ptr(bf["rigid"]), B, st), "mp_rigid_segments_loss")
        return bf["terms"].clone()
