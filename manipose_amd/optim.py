"""Fused Adam over the model's flat parameter buffer (mp_adam_step; reference: torch.optim.Adam(lr=4e-5,
weight_decay=1e-6), hpe/main_h36m_lifting.py:234-238) and the data-parallel gradient exchange."""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib


class FusedAdam:
    """One kernel launch per step over all parameters.  ``state_dict``/``load_state_dict`` keep the moments so that
    checkpoints (params_{tag}.pth of the reference, main_h36m_lifting.py:75-98) can be resumed."""

    def __init__(self, model, lr: float = 4e-5, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-6):
        self.model = model
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.step_count = 0
        self.exp_avg: Optional[torch.Tensor] = None
        self.exp_avg_sq: Optional[torch.Tensor] = None
        self.param_groups = [{"lr": lr}]          # so torch lr schedulers' bookkeeping style code can read/modify lr
        self.lr_mult: Optional[torch.Tensor] = None      # MuAdam: per-element lr / weight-decay multipliers in the flat layout
        self.wd_mult: Optional[torch.Tensor] = None

    def set_multipliers(self, mults) -> None:
        """{state-dict key: (lr multiplier, weight-decay multiplier)} -> MuAdam-style training (mup_lite.mup_lr_multipliers)."""
        m = self.model
        flat = m.flat_parameters()
        names = {id(p): n for n, p in m.named_parameters()}
        self.lr_mult, self.wd_mult = torch.ones_like(flat), torch.ones_like(flat)
        for (off, n), p in zip(m._slots, m._plist):
            lm, wm = mults[names[id(p)]]
            self.lr_mult[off:off + n] = lm
            self.wd_mult[off:off + n] = wm

    def _flat_grad(self) -> torch.Tensor:
        m = self.model
        flat = m.flat_parameters()
        g = m._last_flat_grad
        ok = g is not None and all(p.grad is not None and p.grad.data_ptr() == g.data_ptr() + 4 * off
                                   for (off, _), p in zip(m._slots, m._plist))
        if ok:
            return g
        g = torch.zeros_like(flat)            # gradients were accumulated/replaced by the caller: gather them
        for (off, n), p in zip(m._slots, m._plist):
            if p.grad is not None:
                g[off:off + n].copy_(p.grad.reshape(-1))
        return g

    def step(self, flat_grad: Optional[torch.Tensor] = None, grad_scale: float = 1.0) -> None:
        lib = _lib.load()
        flat = self.model.flat_parameters()
        g = flat_grad if flat_grad is not None else self._flat_grad()
        if self.exp_avg is None:
            self.exp_avg = torch.zeros_like(flat)
            self.exp_avg_sq = torch.zeros_like(flat)
        self.step_count += 1
        self.model._fused_updates = getattr(self.model, "_fused_updates", 0) + 1      # _fused.py: the parameter bits change (the kernel writes through raw pointers)
        lr = self.param_groups[0]["lr"]
        if self.lr_mult is not None:
            if self.lr_mult.device != flat.device:
                self.lr_mult, self.wd_mult = self.lr_mult.to(flat.device), self.wd_mult.to(flat.device)
            _lib.check(lib.mp_adam_step_scaled(_lib.ptr(flat), _lib.ptr(g), _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq), flat.numel(),
                                               self.step_count, lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, grad_scale,
                                               _lib.ptr(self.lr_mult), _lib.ptr(self.wd_mult), _lib.stream_ptr()), "mp_adam_step_scaled")
            return
        _lib.check(lib.mp_adam_step(_lib.ptr(flat), _lib.ptr(g), _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq),
                                    flat.numel(), self.step_count, lr, self.betas[0], self.betas[1], self.eps,
                                    self.weight_decay, grad_scale, _lib.stream_ptr()), "mp_adam_step")

    def zero_grad(self, set_to_none: bool = True) -> None:
        for p in self.model.parameters():
            p.grad = None
        self.model._last_flat_grad = None

    def state_dict(self):
        """torch.optim.Adam's state_dict layout ({"state": {i: {"step", "exp_avg", "exp_avg_sq"}}, "param_groups": [...]}, parameter i =
        i-th entry of model.parameters()), so that params{tag}.pth files (main_h36m_lifting.py:75-98) are interchangeable with the
        reference's.  The moments are copies cut out of the flat buffers."""
        m = self.model
        m.flat_parameters()
        index = {id(p): i for i, p in enumerate(m.parameters())}
        state = {}
        if self.exp_avg is not None:
            for (off, n), p in zip(m._slots, m._plist):
                state[index[id(p)]] = {"step": torch.tensor(float(self.step_count)),
                                       "exp_avg": self.exp_avg[off:off + n].view(p.shape).clone(),
                                       "exp_avg_sq": self.exp_avg_sq[off:off + n].view(p.shape).clone()}
        group = {"lr": self.param_groups[0]["lr"], "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(index)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        """Accepts torch.optim.Adam's layout (a reference checkpoint, or one written by state_dict() above), the several-group layout
        mup.optim.MuAdam writes (a reference params*.pth of a model.mup run; the third-party package is absent offline, its published
        grouping is restated: parity unpinned for that part) and the flat layout {"step", "exp_avg", "exp_avg_sq"} this class wrote before."""
        m = self.model
        flat = m.flat_parameters()
        if "param_groups" in sd:
            groups = sd["param_groups"]
            params = list(m.parameters())
            if len(groups) == 1:
                order = params                                         # torch.optim.Adam: index i = i-th entry of model.parameters()
                self.param_groups[0]["lr"] = groups[0].get("lr", self.lr)
            else:
                # mup.optim.MuAdam (mup 1.0.0, the reference's optimizer under model.mup: main_h36m_lifting.py:227-232) splits the one
                # group into one group per width multiplier of the matrix-like parameters (two infinite dimensions; lr / width_mult,
                # weight_decay * width_mult; in first-seen order) followed by one group of everything else; the state indices run
                # through the groups in that order.  Rebuild the same order from the model's infshapes (mup_lite.set_base_shapes).
                if any(getattr(p, "infshape", None) is None for p in params):
                    raise ValueError("optimizer state with several param groups (MuAdam) needs a model with base shapes: build it with "
                                     "model.mup=true / mup_lite.set_base_shapes first")
                matrix, vector = {}, []
                for p in params:
                    if p.infshape.ninf() == 2:
                        matrix.setdefault(p.infshape.width_mult(), []).append(p)
                    else:
                        vector.append(p)
                mine = list(matrix.values()) + [vector]
                if [len(g["params"]) for g in groups] != [len(g) for g in mine]:
                    raise ValueError(f"MuAdam optimizer state has groups of {[len(g['params']) for g in groups]} parameters, the model's base "
                                     f"shapes give {[len(g) for g in mine]}")
                order = [p for g in mine for p in g]
                self.param_groups[0]["lr"] = groups[-1].get("lr", self.lr)       # the vector-like group trains at the base learning rate
            n_sd = sum(len(g["params"]) for g in groups)
            if n_sd != len(params):
                raise ValueError(f"optimizer state has {n_sd} parameters, the model {len(params)}")
            ids = [pid for g in groups for pid in g["params"]]
            slot = {id(p): s for s, p in zip(m._slots, m._plist)}
            self.exp_avg, self.exp_avg_sq, self.step_count = torch.zeros_like(flat), torch.zeros_like(flat), 0
            for i, pid in enumerate(ids):
                st = sd["state"].get(pid)
                if st is None:
                    continue
                off, n = slot[id(order[i])]
                if st["exp_avg"].numel() != n:
                    raise ValueError(f"optimizer state of parameter {i} has {st['exp_avg'].numel()} elements, expected {n}")
                self.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1).to(flat.device, torch.float32))
                self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1).to(flat.device, torch.float32))
                self.step_count = max(self.step_count, int(float(st["step"])))
            return
        self.step_count = int(sd["step"])
        self.param_groups[0]["lr"] = sd.get("lr", self.lr)
        self.exp_avg = sd["exp_avg"].to(flat.device) if sd["exp_avg"] is not None else None
        self.exp_avg_sq = sd["exp_avg_sq"].to(flat.device) if sd["exp_avg_sq"] is not None else None


class CosineAnnealingLR:
    """torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max, eta_min) as the reference builds it (main_h36m_lifting.py:244-250),
    in closed form, for optimizers that only expose ``param_groups[0]["lr"]``; ``step()`` once per epoch."""

    def __init__(self, optimizer, T_max: int, eta_min: float = 0.0):
        self.optimizer, self.T_max, self.eta_min = optimizer, int(T_max), float(eta_min)
        self.base_lr = float(optimizer.param_groups[0]["lr"])
        self.last_epoch = 0

    def get_last_lr(self):
        return [self.optimizer.param_groups[0]["lr"]]

    def step(self, metric=None) -> None:
        import math
        self.last_epoch += 1
        lr = self.eta_min + (self.base_lr - self.eta_min) * (1 + math.cos(math.pi * self.last_epoch / self.T_max)) / 2
        self.optimizer.param_groups[0]["lr"] = lr

    def state_dict(self):
        return {"kind": "cosine", "T_max": self.T_max, "eta_min": self.eta_min, "base_lr": self.base_lr, "last_epoch": self.last_epoch}

    def load_state_dict(self, sd) -> None:
        self.T_max, self.eta_min, self.base_lr, self.last_epoch = sd["T_max"], sd["eta_min"], sd["base_lr"], sd["last_epoch"]


class ReduceLROnPlateau:
    """torch.optim.lr_scheduler.ReduceLROnPlateau(mode="min", threshold_mode="rel", cooldown=0, eps=1e-8) as the reference builds it
    (main_h36m_lifting.py:251-259); ``step(metric)`` after every validation."""

    def __init__(self, optimizer, mode: str = "min", factor: float = 0.5, patience: int = 10, threshold: float = 1e-4, min_lr: float = 0.0,
                 eps: float = 1e-8):
        if mode != "min":
            raise ValueError("only mode='min' is used by the lifting entry points")
        self.optimizer, self.factor, self.patience, self.threshold, self.min_lr, self.eps = optimizer, factor, patience, threshold, min_lr, eps
        self.best, self.num_bad_epochs, self.last_epoch = float("inf"), 0, 0

    def get_last_lr(self):
        return [self.optimizer.param_groups[0]["lr"]]

    def step(self, metric) -> None:
        current = float(metric)
        self.last_epoch += 1
        if current < self.best * (1.0 - self.threshold):
            self.best, self.num_bad_epochs = current, 0
        else:
            self.num_bad_epochs += 1
        if self.num_bad_epochs > self.patience:
            old = float(self.optimizer.param_groups[0]["lr"])
            new = max(old * self.factor, self.min_lr)
            if old - new > self.eps:
                self.optimizer.param_groups[0]["lr"] = new
            self.num_bad_epochs = 0

    def state_dict(self):
        return {"kind": "plateau", "best": self.best, "num_bad_epochs": self.num_bad_epochs, "last_epoch": self.last_epoch,
                "factor": self.factor, "patience": self.patience, "threshold": self.threshold, "min_lr": self.min_lr}

    def load_state_dict(self, sd) -> None:
        self.best, self.num_bad_epochs, self.last_epoch = sd["best"], sd["num_bad_epochs"], sd["last_epoch"]


def make_lr_scheduler(optimizer, kind: str, epochs: int, n_annealing: int = 1, lr_min: float = 0.0, lr_patience: int = 11,
                      lr_threshold: float = 0.1):
    """The scheduler choice of the reference's train() (main_h36m_lifting.py:243-264)."""
    if kind == "cosine":
        return CosineAnnealingLR(optimizer, T_max=epochs // n_annealing, eta_min=lr_min)
    if kind == "plateau":
        return ReduceLROnPlateau(optimizer, mode="min", factor=0.5, min_lr=lr_min, patience=lr_patience, threshold=lr_threshold)
    raise ValueError(f"Accepted lr_scheduler values are 'cosine' and 'plateau'.Got {kind}.")
