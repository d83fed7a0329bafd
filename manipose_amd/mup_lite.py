"""The slice of the `mup` package (Maximal Update Parametrization, mup==1.0.0 in the reference's requirements_frozen.txt:8) that the
reference's `model.mup=True` mode uses (mix_ste.py:9,118-121,243,330; rmcl_manifold_mix_ste.py:12,278-289; main_h36m_lifting.py:19-20,
227-232,673-708,762-764; utils/mup_utils.py).  `mup` is a third-party package that is neither vendored by the reference nor installed
here, so this is a restatement of its published algorithm (Yang & Hu et al., "Tensor Programs V", and the package's documented
behaviour): PARITY UNPINNED - nothing in the reference tests it and the package cannot be run here.  What IS pinned by fixtures generated
from the reference's own code is everything the reference itself implements for muP: the 1 / head_dim attention scale and the
1 / sqrt(depth) residual scale (tests/golden/mup_manifold.npz).

  MuReadout          nn.Linear whose input is multiplied by output_mult / width_mult
  set_base_shapes    gives every parameter an `infshape` (which of its dimensions grow with the model width, and by how much relative to a
                     base model), rescales the MuReadout parameters once (standard -> muP initialisation)
  mu_init_params     the reference's re-initialisation (utils/mup_utils.py:8-21)
  mup_lr_multipliers per-parameter learning-rate / weight-decay multipliers of MuAdam: matrix-like parameters (two width dimensions)
                     train with lr / width_mult (and weight_decay * width_mult), everything else with the plain values
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch
from torch import nn


class InfShape:
    """Per-dimension (base_dim, dim) of one parameter; a dimension is a WIDTH dimension when it differs between the base and the delta
    model (mup.make_base_shapes)."""

    def __init__(self, dims, base_dims, is_inf):
        self.dims, self.base_dims, self.is_inf = tuple(dims), tuple(base_dims), tuple(is_inf)

    def ninf(self) -> int:
        return sum(self.is_inf)

    def width_mult(self) -> float:
        """dim / base_dim of the fan-in width dimension (mup.infshape.InfShape.width_mult: the last width dimension of matrix-like shapes,
        the only width dimension of vector-like ones, 1 for shapes without width dimensions)."""
        idx = [i for i, f in enumerate(self.is_inf) if f]
        if not idx:
            return 1.0
        i = idx[-1] if len(idx) >= 2 else idx[0]
        return self.dims[i] / self.base_dims[i]


class MuReadout(nn.Linear):
    """mup.MuReadout: the output layer of a muP network, y = W (output_mult * x / width_mult) + b."""

    def __init__(self, *args, readout_zero_init: bool = False, output_mult: float = 1.0, **kwargs):
        self.output_mult = output_mult
        self.readout_zero_init = readout_zero_init
        super().__init__(*args, **kwargs)
        self._has_rescaled_params = False

    def reset_parameters(self) -> None:
        if self.readout_zero_init:
            self.weight.data[:] = 0
            if self.bias is not None:
                self.bias.data[:] = 0
        else:
            super().reset_parameters()

    def width_mult(self) -> float:
        shape = getattr(self.weight, "infshape", None)
        return shape.width_mult() if shape is not None else 1.0

    def input_multiplier(self) -> float:
        return self.output_mult / self.width_mult()

    def _rescale_parameters(self) -> None:
        if self._has_rescaled_params:
            raise RuntimeError("MuReadout parameters were already rescaled (set_base_shapes is not idempotent)")
        s = self.width_mult() ** 0.5
        with torch.no_grad():
            if self.bias is not None:
                self.bias.mul_(s)
            self.weight.mul_(s)
        self._has_rescaled_params = True

    def forward(self, x):
        return super().forward(self.input_multiplier() * x)


def make_base_shapes(base_model: nn.Module, delta_model: nn.Module) -> Dict[str, Tuple[Tuple[int, ...], Tuple[bool, ...]]]:
    """{parameter name: (base shape, which dimensions are width dimensions)} from a base model and a model that differs from it in every
    width to be scaled (the reference builds them at channels 64 / seq_len 27 and 128 / 81, main_h36m_lifting.py:681-693)."""
    delta = dict(delta_model.named_parameters())
    out = {}
    for name, p in base_model.named_parameters():
        q = delta[name]
        if p.dim() != q.dim():
            raise ValueError(f"{name}: rank differs between the base and the delta model")
        out[name] = (tuple(p.shape), tuple(a != b for a, b in zip(p.shape, q.shape)))
    return out


def set_base_shapes(model: nn.Module, base_shapes, rescale_params: bool = True) -> nn.Module:
    named = dict(model.named_parameters())
    if set(named) != set(base_shapes):
        raise ValueError(f"parameter names differ from the base shapes: {sorted(set(named) ^ set(base_shapes))[:4]}")
    for name, p in named.items():
        base, is_inf = base_shapes[name]
        p.infshape = InfShape(p.shape, base, is_inf)
    if rescale_params:
        for mod in model.modules():
            if isinstance(mod, MuReadout):
                mod._rescale_parameters()
    if hasattr(model, "_engine"):                 # the engine bakes the readout multipliers in: rebuild it on the next call
        model._engine = None
    return model


def mu_init_params(model: nn.Module) -> None:
    """utils/mup_utils.py:8-21: Kaiming-uniform(a = sqrt(5)) on every `*weight` that has a fan-in, uniform(+-1/sqrt(fan_in)) on the bias
    that follows it; 1-D weights (LayerNorm) and the positional tables keep their values (the reference warns and moves on).  mup.init's
    kaiming_uniform_ equals torch's for fan-in initialisation."""
    fan_in = 0
    for name, p in model.named_parameters():
        try:
            if name.endswith("weight"):
                fan_in, _ = nn.init._calculate_fan_in_and_fan_out(p)
                nn.init.kaiming_uniform_(p, a=math.sqrt(5))
            elif name.endswith("bias"):
                bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
                nn.init.uniform_(p, -bound, bound)
        except ValueError:
            pass


def mup_lr_multipliers(model: nn.Module) -> Dict[str, Tuple[float, float]]:
    """{name: (lr multiplier, weight-decay multiplier)} of mup.optim.MuAdam (coupled weight decay, as torch.optim.Adam's)."""
    out = {}
    for name, p in model.named_parameters():
        shape = getattr(p, "infshape", None)
        if shape is None:
            raise ValueError(f"{name} has no infshape: call set_base_shapes(model, ...) first")
        if shape.ninf() > 2:
            raise NotImplementedError(f"{name}: more than two width dimensions")
        if shape.ninf() == 2:
            wm = shape.width_mult()
            out[name] = (1.0 / wm, wm)
        else:
            out[name] = (1.0, 1.0)
    return out
