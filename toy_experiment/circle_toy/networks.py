"""Models of the toy experiment: a plain MLP regressing the 2-D point, the manifold-constrained MLP that regresses an ANGLE and
decodes it onto the circle (so every prediction lies on the manifold by construction - the 1-D analogue of ManiPose's rotation + forward
kinematics decoder), and its multi-hypothesis (rMCL) version.  Counterpart of the reference's toy_experiment/models/{mlp.py:5-39,
constrained_mlp.py:9-35, constrained_mlp_rmcl.py:8-123, squared_relu.py}; module tree, state-dict keys and the order in which the
initialisers consume the torch RNG are the reference's (seeded models are bit-identical: tests/golden/toy.npz)."""
from __future__ import annotations

import torch
from torch import nn
import torch.nn.functional as F


class SquaredReLU(nn.Module):
    def forward(self, x):
        return F.relu(x) ** 2


def _unit(n_in: int, n_out: int, act: nn.Module) -> nn.Sequential:
    return nn.Sequential(nn.Linear(n_in, n_out), act, nn.BatchNorm1d(n_out))       # Linear -> activation -> BatchNorm


class Mlp(nn.Module):
    """fc_in, `n_layers` hidden units, fc_out; ONE activation module shared by every unit."""

    def __init__(self, in_features: int, hidden_features: int, out_features: int, n_layers: int, act_layer=nn.Tanh):
        super().__init__()
        self.act = act_layer()
        self.fc_in = _unit(in_features, hidden_features, self.act)
        self.fcs = nn.Sequential(*[_unit(hidden_features, hidden_features, self.act) for _ in range(n_layers)])
        self.fc_out = nn.Linear(hidden_features, out_features)

    def trunk(self, x: torch.Tensor) -> torch.Tensor:
        return self.fcs(self.fc_in(x))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.fc_out(self.trunk(x))


class ConstrainedMlp(Mlp):
    """x -> theta -> (r cos theta, r sin theta)."""

    def __init__(self, in_features: int, hidden_features: int, n_layers: int, out_features: int = 1, act_layer=nn.ReLU, radius: float = 1.0):
        super().__init__(in_features, hidden_features, out_features, n_layers, act_layer)
        self.radius = radius

    def polar2cartesian(self, theta: torch.Tensor):
        return self.radius * torch.cos(theta), self.radius * torch.sin(theta)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return torch.cat(self.polar2cartesian(super().forward(x)), dim=1)


class ConstrainedMlpRmcl(ConstrainedMlp):
    """`n_hyp` heads, each an angle and a score logit; output (B, H, 3) = (x, y, score) with the scores soft-maxed over the heads.
    The loss is the resilient multiple-choice-learning loss of ManiPose: winner-takes-all L2 + beta * BCE of the scores against the one-hot
    winner.  (The reference's module-level loss helpers are shadowed by their 3-D versions at HEAD, which breaks this 2-D model there;
    the 2-D definitions are the ones implemented here.)"""

    def __init__(self, in_features: int, hidden_features: int, n_layers: int, out_features: int = 1, act_layer=nn.ReLU, radius: float = 1.0,
                 n_hyp: int = 5, beta: float = 1.0):
        super().__init__(in_features=in_features, hidden_features=hidden_features, out_features=out_features, n_layers=n_layers,
                         act_layer=act_layer, radius=radius)
        self.n_hyp, self.beta = n_hyp, beta
        self.fc_out = nn.ModuleList([nn.Linear(hidden_features, out_features + 1) for _ in range(n_hyp)])      # replaces the single head

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        z = self.trunk(x)
        heads = torch.stack([head(z) for head in self.fc_out], dim=1)                      # (B, H, 2): angle, logit
        px, py = self.polar2cartesian(heads[..., 0])
        return torch.stack([px, py, heads[..., 1].softmax(dim=1)], dim=2)

    def aggregate(self, hypothesis: torch.Tensor, mode: str = "weighted_ave") -> torch.Tensor:
        if mode == "weighted_ave":
            return (hypothesis[..., :2] * hypothesis[..., 2:3]).sum(dim=1)
        if mode == "best_score":
            best = hypothesis[..., 2].argmax(dim=1)
            return hypothesis[torch.arange(hypothesis.shape[0]), best, :2]
        raise ValueError(f"Only best_score and weighted_ave modes are implemented.Got {mode}.")

    def wta_with_scoring_l2_loss(self, hypothesis: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        per_hyp = l2_loss_per_hyp(hypothesis, y)
        wta, winner = per_hyp.min(dim=1)
        if self.beta == 0:
            return wta.mean()
        target = F.one_hot(winner, per_hyp.shape[1]).to(per_hyp.dtype)
        return wta.mean() + self.beta * F.binary_cross_entropy(hypothesis[..., 2], target)


def l2_loss_per_hyp(hypothesis: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """(B, H): mean squared coordinate error of every hypothesis."""
    return ((hypothesis[..., :2] - y[:, None, :]) ** 2).mean(dim=2)


def wta_l2_loss(hypothesis: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    return l2_loss_per_hyp(hypothesis, y).min(dim=1).values
