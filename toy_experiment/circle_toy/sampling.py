"""Data side of the toy experiment: mixtures of von Mises distributions on a circle of given radius, the four scenarios of the paper's
figure, and the train / validation / test split drawn once.  Counterpart of the reference's toy_experiment/data/{distributions.py:10-58,
264-282, scenarios.py:6-47, datasets.py:11-58,115-119}; samples are bit-identical to the reference's for the same seed (same draws from the
same numpy Generator in the same order: tests/golden/toy.npz)."""
from __future__ import annotations

import numpy as np
import torch
from scipy.special import i0
from torch.utils.data import DataLoader, TensorDataset


class CircleMixture:
    """x -> (x, y) lifting on a circle: the angle follows sum_c w_c vonMises(mu_c, kappa_c); the input is the abscissa r cos(theta), the
    target the point (r cos(theta), r sin(theta)).  Two angles share every abscissa: the problem is multi-modal unless the mixture is
    concentrated on one half of the circle."""

    def __init__(self, radius: float, weights, modes, dispersions, random_state=None):
        self.weights = np.asarray(weights, dtype=float)
        self.modes = np.asarray(modes, dtype=float)
        self.dispersions = np.asarray(dispersions, dtype=float)
        if not (self.weights.shape == self.modes.shape == self.dispersions.shape) or self.weights.ndim != 1:
            raise ValueError("weights, modes and dispersions must be 1-D and of one length")
        if np.any(self.weights < 0) or abs(self.weights.sum() - 1.0) > 1e-12:
            raise ValueError("weights must be non-negative and sum to 1")
        if not radius > 0:
            raise ValueError("radius must be positive")
        self.radius = float(radius)
        self.rng = random_state if isinstance(random_state, np.random.Generator) else np.random.default_rng(random_state)

    def sample_angles(self, size: int) -> np.ndarray:
        which = self.rng.choice(np.arange(len(self.weights)), size=size, p=self.weights)     # component of every sample first ...
        theta = np.empty(size)
        for c in range(len(self.weights)):                                                   # ... then the angles, component by component
            sel = which == c
            theta[sel] = self.rng.vonmises(self.modes[c], kappa=self.dispersions[c], size=int(sel.sum()))
        return theta

    def sample(self, size: int):
        theta = self.sample_angles(size)
        x, y = self.radius * np.cos(theta), self.radius * np.sin(theta)
        return x, np.stack([x, y], axis=1)

    def pdf(self, theta) -> np.ndarray:
        t = np.atleast_1d(np.asarray(theta, dtype=float))[:, None]
        dens = self.weights * np.exp(self.dispersions * np.cos(t - self.modes)) / (2 * np.pi * i0(self.dispersions))
        return dens.sum(axis=1)


# scenario name (conf data.scenario) -> (weights, modes, dispersions); all concentrations 20
SCENARIOS = {
    "easy": ([1.0], [4 * np.pi / 10], [20]),                                                   # one mode in the upper half: unimodal lifting
    "hard-1": ([1.0], [0.0], [20]),                                                            # one mode across the x axis
    "hard-2": ([2 / 3, 1 / 3], [np.pi / 3, -np.pi / 3], [20, 20]),                             # mirrored modes: two answers per input
    "hard-4": ([0.3, 0.1, 0.4, 0.2], [5 * np.pi / 6, 7 * np.pi / 6, np.pi / 3, -np.pi / 3], [20] * 4),
}


def scenario(name: str, radius: float, random_state) -> CircleMixture:
    if name not in SCENARIOS:
        raise ValueError(f"Possible values for scenario are 'easy', 'hard-1', 'hard-2', 'hard-4' or 'torus-2Dto3D'. Got {name}.")
    w, m, k = SCENARIOS[name]
    return CircleMixture(radius, w, m, k, random_state)


def _named(name):
    def make(radius: float, random_state):
        return scenario(name, radius, random_state)
    return make


EasyDist, HardUnimodalDist, HardBimodalDist, HardQuadmodalDist = (_named(n) for n in ("easy", "hard-1", "hard-2", "hard-4"))
LiftingDist1Dto2D = CircleMixture


class LiftingDataset:
    """Training, validation and test sets drawn ONCE from the distribution, in this order, as float32 tensors (X (n, 1), Y (n, 2))."""

    def __init__(self, distribution: CircleMixture, n_train: int, n_val: int, n_test: int):
        self.distribution = distribution
        parts = []
        for n in (n_train, n_val, n_test):
            x, y = distribution.sample(n)
            parts.append((torch.from_numpy(x[:, None]).float(), torch.from_numpy(y).float()))
        (self.X_train, self.Y_train), (self.X_val, self.Y_val), (self.X_test, self.Y_test) = parts
        self.training_set, self.validation_set, self.test_set = (TensorDataset(a, b) for a, b in parts)

    def get_tr_loader(self, **kwargs) -> DataLoader:
        if "shuffle" in kwargs:
            raise ValueError("shuffle is fixed: True for the training loader, False for the others")
        return DataLoader(self.training_set, shuffle=True, **kwargs)

    def get_loaders(self, **kwargs):
        if "shuffle" in kwargs:
            raise ValueError("shuffle is fixed: True for the training loader, False for the others")
        return (DataLoader(self.training_set, shuffle=True, **kwargs), DataLoader(self.validation_set, shuffle=False, **kwargs),
                DataLoader(self.test_set, shuffle=False, **kwargs))
