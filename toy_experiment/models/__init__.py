"""Import names of the reference's toy_experiment/models package (implementation: circle_toy/networks.py)."""
from circle_toy.networks import ConstrainedMlp, ConstrainedMlpRmcl, Mlp, SquaredReLU  # noqa: F401
