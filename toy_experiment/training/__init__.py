"""Import names of the reference's toy_experiment/training package (implementation: circle_toy/fit.py)."""
from circle_toy.fit import Trainer, calc_mpjpe, distance_to_circle, oracle_multihyp_mpjpe  # noqa: F401
