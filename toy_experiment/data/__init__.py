"""Import names of the reference's toy_experiment/data package (implementation: circle_toy/sampling.py)."""
from circle_toy.sampling import EasyDist, HardBimodalDist, HardQuadmodalDist, HardUnimodalDist, LiftingDataset, LiftingDist1Dto2D  # noqa: F401
