"""1-D -> 2-D lifting on a circle: the toy experiment of the ManiPose paper (BASELINE config #1; CPU, no GPU kernels involved)."""
from .fit import AverageMeter, Trainer, calc_mpjpe, distance_to_circle, oracle_multihyp_mpjpe
from .networks import ConstrainedMlp, ConstrainedMlpRmcl, Mlp, SquaredReLU, l2_loss_per_hyp, wta_l2_loss
from .sampling import (CircleMixture, EasyDist, HardBimodalDist, HardQuadmodalDist, HardUnimodalDist, LiftingDataset, LiftingDist1Dto2D,
                       SCENARIOS, scenario)
