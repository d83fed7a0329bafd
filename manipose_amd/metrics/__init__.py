from .losses import (STANDARD_H36M_WEIGHTS, manifold_training_loss, mean_velocity_error, rmcl_training_loss,
                     smoothness_regularization, weighted_mpjpe_loss, wta_l2_loss_and_activate_head, wta_with_scoring_loss)
from .mean_joint_errors import mpjpe_error

__all__ = ["STANDARD_H36M_WEIGHTS", "manifold_training_loss", "mean_velocity_error", "rmcl_training_loss",
           "smoothness_regularization", "weighted_mpjpe_loss", "wta_l2_loss_and_activate_head", "wta_with_scoring_loss",
           "mpjpe_error"]
