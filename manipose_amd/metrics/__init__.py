from .analytics import PoseAnalytics, pose_analytics
from .losses import (STANDARD_H36M_WEIGHTS, manifold_training_loss, mean_velocity_error, rmcl_training_loss,
                     smoothness_regularization, weighted_mpjpe_loss, weighted_mse_loss, wta_l2_loss_and_activate_head, wta_with_scoring_loss)
from .mean_joint_errors import jointwise_error, jointwise_mse, mpjpe_error, mse_error, p_mpjpe, segments_len_err
from .pck import keypoint_3d_auc, keypoint_3d_pck
from .regularizations import (sagittal_symmetry, sagittal_symmetry_per_bone, segments_time_consistency,
                              segments_time_consistency_per_bone)

__all__ = ["STANDARD_H36M_WEIGHTS", "manifold_training_loss", "mean_velocity_error", "rmcl_training_loss",
           "smoothness_regularization", "weighted_mpjpe_loss", "weighted_mse_loss", "wta_l2_loss_and_activate_head", "wta_with_scoring_loss",
           "mpjpe_error", "p_mpjpe", "mse_error", "jointwise_error", "jointwise_mse", "segments_len_err", "keypoint_3d_pck", "keypoint_3d_auc",
           "sagittal_symmetry", "sagittal_symmetry_per_bone", "segments_time_consistency", "segments_time_consistency_per_bone",
           "PoseAnalytics", "pose_analytics"]
