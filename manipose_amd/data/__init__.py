from .generators import PoseSequenceGenerator
from .skeleton import Skeleton, h36m_skeleton, T_POSE_OPERATORS, H36M_PARENTS, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT

__all__ = ["PoseSequenceGenerator", "Skeleton", "h36m_skeleton", "T_POSE_OPERATORS", "H36M_PARENTS", "H36M_JOINTS_LEFT", "H36M_JOINTS_RIGHT"]
