from .skeleton import Skeleton, h36m_skeleton, T_POSE_OPERATORS, H36M_PARENTS, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT

__all__ = ["Skeleton", "h36m_skeleton", "T_POSE_OPERATORS", "H36M_PARENTS", "H36M_JOINTS_LEFT", "H36M_JOINTS_RIGHT"]
