from .generators import PoseSequenceGenerator
from .skeleton import Skeleton, h36m_skeleton, T_POSE_OPERATORS, H36M_PARENTS, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT
from .ingest import Dataset3DHP, Human36mDataset, create_2d_data, fetch, read_3d_data

__all__ = ["PoseSequenceGenerator", "Skeleton", "Human36mDataset", "Dataset3DHP", "read_3d_data", "create_2d_data", "fetch",
           "h36m_skeleton", "T_POSE_OPERATORS", "H36M_PARENTS", "H36M_JOINTS_LEFT", "H36M_JOINTS_RIGHT"]
