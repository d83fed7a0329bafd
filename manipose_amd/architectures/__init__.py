from .mix_ste import MixSTE
from .manifold_mix_ste import ManifoldMixSTE, BonesMixSTE
from .rmcl_manifold_mix_ste import RMCLManifoldMixSTE, RMCLRotMixSTE, MCLHead
from .pose_decoder import PoseDecoder
from .engine import LiftEngine

__all__ = ["MixSTE", "ManifoldMixSTE", "BonesMixSTE", "RMCLManifoldMixSTE", "RMCLRotMixSTE", "MCLHead", "PoseDecoder",
           "LiftEngine"]
