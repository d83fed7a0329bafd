"""manipose_amd: MI355X (gfx950)-native implementation of ManiPose's 2D->3D lifting hot path.

Host side in Python (the reference's nn.Module contract), compute in hand-written HIP kernels behind the C ABI of
include/manipose_hip.h (libmanipose_hip.so).  See DESIGN.md.
"""
from . import _lib  # noqa: F401
from .architectures import ManifoldMixSTE, MixSTE, RMCLManifoldMixSTE  # noqa: F401
from .data import Skeleton, h36m_skeleton  # noqa: F401

__version__ = "0.1.0"
from . import ops  # noqa: F401  (registers torch.ops.manipose.*)
