"""RMCLManifoldMixSTE: the multi-hypothesis ManiPose model, reference contract of
hpe/mh_so3_hpe/architectures/rmcl_manifold_mix_ste.py (RMCLManifoldMixSTE :15-185, RMCLRotMixSTE :188-237,
MCLHead :267-289).  ``forward`` = one native engine call; ``aggregate`` runs the HIP aggregation kernel."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .. import _lib
from .manifold_mix_ste import ManifoldMixSTE
from ..mup_lite import MuReadout
from .mix_ste import MixSTE


class MCLHead(nn.Module):
    """One hypothesis head: LayerNorm -> Linear(C, out_dim + 1); Linear(num_joints, 1) scores the extra channel."""

    def __init__(self, embed_dim: int, out_dim: int, num_joints: int, mup: bool = False):
        super().__init__()
        self.norm = nn.LayerNorm(embed_dim)
        linear = MuReadout if mup else nn.Linear                                      # rmcl_manifold_mix_ste.py:278-289
        self.prediction_head = linear(embed_dim, out_dim + 1)
        self.score_head = linear(num_joints, 1)


class RMCLRotMixSTE(MixSTE):
    def __init__(self, num_frame: int = 243, num_joints: int = 17, in_chans: int = 2, out_dim=6, embed_dim: int = 512,
                 depth: int = 8, num_heads: int = 8, mlp_ratio: float = 2.0, qkv_bias: bool = True, qk_scale: float = None,
                 drop_rate: float = 0.0, attn_drop_rate: float = 0.0, drop_path_rate: float = 0.2,
                 norm_layer: nn.Module = None, n_hyp: int = 5, mup: bool = False):
        # like the reference (:208-223) the backbone is built with mup=False regardless of the flag
        super().__init__(num_frame, num_joints, in_chans, out_dim, embed_dim, depth, num_heads, mlp_ratio, qkv_bias,
                         qk_scale, drop_rate, attn_drop_rate, drop_path_rate, norm_layer)
        self.n_hyp = n_hyp
        self.head = nn.ModuleList([MCLHead(embed_dim=embed_dim, out_dim=out_dim, num_joints=num_joints, mup=mup)
                                   for _ in range(self.n_hyp)])


class RMCLManifoldMixSTE(ManifoldMixSTE):
    def __init__(self, skeleton, num_frame: int = 243, num_joints: int = 17, num_bones: int = 16, in_chans: int = 2,
                 rot_rep_dim: int = 6, embed_dim_rot: int = 512, depth_rot: int = 8, num_heads_rot: int = 8,
                 embed_dim_seg: int = 128, depth_seg: int = 2, num_heads_seg: int = 8, mlp_ratio: float = 2.0,
                 qkv_bias: bool = True, qk_scale: float = None, drop_rate: float = 0.0, attn_drop_rate: float = 0.0,
                 drop_path_rate: float = 0.2, norm_layer: nn.Module = None, n_hyp: int = 5, mup: bool = False):
        # building the single-hypothesis parent first reproduces the reference's RNG consumption at init (:40-61)
        super().__init__(skeleton=skeleton, num_frame=num_frame, num_joints=num_joints, num_bones=num_bones,
                         in_chans=in_chans, rot_rep_dim=rot_rep_dim, embed_dim_rot=embed_dim_rot, depth_rot=depth_rot,
                         num_heads_rot=num_heads_rot, embed_dim_seg=embed_dim_seg, depth_seg=depth_seg,
                         num_heads_seg=num_heads_seg, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                         drop_rate=drop_rate, attn_drop_rate=attn_drop_rate, drop_path_rate=drop_path_rate,
                         norm_layer=norm_layer, mup=mup)
        if not 1 <= n_hyp <= 8:
            raise NotImplementedError("manipose_amd: 1 <= n_hyp <= 8 hypotheses are supported by the HIP kernels")
        self.n_hyp = n_hyp
        self.rotations_module = RMCLRotMixSTE(num_frame=num_frame, num_joints=num_joints, in_chans=in_chans,
                                              out_dim=rot_rep_dim, embed_dim=embed_dim_rot, depth=depth_rot,
                                              num_heads=num_heads_rot, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                                              qk_scale=qk_scale, drop_rate=drop_rate, attn_drop_rate=attn_drop_rate,
                                              drop_path_rate=drop_path_rate, norm_layer=norm_layer, n_hyp=n_hyp, mup=mup)
        cfg = dict(self._engine_cfg, n_hyp=n_hyp)
        self._init_fused("rmcl_manifold", cfg)

    def forward(self, x: torch.Tensor):
        """x (B, L, J, 2) -> (poses (B, H, L, J, 3), scores (B, H, L, 1)); scores sum to 1 over H (reference :83-106)."""
        return self._run(x)

    # ---- eval-time helpers (reference :108-185) ------------------------------------------------
    def concat_hyp_and_scores(self, hypothesis: torch.Tensor, scores: torch.Tensor) -> torch.Tensor:
        return torch.cat((hypothesis, scores.unsqueeze(3).expand(-1, -1, -1, self.num_joints, -1)), dim=-1)

    def poses_from_hyp_idx(self, hypothesis: torch.Tensor, hyp_indices: torch.Tensor) -> torch.Tensor:
        B, _, L, J, D = hypothesis.shape
        idx = hyp_indices.to(hypothesis.device)[:, None, :, None, None].expand(B, 1, L, J, D)
        return hypothesis.gather(1, idx)[:, 0]

    def aggregate(self, hypothesis: torch.Tensor, scores: torch.Tensor = None, mode: str = "weighted_ave",
                  ground_truth: Optional[torch.Tensor] = None):
        modes = {"weighted_ave": 0, "best_score": 1, "oracle": 2}
        if mode not in modes:
            raise ValueError(f"Only best_score and weighted_ave modes are implemented.Got {mode}.")
        if mode != "oracle":
            assert scores is not None, "Scores required to aggregate hypothesis."
        else:
            assert ground_truth is not None, "Ground-truth required to compute best hypothesis."
        lib = _lib.load()
        B, K, T = hypothesis.shape[:3]
        hyp = hypothesis.detach().contiguous().float()
        sc = scores.detach().contiguous().float() if scores is not None else None
        gt = ground_truth.detach().contiguous().float() if ground_truth is not None else None
        out = torch.empty(B, T, 17, 3, dtype=torch.float32, device=hyp.device)
        _lib.check(lib.mp_aggregate(_lib.ptr(hyp), _lib.ptr(sc), _lib.ptr(gt), modes[mode], _lib.ptr(out), B, K, T,
                                    _lib.stream_ptr()), "mp_aggregate")
        if mode == "oracle":
            from ..metrics.losses import wta_l2_loss_and_activate_head
            oracle_mpjpe, _ = wta_l2_loss_and_activate_head(hyp, gt, weights=None, squared=False)
            return oracle_mpjpe, out
        return out
