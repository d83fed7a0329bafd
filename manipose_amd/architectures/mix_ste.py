"""Parameter containers of the MixSTE backbone with the reference's module tree and constructor contract.

Mirrors the *interface* of hpe/mh_so3_hpe/architectures/mix_ste.py (MixSTE :12-126, Mlp :194-214, Attention
:225-253, Block :285-350): same constructor arguments, attribute names, parameter shapes, registration order (so
that the default torch initialisation consumes the RNG exactly like the reference and state-dict keys are
identical).  None of these modules computes anything in PyTorch: the arithmetic of
``MixSTE.STE_forward / TTE_foward / ST_foward``, ``Block.forward``, ``Attention.forward`` and ``Mlp.forward``
runs inside the native engine (manipose_amd/csrc/engine.hip) launched by the model that owns the backbone.
"""
from __future__ import annotations

from functools import partial

import torch
import torch.nn as nn

from ..mup_lite import MuReadout
from ._fused import FusedLiftingMixin

_ENGINE_ONLY = ("manipose_amd: {} holds parameters only; its arithmetic runs inside the fused HIP engine. Call the "
                "owning model (RMCLManifoldMixSTE / ManifoldMixSTE) on a ROCm tensor instead.")


class DropPath(nn.Module):
    """Stochastic-depth marker (timm.models.layers.DropPath in the reference, mix_ste.py:8,334-336). The per-sample
    Bernoulli(keep)/keep masks are drawn by the engine (csrc/elementwise.hip: droppath_masks_kernel)."""

    def __init__(self, drop_prob: float = 0.0, scale_by_keep: bool = True):
        super().__init__()
        self.drop_prob = drop_prob
        self.scale_by_keep = scale_by_keep

    def forward(self, x):
        raise RuntimeError(_ENGINE_ONLY.format("DropPath"))


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0,
                 changedim=False, currentdim=0, depth=0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        if act_layer is not nn.GELU or drop != 0.0:
            raise NotImplementedError("manipose_amd: the fused MLP kernel implements exact-erf GELU with dropout 0 "
                                      "(the only setting the reference uses)")
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        raise RuntimeError(_ENGINE_ONLY.format("Mlp"))


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0.0, proj_drop=0.0, comb=False,
                 vis=False, mup=False):
        super().__init__()
        if comb or attn_drop != 0.0 or proj_drop != 0.0 or not qkv_bias:
            raise NotImplementedError("manipose_amd: attention kernels implement the reference defaults (qkv_bias=True, comb=False, "
                                      "no dropout)")
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or (1 / head_dim if mup else head_dim ** -0.5)          # mix_ste.py:243-244
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.comb = comb
        self.vis = vis

    def forward(self, x, vis=False):
        raise RuntimeError(_ENGINE_ONLY.format("Attention"))


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, attention=Attention, qkv_bias=False, qk_scale=None, drop=0.0,
                 attn_drop=0.0, drop_path=0.0, act_layer=nn.GELU, norm_layer=nn.LayerNorm, comb=False, changedim=False,
                 currentdim=0, depth=0, vis=False, mup=False):
        super().__init__()
        if changedim:
            raise NotImplementedError("manipose_amd: changedim blocks are never built by the reference models")
        self.changedim, self.currentdim, self.depth = changedim, currentdim, depth
        self.norm1 = norm_layer(dim)
        self.attn = attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop,
                              proj_drop=drop, comb=comb, vis=vis, mup=mup)
        self.residual_scale = 1 / depth ** 0.5 if mup else 1.0                        # mix_ste.py:327-330 (muP across depth)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.vis = vis

    def forward(self, x, vis=False):
        raise RuntimeError(_ENGINE_ONLY.format("Block"))


class MixSTE(FusedLiftingMixin, nn.Module):
    """The MixSTE backbone (reference mix_ste.py:12-191).  Inside ManifoldMixSTE / RMCLManifoldMixSTE it is a parameter container
    (the owning model's engine runs it).  On its own - ``model.arch=mixste`` of the entry points (main_h36m_lifting.py:617-628):
    ``MixSTE(num_frame, num_joints=17, in_chans=2, out_dim=3, ...)`` - ``forward`` maps (B, T, 17, 2) keypoints straight to
    (B, T, 17, 3) poses through the same engine (arch 2: backbone + LayerNorm/Linear head, no bones net, no decoder)."""

    def __init__(self, num_frame=243, num_joints=17, in_chans=2, out_dim=3, embed_dim=512, depth=8, num_heads=8,
                 mlp_ratio=2.0, qkv_bias=True, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.2,
                 norm_layer=None, mup=False):
        super().__init__()
        if mlp_ratio != 2.0 or norm_layer is not None or drop_rate != 0.0:
            raise NotImplementedError("manipose_amd: the engine is built for mlp_ratio=2, LayerNorm(eps=1e-6), drop_rate=0")
        norm_layer = partial(nn.LayerNorm, eps=1e-6)
        self.embed_dim = embed_dim
        self.num_frame = num_frame
        self.num_heads = num_heads
        self.drop_path_rate = drop_path_rate
        self.Spatial_patch_to_embedding = nn.Linear(in_chans, embed_dim)
        self.Spatial_pos_embed = nn.Parameter(torch.zeros(1, num_joints, embed_dim))
        self.Temporal_pos_embed = nn.Parameter(torch.zeros(1, num_frame, embed_dim))
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, depth)]
        self.block_depth = depth
        common = dict(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                      drop=drop_rate, attn_drop=attn_drop_rate, norm_layer=norm_layer, mup=mup)
        self.STEblocks = nn.ModuleList([Block(drop_path=dpr[i], depth=depth if mup else 0, **common) for i in range(depth)])
        self.TTEblocks = nn.ModuleList([Block(drop_path=dpr[i], currentdim=i + 1, depth=depth, **common)
                                        for i in range(depth)])
        self.Spatial_norm = norm_layer(embed_dim)
        self.Temporal_norm = norm_layer(embed_dim)
        self.head = nn.Sequential(nn.LayerNorm(embed_dim), (MuReadout if mup else nn.Linear)(embed_dim, out_dim))    # mix_ste.py:118-126
        self._standalone = in_chans == 2 and out_dim == 3 and num_joints == 17
        self._init_fused("mixste", dict(num_frame=num_frame, num_joints=num_joints, num_bones=16, embed_dim_rot=embed_dim, depth_rot=depth,
                                        num_heads_rot=num_heads, embed_dim_seg=128, depth_seg=1, num_heads_seg=8, n_hyp=1,
                                        drop_path_rate=drop_path_rate))

    def forward(self, x):
        """x (B, T, 17, 2) -> (B, T, 17, 3) (mix_ste.py:175-191)."""
        if not self._standalone or type(self) is not MixSTE:
            raise RuntimeError(_ENGINE_ONLY.format(type(self).__name__))
        return self._run(x)[:, 0]
