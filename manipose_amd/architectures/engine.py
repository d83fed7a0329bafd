"""Python handle on the native model engine (``mp_model_*`` in include/manipose_hip.h)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import torch

from .. import _lib

# "bf16x3": split-precision forward (hi/lo bf16 planes, three matrix-core products per product: MPJPE within 1e-4 m of the fp32
# reference), bf16 backward
PRECISIONS = {"fp32": 0, "bf16": 1, "bf16x3": 2}


class LiftEngine:
    """Owns one ``mp_model`` (activation workspace + launch sequence) on the current ROCm device."""

    def __init__(self, *, arch: str, num_frame: int, num_joints: int, num_bones: int, embed_dim_rot: int, depth_rot: int,
                 num_heads_rot: int, embed_dim_seg: int, depth_seg: int, num_heads_seg: int, n_hyp: int,
                 drop_path_rate: float, max_batch: int, precision: str = "fp32", rot_rep_dim: int = 6, qk_scale_rot: float = 0.0,
                 resid_scale_rot: float = 0.0, readout_mult_rot: float = 0.0, qk_scale_seg: float = 0.0, resid_scale_seg: float = 0.0,
                 readout_mult_seg: float = 0.0, f16f8: int = 0, f16_backward: bool = False, side_stream: bool = True, wgrad_stream: bool = True,
                 hazard_check: bool = False):
        """f16f8 / f16_backward / side_stream / wgrad_stream: mp_model_config::f16f8, f16_backward, streams (include/manipose_hip.h) - the
        operand form of the qkv / fc1 (/ fc2) Linear layers of a bf16x3 model, and which of the engine's two extra streams it uses.
        hazard_check: mp_model_config::debug bit 0, the host-side stream-hazard check (hazard_report())."""
        self.lib = _lib.load()
        if precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {list(PRECISIONS)}, got {precision}")
        self.cfg = _lib.ModelConfig(arch={"rmcl_manifold": 0, "manifold": 1, "mixste": 2}[arch], num_frame=num_frame,
                                    num_joints=num_joints, num_bones=num_bones, embed_dim_rot=embed_dim_rot,
                                    depth_rot=depth_rot, num_heads_rot=num_heads_rot, embed_dim_seg=embed_dim_seg,
                                    depth_seg=depth_seg, num_heads_seg=num_heads_seg, n_hyp=max(1, n_hyp),
                                    drop_path_rate=drop_path_rate, max_batch=max_batch,
                                    precision=PRECISIONS[precision], rot_rep_dim=rot_rep_dim, qk_scale_rot=qk_scale_rot,
                                    resid_scale_rot=resid_scale_rot, readout_mult_rot=readout_mult_rot, qk_scale_seg=qk_scale_seg,
                                    resid_scale_seg=resid_scale_seg, readout_mult_seg=readout_mult_seg,
                                    f16f8=int(f16f8), f16_backward=int(bool(f16_backward)),
                                    streams=(0 if side_stream else 1) | (0 if wgrad_stream else 2), debug=int(bool(hazard_check)))
        self.arch = arch
        self.K = max(1, n_hyp) if arch == "rmcl_manifold" else 1
        self.max_batch = max_batch
        self.device = torch.device("cuda", torch.cuda.current_device()) if max_batch > 0 else torch.device("cpu")
        h = C.c_void_p()
        _lib.check(self.lib.mp_model_create(C.byref(self.cfg), C.byref(h)), "mp_model_create")
        self.handle = h
        self.forward_serial = 0          # counts mp_model_forward calls: the engine holds the activations of the last one only (_fused.py)
        self.flat_size = int(self.lib.mp_model_flat_size(h))
        self.layout: List[Tuple[str, int, int]] = []
        buf = C.create_string_buffer(256)
        off, num = C.c_int64(), C.c_int64()
        for i in range(self.lib.mp_model_num_params(h)):
            _lib.check(self.lib.mp_model_param_info(h, i, buf, 256, C.byref(off), C.byref(num)), "mp_model_param_info")
            self.layout.append((buf.value.decode(), int(off.value), int(num.value)))

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.mp_model_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    @property
    def workspace_bytes(self) -> int:
        return int(self.lib.mp_model_workspace_bytes(self.handle))

    def mask_layout(self, B: int) -> List[Tuple[str, int, int, float]]:
        buf = C.create_string_buffer(256)
        off, cnt, keep = C.c_int64(), C.c_int64(), C.c_float()
        out = []
        for i in range(self.lib.mp_model_num_mask_branches(self.handle)):
            _lib.check(self.lib.mp_model_mask_info(self.handle, B, i, buf, 256, C.byref(off), C.byref(cnt), C.byref(keep)),
                       "mp_model_mask_info")
            out.append((buf.value.decode(), int(off.value), int(cnt.value), float(keep.value)))
        return out

    def pack_masks(self, B: int, masks: Dict[str, torch.Tensor]) -> torch.Tensor:
        """Mask dict (branch name -> (count,) multipliers) -> flat device buffer in engine layout (missing = 1)."""
        n = int(self.lib.mp_model_mask_floats(self.handle, B))
        flat = torch.ones(max(n, 1), dtype=torch.float32, device=self.device)
        for name, off, cnt, _ in self.mask_layout(B):
            if name in masks:
                flat[off:off + cnt] = masks[name].reshape(-1).to(self.device, torch.float32)
        return flat

    def forward(self, flat_params: torch.Tensor, x: torch.Tensor, train: bool = False,
                masks: Optional[torch.Tensor] = None, seed: int = 0, step: int = 0, infer: bool = False):
        """infer: no backward will follow (torch.no_grad()): the engine skips what only the backward reads."""
        B, T = x.shape[0], x.shape[1]
        self.forward_serial += 1
        poses = torch.empty(B, self.K, T, 17, 3, dtype=torch.float32, device=x.device)
        scores = torch.empty(B, self.K, T, 1, dtype=torch.float32, device=x.device) if self.arch == "rmcl_manifold" else None
        _lib.check(self.lib.mp_model_forward(self.handle, _lib.ptr(flat_params), _lib.ptr(x), B, _lib.ptr(poses),
                                             _lib.ptr(scores), int(bool(train)) | (2 if infer else 0), _lib.ptr(masks), seed, step, _lib.stream_ptr()),
                   "mp_model_forward")
        return poses, scores

    def backward(self, flat_params: torch.Tensor, flat_grads: torch.Tensor, d_poses: torch.Tensor,
                 d_scores: Optional[torch.Tensor]) -> None:
        _lib.check(self.lib.mp_model_backward(self.handle, _lib.ptr(flat_params), _lib.ptr(flat_grads), _lib.ptr(d_poses),
                                              _lib.ptr(d_scores), _lib.stream_ptr()), "mp_model_backward")

    def peek(self, which: int) -> torch.Tensor:
        """Copy of an intermediate of the last forward: 0 head output (K, B*T*17, O), 1 segment lengths (B, 16)."""
        p, n = C.c_void_p(), C.c_int64()
        _lib.check(self.lib.mp_model_peek(self.handle, which, C.byref(p), C.byref(n)), "mp_model_peek")
        out = torch.empty(int(n.value), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.mp_model_peek_copy(self.handle, which, _lib.ptr(out), out.numel(), _lib.stream_ptr()),
                   "mp_model_peek_copy")
        return out

    def grad_buckets(self) -> List[Tuple[int, int]]:
        """[(offset, numel)] of the gradient buckets (one per layer of the rotations net) in the flat gradient buffer."""
        off, n = C.c_int64(), C.c_int64()
        out = []
        for i in range(self.lib.mp_model_grad_bucket_count(self.handle)):
            _lib.check(self.lib.mp_model_grad_bucket_info(self.handle, i, C.byref(off), C.byref(n)), "mp_model_grad_bucket_info")
            out.append((int(off.value), int(n.value)))
        return out

    def grad_bucket_wait(self, index: int, stream: "torch.cuda.Stream") -> None:
        """Make `stream` wait (on the device) until bucket `index` of the last backward is final."""
        _lib.check(self.lib.mp_model_grad_bucket_wait(self.handle, index, stream.cuda_stream), "mp_model_grad_bucket_wait")

    def grad_health(self) -> Dict[str, float]:
        """Scaled-fp16 gradient operands of the last backward (f16_backward models): the scale S and how many stores hit the +-65504 clamp /
        met a non-finite value.  Synchronises the current stream."""
        out = (C.c_float * 4)()
        _lib.check(self.lib.mp_model_grad_health(self.handle, out, _lib.stream_ptr()), "mp_model_grad_health")
        return {"scale": out[0], "saturated": int(out[1]), "non_finite": int(out[2]), "inv_scale": out[3]}

    def grad_health_async(self, out: torch.Tensor) -> None:
        """The same four values (scale, saturated, non_finite, inv_scale) as floats into the device tensor `out` (4 floats), enqueued on the
        current stream without a host synchronisation."""
        _lib.check(self.lib.mp_model_grad_health_async(self.handle, _lib.ptr(out), _lib.stream_ptr()), "mp_model_grad_health_async")

    def set_streams(self, side_stream: bool = True, wgrad_stream: bool = True) -> None:
        """Which of the engine's two extra streams the next forward / backward calls use (mp_model_set_streams)."""
        _lib.check(self.lib.mp_model_set_streams(self.handle, (0 if side_stream else 1) | (0 if wgrad_stream else 2)), "mp_model_set_streams")

    def hazard_report(self) -> Dict[str, object]:
        """Stream-hazard check (hazard_check=True): launches declared, conflicting cross-stream pairs found ordered by an event path, pairs
        found UNORDERED (violations) with their descriptions, events recorded."""
        out = (C.c_int64 * 4)()
        buf = C.create_string_buffer(1 << 15)
        _lib.check(self.lib.mp_model_hazard_report(self.handle, out, buf, len(buf)), "mp_model_hazard_report")
        return {"launches": int(out[0]), "ordered_pairs": int(out[1]), "violations": int(out[2]), "events": int(out[3]),
                "messages": [l for l in buf.value.decode().splitlines() if l]}

    def prof_enable(self, on: bool = True) -> None:
        _lib.check(self.lib.mp_prof_enable(self.handle, int(on)), "mp_prof_enable")

    def prof_collect(self) -> Dict[str, Dict[str, float]]:
        n = len(_lib.PROF_CLASSES)
        ms, cnt, fl, by, mf = (C.c_double * n)(), (C.c_int64 * n)(), (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)()
        _lib.check(self.lib.mp_prof_collect(self.handle, ms, cnt, fl, by, mf), "mp_prof_collect")
        return {name: {"ms": ms[i], "launches": int(cnt[i]), "flops": fl[i], "bytes": by[i], "model_flops": mf[i]} for i, name in enumerate(_lib.PROF_CLASSES)}

    def prof_kinds(self) -> Dict[str, Dict[str, float]]:
        """The Linear GEMM launches of the interval the last prof_collect() closed, by kind ("rot.qkv.fwd", ...: mp_prof_kinds)."""
        n = len(_lib.PROF_KINDS)
        ms, cnt, per, fl, by, mf = (C.c_double * n)(), (C.c_int64 * n)(), (C.c_int64 * n)(), (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)()
        _lib.check(self.lib.mp_prof_kinds(self.handle, ms, cnt, per, fl, by, mf), "mp_prof_kinds")
        return {name: {"ms": ms[i], "launches": int(cnt[i]), "persist_launches": int(per[i]), "flops": fl[i], "bytes": by[i], "model_flops": mf[i]}
                for i, name in enumerate(_lib.PROF_KINDS) if cnt[i] > 0}
