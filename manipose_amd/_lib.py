"""ctypes binding of libmanipose_hip.so (C ABI: include/manipose_hip.h).

There is deliberately NO fallback: if the HIP library is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import re
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# MANIPOSE_HIP_LIB: another build of the same library (A/B timing of two builds on one box)
LIB_PATH = os.environ.get("MANIPOSE_HIP_LIB") or os.path.join(_HERE, "libmanipose_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "manipose_hip.h")

_lib: Optional[C.CDLL] = None
ABI_VERSION = 8

vp, i32, i64, f32, u64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint64


class LossConfig(C.Structure):
    """mp_loss_config; defaults = hpe/conf/config.yaml:32-38 of the reference."""
    _fields_ = [("rmcl_score_reg", f32), ("vel_loss", f32), ("smooth_reg", f32), ("w_loss", i32), ("sq_loss", i32), ("joint_weights", f32 * 17)]


class ModelConfig(C.Structure):
    """mp_model_config."""
    _fields_ = [("arch", i32), ("num_frame", i32), ("num_joints", i32), ("num_bones", i32),
                ("embed_dim_rot", i32), ("depth_rot", i32), ("num_heads_rot", i32),
                ("embed_dim_seg", i32), ("depth_seg", i32), ("num_heads_seg", i32),
                ("n_hyp", i32), ("drop_path_rate", f32), ("max_batch", i32), ("precision", i32), ("rot_rep_dim", i32),
                ("qk_scale_rot", f32), ("resid_scale_rot", f32), ("readout_mult_rot", f32),
                ("qk_scale_seg", f32), ("resid_scale_seg", f32), ("readout_mult_seg", f32),
                ("f16f8", i32), ("f16_backward", i32), ("streams", i32), ("debug", i32)]


_SIGNATURES = {
    "mp_abi_version": (i32, []),
    "mp_last_error": (C.c_char_p, []),
    "mp_fk_decode_fwd": (i32, [vp, i32, i32, vp, vp, i32, i32, i32, vp]),
    "mp_fk_decode_bwd": (i32, [vp, i32, i32, vp, vp, vp, vp, i32, i32, i32, vp]),
    "mp_wta_loss": (i32, [vp, vp, vp, C.POINTER(LossConfig), vp, vp, vp, vp, i32, i32, i32, vp, i64, vp]),
    "mp_single_loss": (i32, [vp, vp, C.POINTER(LossConfig), vp, vp, i32, i32, vp, i64, vp]),
    "mp_rigid_segments_loss": (i32, [vp, f32, vp, vp, i32, i32, vp, i64, vp]),
    "mp_aggregate": (i32, [vp, vp, vp, i32, vp, i32, i32, i32, vp]),
    "mp_mpjpe_sum": (i32, [vp, vp, i64, vp, vp, i64, vp]),
    "mp_adam_step": (i32, [vp, vp, vp, vp, i64, i32, f32, f32, f32, f32, f32, f32, vp]),
    "mp_adam_step_scaled": (i32, [vp, vp, vp, vp, i64, i32, f32, f32, f32, f32, f32, f32, vp, vp, vp]),
    "mp_layernorm_fwd": (i32, [vp, vp, vp, f32, vp, vp, i32, i32, vp]),
    "mp_layernorm_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, i64, vp]),
    "mp_linear_fwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "mp_linear_bwd_slab_floats": (i64, [i32, i32]),
    "mp_linear_bwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, i64, vp]),
    "mp_linear_fwd_bf16": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "mp_linear_bwd_bf16": (i32, [vp, i32, vp, vp, vp, i32, vp, vp, i32, i32, i32, vp, i64, vp]),
    "mp_attention_fwd": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "mp_attention_bwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "mp_attention_fwd_bf16": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "mp_attention_bwd_bf16": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "mp_heads_fold_floats": (i64, [i32]),
    "mp_heads_bwd_scratch_floats": (i64, [i32, i32, i32]),
    "mp_heads_fwd": (i32, [vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, i32, i32, i32, vp]),
    "mp_heads_bwd": (i32, [vp] * 14 + [i32, i32, i32, i32, i32, vp, i64, vp]),
    "mp_split_bf16": (i32, [vp, vp, vp, i64, vp]),
    "mp_linear_fwd_bf16x3": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "mp_linear_fwd_bf16x3_lnres": (i32, [vp] * 11 + [i32] * 6 + [vp]),
    "mp_linear_fwd_f16f8": (i32, [vp] * 6 + [i32] * 3 + [vp]),
    "mp_split_f16f8": (i32, [vp, vp, vp, i64, i32, vp]),
    "mp_attention_fwd_bf16x3": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "mp_model_create": (i32, [C.POINTER(ModelConfig), C.POINTER(vp)]),
    "mp_model_destroy": (None, [vp]),
    "mp_model_workspace_bytes": (i64, [vp]),
    "mp_model_num_params": (i32, [vp]),
    "mp_model_flat_size": (i64, [vp]),
    "mp_model_param_info": (i32, [vp, i32, C.c_char_p, i32, C.POINTER(i64), C.POINTER(i64)]),
    "mp_model_num_mask_branches": (i32, [vp]),
    "mp_model_mask_info": (i32, [vp, i32, i32, C.c_char_p, i32, C.POINTER(i64), C.POINTER(i64), C.POINTER(f32)]),
    "mp_model_mask_floats": (i64, [vp, i32]),
    "mp_model_forward": (i32, [vp, vp, vp, i32, vp, vp, i32, vp, u64, u64, vp]),
    "mp_model_backward": (i32, [vp, vp, vp, vp, vp, vp]),
    "mp_model_grad_bucket_count": (i32, [vp]),
    "mp_model_grad_bucket_info": (i32, [vp, i32, C.POINTER(i64), C.POINTER(i64)]),
    "mp_model_grad_bucket_wait": (i32, [vp, i32, vp]),
    "mp_model_grad_health": (i32, [vp, C.POINTER(f32), vp]),
    "mp_model_grad_health_async": (i32, [vp, vp, vp]),
    "mp_model_set_streams": (i32, [vp, i32]),
    "mp_model_hazard_report": (i32, [vp, C.POINTER(i64), C.c_char_p, i32]),
    "mp_hazard_create": (vp, []),
    "mp_hazard_destroy": (None, [vp]),
    "mp_hazard_launch": (i32, [vp, i32, C.c_char_p, i32, C.POINTER(i64), C.POINTER(i64), C.POINTER(i32)]),
    "mp_hazard_record": (i32, [vp, i32, i32]),
    "mp_hazard_wait": (i32, [vp, i32, i32]),
    "mp_hazard_report": (i32, [vp, C.POINTER(i64), C.c_char_p, i32]),
    "mp_model_peek": (i32, [vp, i32, C.POINTER(vp), C.POINTER(i64)]),
    "mp_model_peek_copy": (i32, [vp, i32, vp, i64, vp]),
    "mp_gather_windows": (i32, [vp, vp, vp, i32, vp, vp, vp, C.POINTER(i32), vp, vp, i32, i32, i32, vp, vp, vp]),
    "mp_ingest_pose3d": (i32, [vp, i32, vp, i64, C.POINTER(i32), i32, C.POINTER(f32), C.POINTER(f32), i32, i32, f32, vp, vp]),
    "mp_ingest_pose2d": (i32, [vp, i32, i32, vp, i64, C.POINTER(i32), i32, f32, f32, vp, vp]),
    "mp_procrustes_errors": (i32, [vp, vp, vp, i64, i32, f32, f32, f32, f32, i32, vp, vp, i64, vp]),
    "mp_pose_metrics_row_floats": (i32, []),
    "mp_bone_length_table": (i32, [vp, C.POINTER(i64), vp, C.POINTER(i64), i32, i32, i32, vp, vp]),
    "mp_pose_metrics": (i32, [vp, C.POINTER(i64), vp, C.POINTER(i64), vp, i32, i32, i32, f32, f32, f32, f32, i32, i32, vp, vp, vp, i64, vp]),
    "mp_set_option": (i32, [C.c_char_p, i32]),
    "mp_prof_enable": (i32, [vp, i32]),
    "mp_prof_kinds": (i32, [vp, C.POINTER(C.c_double), C.POINTER(i64), C.POINTER(i64), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "mp_prof_collect": (i32, [vp, C.POINTER(C.c_double), C.POINTER(i64), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}
PROF_KINDS = tuple(f"{mod}.{lay}.{d}" for mod in ("rot", "seg") for d in ("fwd", "dgrad", "wgrad") for lay in ("qkv", "proj", "fc1", "fc2"))   # index = mp_prof_kinds kind
PROF_CLASSES = ("gemm_fwd", "gemm_dgrad", "gemm_wgrad", "attention", "layernorm", "other", "gemm_persist")   # last: subset of the first two


def declared_symbols(header: str = HEADER_PATH):
    """Names of every function include/manipose_hip.h declares (used by the no-GPU export test)."""
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mp_[a-z0-9_]+)\s*\(", text)))


def load() -> C.CDLL:
    """Load the HIP library; raises RuntimeError (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"manipose_amd: {LIB_PATH} is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or manipose_amd/csrc/build.sh (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.mp_abi_version() != ABI_VERSION:
        raise RuntimeError(f"manipose_amd: ABI version {lib.mp_abi_version()} != {ABI_VERSION}; rebuild the library")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().mp_last_error()
        raise RuntimeError(f"manipose_amd: {what} failed (code {rc}): {msg.decode() if msg else '?'}")


def ptr(t) -> Optional[int]:
    """Device pointer of a contiguous fp32/int32 CUDA tensor (None passes NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("manipose_amd: HIP kernels need tensors on a ROCm device (got a CPU tensor); "
                           "there is no CPU fallback")
    if not t.is_contiguous():
        raise RuntimeError("manipose_amd: tensor must be contiguous")
    return t.data_ptr()


def stream_ptr() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream
