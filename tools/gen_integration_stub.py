"""Regenerates the generated part of INTEGRATION.md section 2 (the ctypes declarations of the C ABI) from the in-tree binding
manipose_amd/_lib.py, so the document cannot drift from include/manipose_hip.h (tests/test_host_cpu.py checks both against the header).
    python tools/gen_integration_stub.py          rewrite INTEGRATION.md in place
    python tools/gen_integration_stub.py --check  exit 1 if the document is out of date"""
import ctypes as C
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BEGIN, END = "<!-- BEGIN GENERATED: ctypes declarations (tools/gen_integration_stub.py) -->", "<!-- END GENERATED -->"
# entry points shown in the document: the path's stand-alone operators and the model engine
SHOWN = ["mp_abi_version", "mp_last_error", "mp_fk_decode_fwd", "mp_fk_decode_bwd", "mp_wta_loss", "mp_aggregate", "mp_mpjpe_sum", "mp_adam_step",
         "mp_split_bf16", "mp_linear_fwd_bf16x3", "mp_linear_fwd_bf16x3_lnres", "mp_split_f16f8", "mp_linear_fwd_f16f8", "mp_attention_fwd_bf16x3", "mp_heads_fwd", "mp_heads_bwd", "mp_model_create", "mp_model_destroy", "mp_model_flat_size",
         "mp_model_num_params", "mp_model_param_info", "mp_model_forward", "mp_model_backward", "mp_model_grad_health", "mp_prof_kinds", "mp_set_option", "mp_pose_metrics", "mp_gather_windows"]


def tname(t):
    from manipose_amd import _lib
    if t is None:
        return "None"
    names = {C.c_void_p: "vp", C.c_int: "i32", C.c_int64: "i64", C.c_float: "f32", C.c_uint64: "u64", C.c_char_p: "C.c_char_p",
             C.c_double: "C.c_double"}
    if t in names:
        return names[t]
    if isinstance(t, type) and issubclass(t, C.Array):
        return f"{tname(t._type_)} * {t._length_}"
    if t is C.POINTER(_lib.ModelConfig):
        return "C.POINTER(ModelConfig)"
    if t is C.POINTER(_lib.LossConfig):
        return "C.POINTER(LossConfig)"
    for base, n in ((C.c_void_p, "vp"), (C.c_int, "i32"), (C.c_int64, "i64"), (C.c_float, "f32"), (C.c_double, "C.c_double")):
        if t is C.POINTER(base):
            return f"C.POINTER({n})"
    raise ValueError(t)


def fields(cls):
    return ", ".join(f'("{n}", {tname(t)})' for n, t in cls._fields_)


def generate():
    from manipose_amd import _lib
    out = [BEGIN, "```python", "import ctypes as C", 'lib = C.CDLL("manipose_amd/libmanipose_hip.so")',
           "vp, i32, i64, f32, u64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint64",
           f"assert lib.mp_abi_version() == {_lib.ABI_VERSION}", "",
           "class LossConfig(C.Structure):      # mp_loss_config", f"    _fields_ = [{fields(_lib.LossConfig)}]", "",
           "class ModelConfig(C.Structure):     # mp_model_config", f"    _fields_ = [{fields(_lib.ModelConfig)}]", ""]
    for name in SHOWN:
        res, args = _lib._SIGNATURES[name]
        out.append(f"lib.{name}.restype = {tname(res)}; lib.{name}.argtypes = [{', '.join(tname(a) for a in args)}]")
    out += ["```", END]
    return "\n".join(out)


def main():
    path = os.path.join(ROOT, "INTEGRATION.md")
    text = open(path).read()
    new = re.sub(re.escape(BEGIN) + r".*?" + re.escape(END), lambda m: generate(), text, flags=re.S)
    if "--check" in sys.argv:
        sys.exit(0 if new == text else 1)
    open(path, "w").write(new)


if __name__ == "__main__":
    main()
