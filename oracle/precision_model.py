#!/usr/bin/env python
"""CPU study (test infrastructure, never shipped or timed): which operand formats keep the Linear layers of the full-size model inside the
north star's 1e-4 m, emulated on the oracle's forward by rounding the operands of every F.linear (mix_ste.py:216-222, 257-281: the qkv / proj /
fc1 / fc2 products, plus the embeddings and heads) and multiplying the rounded values in fp64.  Attention, LayerNorm, the decoder stay fp32.

    python oracle/precision_model.py [windows]

Schemes (x = activation row, W = weight row, both split along the reduction index):
  bf16        x_hi W_hi, hi = bf16(.)                                          (one matrix-core product; BASELINE's "bf16")
  bf16x3      x_hi W_hi + x_lo W_hi + x_hi W_lo, all four planes bf16          (the shipped split precision: three products)
  bf16+fp8    bf16 hi planes, the two correction products on MX e4m3 operands  (block of 32 along the reduction, power-of-two scale)
  fp16+fp8    fp16 hi planes, corrections on MX e4m3                           (one fp16 product + two products at twice the rate = 2)
  fp16+fp8s   the same with STATIC power-of-two scales (x_lo 2^11, W_hi 2^4, x_hi 1, W_lo 2^15), plain e4m3 casts: no scale planes
  fp16+fp6    ... on MX e2m3                                                   (1 + 2/4 = 1.5 products)
  fp16+fp4    ... on MX e2m1
  fp16        x_hi W_hi, hi = fp16(.)
The table it prints is quoted in DESIGN.md section 7 (what comes next); nothing else depends on this file.
"""
import os
import sys
import math

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import manipose_ref as orc  # noqa: E402

FORMATS = {"e4m3": (3, -6, 448.0, 8), "e2m3": (3, 0, 7.5, 2), "e2m1": (1, 0, 6.0, 2)}     # mantissa bits, min exponent, max value, max exponent


def mx_round(v: torch.Tensor, fmt: str) -> torch.Tensor:
    """OCP MX rounding along the last axis: blocks of 32 share a power-of-two scale, elements are rounded to `fmt` (nearest even)."""
    mb, emin, vmax, emax = FORMATS[fmt]
    K = v.shape[-1]
    pad = (-K) % 32
    w = F.pad(v, (0, pad)).reshape(*v.shape[:-1], -1, 32).double()
    amax = w.abs().amax(-1, keepdim=True).clamp_min(1e-300)
    scale = torch.exp2(torch.floor(torch.log2(amax)) - emax)
    u = w / scale
    e = torch.floor(torch.log2(u.abs().clamp_min(1e-300))).clamp_min(emin)
    q = torch.exp2(e - mb)
    r = (torch.round(u / q) * q).clamp(-vmax, vmax)
    return (r * scale).reshape(*v.shape[:-1], -1)[..., :K]


def split(v, hi_dtype):
    hi = v.to(hi_dtype).double()
    return hi, v.double() - hi


def make_linear(scheme: str):
    def linear(x, W, b=None):
        xd, Wd = x.double(), W.double()
        if scheme == "fp32":
            y = xd @ Wd.T
        elif scheme in ("bf16", "fp16"):
            dt = torch.bfloat16 if scheme == "bf16" else torch.float16
            y = x.to(dt).double() @ W.to(dt).double().T
        elif scheme == "bf16x3":
            xh, xl = split(x, torch.bfloat16)
            Wh, Wl = split(W, torch.bfloat16)
            xl, Wl = xl.float().to(torch.bfloat16).double(), Wl.float().to(torch.bfloat16).double()
            y = xh @ Wh.T + xl @ Wh.T + xh @ Wl.T
        elif scheme == "fp16+fp8s":      # static power-of-two scales instead of per-block ones: x_lo 2^11, W_hi 2^4, x_hi 1, W_lo 2^15
            def f8(v, p):
                return (v * 2.0 ** p).float().clamp(-448, 448).to(torch.float8_e4m3fn).double() * 2.0 ** -p
            xh, xl = split(x, torch.float16)
            Wh, Wl = split(W, torch.float16)
            y = xh @ Wh.T + f8(xl, 11) @ f8(Wh, 4).T + f8(xh, 0) @ f8(Wl, 15).T
        else:
            hi, lo = scheme.split("+")
            dt = torch.bfloat16 if hi == "bf16" else torch.float16
            fmt = {"fp8": "e4m3", "fp6": "e2m3", "fp4": "e2m1"}[lo]
            xh, xl = split(x, dt)
            Wh, Wl = split(W, dt)
            y = xh @ Wh.T + mx_round(xl, fmt) @ mx_round(Wh, fmt).T + mx_round(xh, fmt) @ mx_round(Wl, fmt).T
        if b is not None:
            y = y + b.double()
        return y.float()
    return linear


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    cfg = dict(orc.FULL_CFG, T=243, n_hyp=5)
    st = orc.make_state(cfg, seed=0)
    X, _ = orc.synthetic_batch(B, 243, seed=42)
    real = F.linear
    out = {}
    with torch.no_grad():
        for scheme in ("fp32", "bf16", "fp16", "bf16x3", "bf16+fp8", "fp16+fp8", "fp16+fp8s", "fp16+fp6", "fp16+fp4"):
            orc.F.linear = make_linear(scheme)
            try:
                poses, scores = orc.rmcl_manifold_forward(X, st, orc.oracle_cfg(cfg))
            finally:
                orc.F.linear = real
            out[scheme] = poses.double()
            if scheme != "fp32":
                err = (out[scheme] - out["fp32"]).norm(dim=-1)
                print(f"{scheme:10s} MPJPE vs fp64-product forward {err.mean().item():.3e} m   p99.9 {err.flatten().kthvalue(int(0.999 * err.numel())).values.item():.3e}"
                      f"   max {err.max().item():.3e}", flush=True)


if __name__ == "__main__":
    main()
