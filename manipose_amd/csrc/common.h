// Shared helpers for the ManiPose gfx950 kernels (internal; the public surface is include/manipose_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>

namespace mp {

// ---- error plumbing -------------------------------------------------------------------------
void set_error(const char* fmt, ...);
#define MP_CHECK(cond, code, ...)                 \
  do {                                            \
    if (!(cond)) {                                \
      mp::set_error(__VA_ARGS__);                 \
      return (code);                              \
    }                                             \
  } while (0)
#define MP_HIP(expr)                                                                         \
  do {                                                                                       \
    hipError_t e__ = (expr);                                                                 \
    if (e__ != hipSuccess) {                                                                 \
      mp::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
      return MP_ERR_HIP;                                                                     \
    }                                                                                        \
  } while (0)
#define MP_LAUNCH_CHECK() MP_HIP(hipGetLastError())

enum { MP_OK = 0, MP_ERR_ARG = 1, MP_ERR_HIP = 2, MP_ERR_STATE = 3 };

__host__ __device__ static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- activation storage types ---------------------------------------------------------------
typedef __hip_bfloat16 bf16;

__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(bf16 v) { return __bfloat162float(v); }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f<bf16>(float v) { return __float2bfloat16(v); }

// 4-wide vector load/store of activations as floats (16 B for f32, 8 B for bf16)
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 ld4(const bf16* p) {
  uint2 r = *reinterpret_cast<const uint2*>(p);
  float4 v;
  v.x = __uint_as_float(r.x << 16);
  v.y = __uint_as_float(r.x & 0xffff0000u);
  v.z = __uint_as_float(r.y << 16);
  v.w = __uint_as_float(r.y & 0xffff0000u);
  return v;
}
// two fp32 -> one dword of two bf16 (round to nearest even): a single v_cvt_pk_bf16_f32 (the scalar __float2bfloat16 route costs
// two conversions, a shift and an or per pair)
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {lo, hi};
  const bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
  return *reinterpret_cast<const unsigned*>(&r);
}
__device__ __forceinline__ void st4(bf16* p, float4 v) {
  uint2 r;
  r.x = pack_bf16x2(v.x, v.y);
  r.y = pack_bf16x2(v.z, v.w);
  *reinterpret_cast<uint2*>(p) = r;
}

// ---- split precision ("bf16x3"): a value is carried as TWO bf16 numbers hi = bf16(x), lo = bf16(x - hi) (16 significand bits
// together) stored in two planes of the same shape; a product a b is evaluated as a_lo b_hi + a_hi b_lo + a_hi b_hi on the bf16
// matrix cores with fp32 accumulation (the dropped a_lo b_lo term is 2^-16 relative).  The hi plane alone IS the bf16 tensor, so
// bf16 kernels can read a planar buffer unchanged.  `bf16p` tags a pointer to the hi plane of such a pair.
struct bf16p { unsigned short v; };
__device__ __forceinline__ void split_bf16x2(float a, float b, unsigned& hi, unsigned& lo) {
  // What is split is the ROUNDED fp32 value.  Without the opaque move hipcc may fuse the multiply that produced it into the subtraction below
  // (fma(x, y, -hi): the unrounded product) in some unrolled copies of an epilogue and not in others - round 5 found the lo plane of the GELU
  // epilogue one ulp different in the last 4 of a wave's 128 rows, i.e. a token's result depending on its row position (tools/probes/half_batch_ops.py)
  asm("" : "+v"(a), "+v"(b));
  hi = pack_bf16x2(a, b);
  lo = pack_bf16x2(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}
// lo_off: element offset from the hi plane to the lo plane
__device__ __forceinline__ void st4(bf16p* p, float4 v, long lo_off) {
  uint2 h, l;
  split_bf16x2(v.x, v.y, h.x, l.x);
  split_bf16x2(v.z, v.w, h.y, l.y);
  *reinterpret_cast<uint2*>(p) = h;
  *reinterpret_cast<uint2*>(p + lo_off) = l;
}
__device__ __forceinline__ void st4(float* p, float4 v, long) { st4(p, v); }
__device__ __forceinline__ void st4(bf16* p, float4 v, long) { st4(p, v); }

// ---- "f16f8" operand format (include/manipose_hip.h, mp_linear_fwd_f16f8; gemm_bf16.hip, mma_stage_mix): hi = fp16(v) in a plane of 2-byte
// elements plus a correction plane of the same byte geometry - 2 bytes per element, the 8 bytes of elements 4 q .. 4 q + 3 of a row being
//   activation:  4 x e4m3(2^11 (v - hi)) | 4 x e4m3(hi)            weight:  4 x e4m3(2^4 hi) | 4 x e4m3(2^15 (v - hi))
// (groups of four so that a thread holding four consecutive channels writes ONE 8-byte store and a wave whole 128-byte lines; the matrix
// instruction only needs both operands to use the same byte order).  `f16f8` tags a pointer to the fp16 plane.  Values are clamped to the
// e4m3 range (+-448) ahead of the conversion.
struct f16f8 { unsigned short v; };
// GEMM epilogue form: p = the element in the fp16 plane, lo_off = distance to the correction plane in 2-byte elements (both planes have 2 bytes per element)
__device__ __forceinline__ void st4_f16f8(f16f8* hi16, char* corr8, float4 v, bool weight);
__device__ __forceinline__ void st4(f16f8* p, float4 v, long lo_off) { st4_f16f8(p, reinterpret_cast<char*>(p + lo_off), v, false); }
// Four gradient values as SATURATING fp16 (a gradient operand carried as fp16 of S x value: kernels.h GemmB16Args::f16 / gout).  fp16 ends at
// 65504 and S is chosen from the loss gradient only (grad_scale, elementwise.hip), so an interior gradient far above it must not become inf -
// inf in one operand element is NaN in a whole weight gradient, and Adam spreads that over every weight.  Values beyond +-65504 are clamped,
// non-finite ones are written as 0, and both are counted in cnt[0] / cnt[1] (mp_model::gsc + 4; mp_model_grad_health) - on a path that
// costs three max, three adds and a compare per four elements while nothing saturates.
__device__ __forceinline__ uint2 sat_f16x4(float a, float b, float c, float d, unsigned* __restrict__ cnt) {
  typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
  const float m = fmaxf(fmaxf(fabsf(a), fabsf(b)), fmaxf(fabsf(c), fabsf(d)));      // (fmaxf drops a NaN operand: the sum below catches those)
  const float t = (a + b) + (c + d);
  if (__builtin_expect(!(m <= 65504.f) || t != t, 0)) {
    unsigned ns = 0, nn = 0;
    float* const v[4] = {&a, &b, &c, &d};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float x = *v[i];
      if (!(fabsf(x) <= 3.0e38f)) { x = 0.f; ++nn; }                  // NaN or inf
      else if (fabsf(x) > 65504.f) { x = copysignf(65504.f, x); ++ns; }
      *v[i] = x;
    }
    if (cnt != nullptr) {
      if (ns) atomicAdd(cnt, ns);
      if (nn) atomicAdd(cnt + 1, nn);
    }
  }
  const h4_t h = {(_Float16)a, (_Float16)b, (_Float16)c, (_Float16)d};
  return __builtin_bit_cast(uint2, h);
}
__device__ __forceinline__ void st4_f16(void* p, float4 v, float s, unsigned* __restrict__ cnt) {
  *reinterpret_cast<uint2*>(p) = sat_f16x4(v.x * s, v.y * s, v.z * s, v.w * s, cnt);
}
__device__ __forceinline__ unsigned pack_e4m3x4(float a, float b, float c, float d) {
  a = __builtin_amdgcn_fmed3f(a, -448.f, 448.f); b = __builtin_amdgcn_fmed3f(b, -448.f, 448.f);
  c = __builtin_amdgcn_fmed3f(c, -448.f, 448.f); d = __builtin_amdgcn_fmed3f(d, -448.f, 448.f);
  int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);       // bytes 0, 1
  r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);            // bytes 2, 3
  return (unsigned)r;
}
// four consecutive elements v (columns c .. c + 3, c % 4 == 0) of a row: hi16 = their place in the fp16 plane, corr8 = their 8 bytes in the
// correction plane (byte offset 2 (row K + c))
// the same in registers, activation form: h16 = the four fp16 values, c8 = their 8 correction bytes.  20 VALU instructions on the common path
// (round 6; 30 before): the pairs go through v_cvt_pk_f16_f32, the 2^11 on the lo parts is the scale operand of v_cvt_scalef32_pk_fp8_f32 (it divides by
// 2^floor(log2 s): tools/probes/cvt_scale.hip), and the +-448 clamps - both fp8 conversions return NaN (0x7f) beyond that, neither saturates - are only
// taken by lanes that hold a value above 448 (|2^11 lo| <= |v|, so nothing can overflow below it).  Same bytes as the clamped form for every input.
__device__ __forceinline__ void pack4_f16f8(float4 v, uint2& h16, uint2& c8) {
  typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
  typedef float f2_t __attribute__((ext_vector_type(2)));
  typedef short s2_t __attribute__((ext_vector_type(2)));
  asm("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));      // split the ROUNDED values (see split_bf16x2: no fusing of a producer's multiply into the subtraction)
  const f2_t a = {v.x, v.y}, b = {v.z, v.w};
  const h2_t ha = __builtin_convertvector(a, h2_t), hb = __builtin_convertvector(b, h2_t);
  h16 = make_uint2(__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb));
  const float4 hf = make_float4((float)ha[0], (float)ha[1], (float)hb[0], (float)hb[1]);
  const float4 lo = make_float4(v.x - hf.x, v.y - hf.y, v.z - hf.z, v.w - hf.w);
  const float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
  if (__builtin_expect(!(m <= 448.f), 0)) {      // (also NaN / inf inputs: the clamped conversions below)
    c8 = make_uint2(pack_e4m3x4(lo.x * 2048.f, lo.y * 2048.f, lo.z * 2048.f, lo.w * 2048.f), pack_e4m3x4(hf.x, hf.y, hf.z, hf.w));
  } else {
    s2_t r = {0, 0};
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, lo.x, lo.y, 0x1p-11f, false);
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, lo.z, lo.w, 0x1p-11f, true);
    int q = __builtin_amdgcn_cvt_pk_fp8_f32(hf.x, hf.y, 0, false);
    q = __builtin_amdgcn_cvt_pk_fp8_f32(hf.z, hf.w, q, true);
    c8 = make_uint2(__builtin_bit_cast(unsigned, r), (unsigned)q);
  }
}
// four fp16 values (the fp16 plane of an f16f8 activation) -> four bf16 values, round to nearest even: what a bf16 backward kernel makes of an
// operand that exists as an fp16 plane only (mp_model_config::f16f8 = 3: no bf16 copy of a1 / ao / a2 / f is written)
__device__ __forceinline__ uint2 f16x4_to_bf16x4(uint2 h) {
  typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
  const h4_t x = __builtin_bit_cast(h4_t, h);
  return make_uint2(pack_bf16x2((float)x[0], (float)x[1]), pack_bf16x2((float)x[2], (float)x[3]));
}
__device__ __forceinline__ void st4_f16f8(f16f8* hi16, char* corr8, float4 v, bool weight) {
  if (!weight) {      // (a compile-time constant at every call site) activation form: the register form above
    uint2 h, c;
    pack4_f16f8(v, h, c);
    *reinterpret_cast<uint2*>(hi16) = h;
    *reinterpret_cast<uint2*>(corr8) = c;
    return;
  }
  typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
  const h4_t h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
  *reinterpret_cast<uint2*>(hi16) = __builtin_bit_cast(uint2, h);
  const float4 hf = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
  const float4 lo = make_float4(v.x - hf.x, v.y - hf.y, v.z - hf.z, v.w - hf.w);
  const float s1 = weight ? 16.f : 2048.f, s2 = weight ? 32768.f : 1.f;
  const float4 f1 = weight ? hf : lo, f2 = weight ? lo : hf;
  *reinterpret_cast<uint2*>(corr8) = make_uint2(pack_e4m3x4(f1.x * s1, f1.y * s1, f1.z * s1, f1.w * s1), pack_e4m3x4(f2.x * s2, f2.y * s2, f2.z * s2, f2.w * s2));
}
__device__ __forceinline__ float4 ld4(const bf16p* p, long lo_off) {
  const float4 h = ld4(reinterpret_cast<const bf16*>(p)), l = ld4(reinterpret_cast<const bf16*>(p + lo_off));
  return make_float4(h.x + l.x, h.y + l.y, h.z + l.z, h.w + l.w);
}

// ---- packed-fp32 guard (the cause, found in round 5) -----------------------------------------------
// Round 3 (tools/gemm_determinism.py, tools/step_determinism.py, tools/batch_invariance.py) found two wrong-result defects - ~1e-4 of the rows
// of the tiled GEMM's recomputed-LayerNorm residual epilogue with one float per lane wrong (2 mm errors on the segment lengths at the
// benchmark's batch), and non-reproducible LayerNorm d gamma - different in every run, more often with other streams busy, lanes 48-63.  Both
// sites had in common a compiler-made (SLP) packed fp32 op (v_pk_mul/add/fma_f32) whose LOW lane takes the HIGH register of a VGPR pair
// (op_sel) that a global_load_dwordx2 had written; both disappeared with any source change that removed that operand form.  Round 4's probe
// (tools/probes/pk_opsel.hip, modes 0-3) ran the form and the whole faulty instruction sequence alone, found 0 mismatches in 8.6e9
// lane-operations and withdrew the diagnosis.  That probe lacked one condition.  Round 5 (profiles/r05_defect_isa/: the ISA of the five
// commits around the fixes - no s_waitcnt is missing in the faulty code - and tools/probes/pk_mfma.hip) reproduces the defect in isolation:
//   a packed fp32 op with op_sel routing the HIGH register of a pair to its LOW lane takes ZERO for that operand in lanes 48-63, some of the
//   time (up to 54 % of the rows of that lane quarter), if the pair was last written by a global_load_dwordx2 AND another wave of the same
//   SIMD is issuing MFMAs.  Never without the MFMA neighbour; never in lanes 0-47 or in the high lane; idle cycles behind the s_waitcnt do
//   not help (64 tried); a VALU copy of the pair, two dword loads, an LDS read of the pair, the mirrored selection, v_pk_mov_b32 and the
//   unselected packed forms are all clean (17 variants x 2.7e8 rows, profiles/r05_defect_isa/README.md).  How often it strikes depends on code
//   outside the sequence (the same asm block: 39.8 M wrong values in one build of the probe, none in the next) - hence "every source change
//   fixes it".  The epilogue ran beside other workgroups' k loops on its SIMD, the LayerNorm backward beside the GEMMs of the other streams.
// Whether that is an erratum of gfx950 or a hazard the compiler should cover cannot be told from here; clang 20 / ROCm 7.2 emits the sequence.
// What the library does about it:
//  * no v_pk_*_f32 and no v_pk_mov_b32 exists in the device code (build.sh: -fno-slp-vectorize and -target-feature -packed-fp32-ops; the
//    vector-typed GELU / softmax source below compiles to plain v_mul / v_fma).  This costs nothing on gfx950: a wave64 v_fma_f32 already issues
//    at the SIMD's full 32 lanes per clock, the packed forms are not faster (MI355X_MICROARCH: "+22 cycles vs two v_fma_f32" beside MFMAs;
//    same-box A/B of the whole library in round 3: 171.6 / 170.9 -> 170.2 / 169.9 ms per step without them);
//  * tools/scan_pk_opsel.py audits the generated code of every kernel for the operand form and fails if a source does not compile (CPU test);
//  * the defects themselves are what the GPU suite watches for, at the scale where they showed: two identical training steps must give
//    identical bits for every output and all 34.4 M gradient values, a window's forward must not depend on its batch, and the split-precision
//    Linear kernels must reproduce their bits over 8 runs under load at the benchmark's token count (tests/test_gpu_parity.py:
//    test_training_step_is_bitwise_reproducible_and_batch_invariant, test_split_precision_linear_kernels_are_reproducible_at_scale).
// lone(): additionally gives a memory-loaded scalar that is multiplied into several values a VALU-written register of its own.
__device__ __forceinline__ float lone(float v) {
  float r;
  asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(v));
  return r;
}
__device__ __forceinline__ float4 lone4(const float4& v) { return make_float4(lone(v.x), lone(v.y), lone(v.z), lone(v.w)); }

// ---- wave (64-lane) reductions ----------------------------------------------------------------
// DPP cross-lane operands (no LDS crossbar: a __shfl_xor butterfly is six dependent ds_bpermute_b32, ~100 clocks each):
// quad_perm [1,0,3,2] and [2,3,0,1], row_half_mirror, row_mirror leave every lane with the result over its row of 16;
// row_bcast:15 / row_bcast:31 carry it into lane 63, which is read back as a wave-uniform scalar.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float old, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f<0xB1, 0xf>(0.f, v);
  v += dpp_f<0x4E, 0xf>(0.f, v);
  v += dpp_f<0x141, 0xf>(0.f, v);
  v += dpp_f<0x140, 0xf>(0.f, v);
  v += dpp_f<0x142, 0xa>(0.f, v);       // rows 1, 3 += last lane of rows 0, 2
  v += dpp_f<0x143, 0xc>(0.f, v);       // rows 2, 3 += lane 31
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  const float ninf = -__builtin_inff();
  v = fmaxf(v, dpp_f<0xB1, 0xf>(ninf, v));
  v = fmaxf(v, dpp_f<0x4E, 0xf>(ninf, v));
  v = fmaxf(v, dpp_f<0x141, 0xf>(ninf, v));
  v = fmaxf(v, dpp_f<0x140, 0xf>(ninf, v));
  v = fmaxf(v, dpp_f<0x142, 0xa>(ninf, v));
  v = fmaxf(v, dpp_f<0x143, 0xc>(ninf, v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// exact (erf) GELU and its derivative, as torch.nn.GELU() (mix_ste.py:297 act_layer=nn.GELU)
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// fast GELU for the bf16 throughput mode: erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, far below bf16 resolution);
// the derivative shares the exponential.  exp(-x^2/2) is computed once.
__device__ __forceinline__ void gelu_parts_fast(float x, float& cdf, float& pdf) {
  const float ax = fabsf(x) * 0.70710678118654752440f;
  const float t = __frcp_rn(1.0f + 0.3275911f * ax);
  const float e = __expf(-ax * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erf_abs = 1.0f - poly * e;
  cdf = 0.5f * (1.0f + copysignf(erf_abs, x));
  pdf = 0.39894228040143267794f * e;
}
// Two elements at a time on the packed-fp32 VALU (v_pk_mul_f32 / v_pk_fma_f32): the GELU epilogue of the fc1 GEMM is bound by
// VALU issue (about 20 instructions per element in the scalar form), not by HBM.  Same formula as gelu_parts_fast; the
// reciprocal is the 1-ulp hardware v_rcp_f32.  Returns f = x cdf and f' = cdf + x pdf.
typedef float mp_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_pair_fast(float x0, float x1, float& f0, float& f1, float& d0, float& d1) {
  const mp_f32x2 x = {x0, x1};
  const mp_f32x2 ax = {fabsf(x0), fabsf(x1)};
  const mp_f32x2 den = ax * (0.3275911f * 0.70710678118654752440f) + 1.0f;
  const mp_f32x2 t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
  const mp_f32x2 arg = (x * -0.72134752044448170368f) * x;                 // -(x^2 / 2) log2 e
  const mp_f32x2 e = {__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])};
  mp_f32x2 poly = t * 1.061405429f + -1.453152027f;
  poly = poly * t + 1.421413741f;
  poly = poly * t + -0.284496736f;
  poly = poly * t + 0.254829592f;
  poly = poly * t;
  const mp_f32x2 erf_abs = 1.0f - poly * e;
  const mp_f32x2 erf_s = {copysignf(erf_abs[0], x0), copysignf(erf_abs[1], x1)};
  const mp_f32x2 cdf = erf_s * 0.5f + 0.5f;
  const mp_f32x2 pdf = e * 0.39894228040143267794f;
  const mp_f32x2 f = x * cdf, d = x * pdf + cdf;
  f0 = f[0]; f1 = f[1]; d0 = d[0]; d1 = d[1];
}
// float4 form used by the GEMM epilogues: v <- gelu(v), returns gelu'(v)
__device__ __forceinline__ float4 gelu_fwd4_fast(float4& v) {
  float4 d;
  gelu_pair_fast(v.x, v.y, v.x, v.y, d.x, d.y);
  gelu_pair_fast(v.z, v.w, v.z, v.w, d.z, d.w);
  return d;
}
__device__ __forceinline__ float gelu_fast(float x) {
  float c, p;
  gelu_parts_fast(x, c, p);
  return x * c;
}
__device__ __forceinline__ float gelu_grad_fast(float x) {
  float c, p;
  gelu_parts_fast(x, c, p);
  return c + x * p;
}

// DropPath row-group index of token row m (layout (b,t,j), m = (b*T + t)*J + j):
// mode 1 = spatial block: sample = (b,t) = m / J ; mode 2 = temporal block: sample = (b,j)
__device__ __forceinline__ float droppath_scale(const float* mask, int mode, int m, int T, int J) {
  if (mask == nullptr || mode == 0) return 1.0f;
  if (mode == 1) return mask[m / J];
  const int b = m / (T * J);
  return mask[b * J + (m % J)];
}

// The same map for the rows of one GEMM tile, without a per-row integer division (~30 VALU instructions each, which dominated
// the residual epilogues): m = b0 TJ + r with b0, r0 divided out ONCE from the wave-uniform first row; per row only the small
// r = r0 + (m - row0) remains, whose quotients by J and TJ are exact in fp32 ((r + 0.5) / d is at least 0.5 / d away from an
// integer and r < 2^20).  TJ is a multiple of J, so m / J = b0 T + r / J and m % J = r % J.
struct DropPathRows {
  const float* mk;
  int mode, T, J, TJ, b0, r0, row0;
  float invJ, invTJ;
  __device__ __forceinline__ void init(const float* mask, int mask_mode, int T_, int J_, int row0_uniform) {
    mk = (mask != nullptr && mask_mode != 0) ? mask : nullptr;
    mode = mask_mode; T = T_; J = J_; TJ = T_ * J_;
    row0 = __builtin_amdgcn_readfirstlane(row0_uniform);
    b0 = mk != nullptr ? row0 / TJ : 0;
    r0 = mk != nullptr ? row0 - b0 * TJ : 0;
    invJ = 1.0f / (float)J;
    invTJ = 1.0f / (float)TJ;
  }
  __device__ __forceinline__ float scale(int m) const {      // m >= row0, m - row0 < 2^19
    if (mk == nullptr) return 1.0f;
    const int r = r0 + (m - row0);
    const int q = (int)(((float)r + 0.5f) * invJ);
    if (mode == 1) return mk[b0 * T + q];
    const int qb = (int)(((float)r + 0.5f) * invTJ);
    return mk[(b0 + qb) * J + (r - q * J)];
  }
};

}  // namespace mp
