// Temporal attention of the MixSTE blocks on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16), forward and backward,
// for windows of T <= 256 frames (H36M: 243, 3DHP: 81/27) and head dims 64 (rotations net) / 16 (bones net).
//
// One workgroup per (window b, joint j, head h): its K/V (and in the backward Q/dO) rows - strided by J*3C in HBM,
// 128 B contiguous each - are staged ONCE into LDS in their natural [t][d] layout (rows padded by 16 B), and every
// matrix product of the layer reads them from there:
//   * products reducing over d (S = Q K^T, dP = dO V^T) take both operands as 16-byte row reads (ds_read_b128);
//   * products reducing over t (O = P V, dQ = dS K, dV = P^T dO, dK = dS^T Q) take the LDS operand through the
//     gfx950 hardware transpose read (ds_read_b64_tr_b16) and the P / dS operand STRAIGHT FROM THE ACCUMULATORS of the
//     previous product: the score tile is computed in the orientation whose accumulator layout (column on the lane,
//     4 rows in registers) is the B-fragment layout of the next MFMA up to a fixed permutation of the reduction index,
//     which is applied to the other operand's transpose-read addresses instead (no LDS round trip, no shuffles).
// Softmax is exact (not online): a wave holds a 16-query x 256-key score strip in 64 accumulator registers.
// fp32 accumulation, fp32 softmax statistics; log-sum-exp saved for the backward.
#include <stdlib.h>
#include "common.h"
#include "kernels.h"

namespace mp {

typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef short bf16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TP = 256;            // padded frames
constexpr int NTILE = TP / 16;

// Single-lane fp32 arithmetic for the softmax that is interleaved with MFMAs: written component by component so that it does not
// compile to v_pk_*_f32 (a packed fp32 instruction beside the matrix cores costs ~20 cycles more than the two plain ones it replaces:
// MI355X_MICROARCH.md, "price of one filler beside MFMAs").  Plain C, not inline asm: hipcc inserts the wait states an MFMA result needs
// before a VALU instruction reads it only for instructions it knows (an asm v_fma_f32 on an accumulator read stale data).
__device__ __forceinline__ float sfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// (contraction off INSIDE the helpers: the flag travels with the instruction when it is inlined, so a product and a sum written through them are never
// fused into one rounding, whatever ends up in one basic block - the kernels of this file stay bit-compatible with one another)
__device__ __forceinline__ float sadd(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float smul(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float ssub(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}

__device__ __forceinline__ float4 bf16x4_to_float4(const uint2& r) {
  return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u));
}

template <int D> struct ACfg {
  // LDS bytes per frame row.  Head dim 64: exactly 128 B, the 16-byte chunk index XOR-swizzled by (frame & 7) - conflict-free for
  // both the row reads (ds_read_b128) and the transpose reads (ds_read_b64_tr_b16); a 16-byte pad (144 B rows) made both of them
  // 2-way conflicted (45 % of the LDS cycles were conflict cycles).  Head dim 16: 32 B + 16 B pad.
  static constexpr int ROWB = D == 64 ? 128 : 2 * D + 16;
  static __device__ __forceinline__ int swz(int row) { return D == 64 ? (row & 7) : 0; }
  static constexpr int KS = (D + 31) / 32;             // k-steps of the d-reductions
  static constexpr int DB = D / 16;                    // 16-wide d blocks of the t-reductions
  static constexpr int CH = D / 8;                     // 16-byte chunks per row
};

// stage `rows` (frames) x D of a strided bf16 matrix into LDS, zero-filling frames >= T
// rows: frames of the image (T rounded up to 32: the products reduce over 32-frame blocks; frames >= T are zero-filled)
// NIMG images at once: img[i] <- rows of base[i] (row strides stride[i]).  Every load of a pass (2 chunks per thread and image) is issued
// before the first LDS write, from an address that is always valid (frame index clamped to T - 1, zero-filled at the write): the
// chunk-at-a-time form - `if (t < T) x = load; write x` - compiled to load / s_waitcnt vmcnt(0) / ds_write per chunk, i.e. one memory round
// trip after the other: 8 of them for the four images of the backward, 8.8 us of a unit's 26 us (round 3, from the ISA).
template <int D, int NIMG>
__device__ __forceinline__ void stage_rows_multi(char* const (&S)[NIMG], const bf16* const (&base)[NIMG], const long (&stride)[NIMG], int T, int rows,
                                                 int tid, int nthreads) {
  constexpr int CH = ACfg<D>::CH, ROWB = ACfg<D>::ROWB, U = 2;
  const int total = rows * CH;
  for (int idx0 = tid; idx0 < total; idx0 += U * nthreads) {
    uint4 x[NIMG][U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = min(idx0 + u * nthreads, total - 1);
      const int t = idx / CH, c = idx - t * CH, tc = min(t, T - 1);
#pragma unroll
      for (int im = 0; im < NIMG; ++im) x[im][u] = *reinterpret_cast<const uint4*>(base[im] + (long)tc * stride[im] + c * 8);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = idx0 + u * nthreads;
      const int t = idx / CH, c = idx - t * CH;
      if (idx < total) {
#pragma unroll
        for (int im = 0; im < NIMG; ++im)
          *reinterpret_cast<uint4*>(S[im] + t * ROWB + ((c ^ ACfg<D>::swz(t)) << 4)) = (t < T) ? x[im][u] : make_uint4(0u, 0u, 0u, 0u);
      }
    }
  }
}

// Reader of one LDS image.  The per-lane byte offsets of its two access patterns are computed once, INCLUDING the image's base
// offset, and made opaque to the optimizer: every LDS address in the hot loops is then one VGPR + a uniform row offset, instead
// of a chain of adds re-deriving it from the lane id and an image base that does not fit the 16-bit DS offset field.
//   rows(ks, row0): 8 consecutive d (d = 32 ks + 8 (lane >> 4) ...) of frame row0 + (lane & 15): the A operand [row][d] or the
//                   B operand [d][col = row] of a product reducing over d;
//   cols(db, t0):   the A operand [d = 16 db + (lane & 15)][kappa] of a product reducing over the 32 frames from t0, with the
//                   reduction index permuted as kappa = 8g + i <-> frame t0 + 4g + i (i < 4), t0 + 16 + 4g + (i - 4) (i >= 4):
//                   exactly the frames whose scores lane group g holds in the accumulators of two 16-frame score tiles
//                   (pack_acc below); two hardware transpose reads (ds_read_b64_tr_b16).
template <int D> struct ImgRd {
  int fr[ACfg<D>::KS], tc[ACfg<D>::DB];
  bool zero;                                           // head dim 16: the upper half of the 32-deep step is zero padding
  __device__ __forceinline__ void init(const char* img, int lane) {
    constexpr int ROWB = ACfg<D>::ROWB;
    // 32-bit LDS address of the image (its relocated base folded in here, once, instead of an add per access)
    typedef const __attribute__((address_space(3))) char* lds_cptr;
    const int img_off = (int)(unsigned)(size_t)((lds_cptr)img);
    const int l15 = lane & 15, g = lane >> 4, q = l15 >> 2, p = l15 & 3;
    zero = D < 32 && 8 * g >= D;
#pragma unroll
    for (int ks = 0; ks < ACfg<D>::KS; ++ks) {
      fr[ks] = img_off + l15 * ROWB + (((4 * ks + g) ^ ACfg<D>::swz(l15)) << 4);
      asm volatile("" : "+v"(fr[ks]));
    }
#pragma unroll
    for (int db = 0; db < ACfg<D>::DB; ++db) {
      const int row = 4 * g + q;                       // + t0 (a multiple of 8: same swizzle); the second read is 16 rows further
      tc[db] = img_off + row * ROWB + (((2 * db + (p >> 1)) ^ ACfg<D>::swz(row)) << 4) + (p & 1) * 8;
      asm volatile("" : "+v"(tc[db]));
    }
  }
  __device__ __forceinline__ bf16x8_t rows(int ks, int row0) const {
    typedef const bf16x8_t __attribute__((address_space(3))) * lds_ptr8;
    bf16x8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
    if (D < 32 && zero) return z;
    return *(lds_ptr8)(size_t)(unsigned)(fr[ks] + row0 * ACfg<D>::ROWB);
  }
  // the same accesses on a second image `delta` bytes (wave-uniform) behind this one: the lo plane of a planar pair
  __device__ __forceinline__ bf16x8_t rows(int ks, int row0, int delta) const {
    typedef const bf16x8_t __attribute__((address_space(3))) * lds_ptr8;
    bf16x8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
    if (D < 32 && zero) return z;
    return *(lds_ptr8)(size_t)(unsigned)(fr[ks] + delta + row0 * ACfg<D>::ROWB);
  }
  __device__ __forceinline__ bf16x8_t cols(int db, int t0, int delta) const {
    typedef bf16x4_t __attribute__((address_space(3))) * lds_ptr;
    const unsigned a0 = (unsigned)(tc[db] + delta + t0 * ACfg<D>::ROWB);
    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(size_t)(a0));
    const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(size_t)(a0 + 16 * ACfg<D>::ROWB));
    bf16x8_t f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3]; f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
  }
  __device__ __forceinline__ bf16x8_t cols(int db, int t0) const {
    typedef bf16x4_t __attribute__((address_space(3))) * lds_ptr;
    const unsigned a0 = (unsigned)(tc[db] + t0 * ACfg<D>::ROWB);
    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(size_t)(a0));
    const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(size_t)(a0 + 16 * ACfg<D>::ROWB));
    bf16x8_t f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3]; f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
  }
};

// two 16x16 accumulator tiles (rows 4g+r of frames [t0, t0+16) and [t0+16, t0+32), column on the lane) -> B fragment
__device__ __forceinline__ bf16x8_t pack_acc(const f32x4& a, const f32x4& b) {
  union { uint4 u; bf16x8_t v; } r;
  r.u.x = pack_bf16x2(a[0], a[1]); r.u.y = pack_bf16x2(a[2], a[3]);
  r.u.z = pack_bf16x2(b[0], b[1]); r.u.w = pack_bf16x2(b[2], b[3]);
  return r.v;
}

__device__ __forceinline__ float group_max(float v) {      // over the 4 lane groups (lane>>4) that share a column
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

__device__ __forceinline__ void store4(bf16* p, const f32x4& v, float s) {
  uint2 r;
  r.x = pack_bf16x2(v[0] * s, v[1] * s);
  r.y = pack_bf16x2(v[2] * s, v[3] * s);
  *reinterpret_cast<uint2*>(p) = r;
}

// gradient outputs (dQ, dK, dV) of the backward kernels: bf16, or - gout != 0 - fp16 of gout * value (a layer whose dgrad / weight-gradient GEMMs
// run on fp16 operands: kernels.h GemmB16Args::f16; gout is a power of two that lifts the gradients into fp16's normal range)
// F16G is a template parameter of the backward kernels, not a run-time branch: the saturating store (slow path, counters: common.h) is inlined
// at every one of their 12-24 store sites, and the default bf16 backward should not carry it (round 4: the same construct had pushed a GEMM
// kernel past the instruction cache)
template <bool F16G>
__device__ __forceinline__ void store4g(bf16* p, const f32x4& v, float s, float gout, unsigned* __restrict__ cnt) {
  if constexpr (F16G) {
    s *= gout;
    *reinterpret_cast<uint2*>(p) = sat_f16x4(v[0] * s, v[1] * s, v[2] * s, v[3] * s, cnt);      // saturating: common.h
  } else store4(p, v, s);
}
// set by the engine for the backward launches issued next on this thread: device address of the scale (null = bf16 outputs); see store4g
static thread_local const float* g_grad_f16 = nullptr;
void attn_grad_f16_override(const float* gout) { g_grad_f16 = gout; }
// set by the engine likewise: the temporal backward launches issued next read `out` as an fp16 plane (f16f8 = 3) instead of bf16
static thread_local int g_out_f16 = 0;
void attn_out_f16_override(int on) { g_out_f16 = on; }

// =============================================================================================
// forward: O = softmax(scale Q K^T) V ; lse = log sum exp of the scaled scores
// =============================================================================================
// 8 waves per workgroup, <= 128 VGPRs: two workgroups (72 KiB of LDS each) = 16 waves per CU, to cover the MFMA -> softmax -> MFMA
// dependency chain of a strip with other strips' work
// NTC: compile-time number of 16-frame tiles (16 for the 243-frame windows: every loop over tiles fully unrolled, LDS row offsets
// become instruction immediates, the padding mask exists in the last tile only) or 0 (run-time count, rolled loops).
template <int D, int NTC>
__global__ __launch_bounds__(512, 4) void attn_tmfma_fwd_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out,
                                                              float* __restrict__ lse, int T, int J, int C, int H, float scale) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  constexpr int ROWB = ACfg<D>::ROWB, KS = ACfg<D>::KS, DB = ACfg<D>::DB;
  // short windows (NTC == 0): the images hold round_up(T, 32) frames and the workgroup has one wave per 16-query strip (at most 8), so
  // that many workgroups share a CU instead of one 64 KiB, 8-wave workgroup with most waves idle
  const int rows = NTC ? TP : (T + 31) & ~31, nw = NTC ? 8 : (int)(blockDim.x >> 6);
  char* Ks = sm;
  char* Vs = sm + rows * ROWB;
  const int unit = blockIdx.x, h = unit % H, bj = unit / H, j = bj % J, b = bj / J;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l15 = lane & 15;
  const long rs3 = (long)J * 3 * C;
  const bf16* qb = qkv + ((long)b * T * J + j) * 3 * C + h * D;
  {
    char* const imgs[2] = {Ks, Vs};
    const bf16* const srcs[2] = {qb + C, qb + 2 * C};
    const long strides[2] = {rs3, rs3};
    stage_rows_multi<D, 2>(imgs, srcs, strides, T, rows, tid, nw * 64);
  }
  __syncthreads();
  const int ntile = NTC ? NTC : (T + 15) >> 4;
  ImgRd<D> Kr, Vr;
  Kr.init(Ks, lane);
  Vr.init(Vs, lane);
  const float scale2 = scale * 1.4426950408889634f;    // softmax evaluated as 2^(x log2 e): one v_exp_f32 per score, no extra multiply
  for (int qt = wave; qt < ntile; qt += nw) {
    const int tq = qt * 16 + l15;                      // this lane's query (column of every tile below)
    bf16x8_t bq[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int d0 = 32 * ks + 8 * g;
      bf16x8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
      bq[ks] = (tq < T && d0 < D) ? *reinterpret_cast<const bf16x8_t*>(qb + (long)tq * rs3 + d0) : z;
    }
    // scores^T strip in the log2 domain: s[kt][r] = (scale log2 e) K[kt*16 + 4g + r] . Q[tq]; only the last key tile has padding
    f32x4 s[NTILE];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NTILE; ++kt) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if (kt < ntile) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Kr.rows(ks, kt * 16), bq[ks], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = smul(acc[r], scale2);
        if (kt == ntile - 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (kt * 16 + 4 * g + r >= T) acc[r] = -INFINITY;
        }
        mx = fmaxf(fmaxf(mx, fmaxf(acc[0], acc[1])), fmaxf(acc[2], acc[3]));
      } else {
        acc = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      }
      s[kt] = acc;
    }
    mx = group_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NTILE; ++kt) {
      if (kt < ntile) {
        f32x4 e;
#pragma unroll
        for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(ssub(s[kt][r], mx));
        s[kt] = e;
        sum += (e[0] + e[1]) + (e[2] + e[3]);
      } else {
        s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    sum = group_sum(sum);
    // O^T[d][tq] = sum_t V[t][d] P[tq][t]
    f32x4 o[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) o[db] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kp = 0; kp < NTILE / 2; ++kp) {
      if (2 * kp < ntile) {
        const bf16x8_t bp = pack_acc(s[2 * kp], s[2 * kp + 1]);
#pragma unroll
        for (int db = 0; db < DB; ++db) o[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Vr.cols(db, kp * 32), bp, o[db], 0, 0, 0);
      }
    }
    if (tq < T) {
      const float inv = 1.0f / sum;
      bf16* orow = out + ((long)(b * T + tq) * J + j) * C + h * D;
#pragma unroll
      for (int db = 0; db < DB; ++db) store4(orow + 16 * db + 4 * g, o[db], inv);
      if (g == 0) lse[(long)unit * T + tq] = (mx + __log2f(sum)) * 0.6931471805599453f;      // natural-log units
    }
  }
}

// =============================================================================================
// backward: dQ, dK, dV from Q, K, V, O, dO and the saved log-sum-exp
// =============================================================================================
// One workgroup of 16 waves (<= 128 VGPRs) per (window, joint, head): Q, K, V, dO staged once (4 x 36 KiB + statistics = 146 KiB,
// one workgroup per CU); wave w owns the 16-query strip w in pass A (dQ) and the 16-key strip w in pass B (dK, dV) and takes
// its own strip's fragments from the LDS images like everyone else's.  16 waves (instead of 8 with two strips each) cover the
// MFMA -> exp -> MFMA dependency chain of a strip with other strips' work.
template <int D, int NTC, bool F16G>
__global__ __launch_bounds__(1024) void attn_tmfma_bwd_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ out,
                                                               const bf16* __restrict__ dout, const float* __restrict__ lse,
                                                               bf16* __restrict__ dqkv, int T, int J, int C, int H, float scale,
                                                               int debug, const float* __restrict__ gout_p, int out_is_f16) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  constexpr int ROWB = ACfg<D>::ROWB, KS = ACfg<D>::KS, DB = ACfg<D>::DB, NW = 16, UNR = NTC ? NTC / 2 : 1;
  const float gout = F16G ? *gout_p : 0.f;      // this backward's gradient scale (engine: grad_scale_kernel); F16G false: bf16 outputs
  unsigned* const gcnt = F16G ? reinterpret_cast<unsigned*>(const_cast<float*>(gout_p)) + 4 : nullptr;      // its saturation counters
  const int rows = NTC ? TP : (T + 31) & ~31, nw = NTC ? NW : (int)(blockDim.x >> 6);     // short windows: see the forward kernel
  char* Qs = sm;
  char* Ks = Qs + rows * ROWB;
  char* Vs = Ks + rows * ROWB;
  char* Gs = Vs + rows * ROWB;                           // dO
  float* Ls = reinterpret_cast<float*>(Gs + rows * ROWB);  // MINUS the log2-domain log-sum-exp per query (-inf for padding -> p = 0)
  float* Dl = Ls + rows;                                 // MINUS delta = -sum_d dO * O per query
  const int unit = blockIdx.x, h = unit % H, bj = unit / H, j = bj % J, b = bj / J;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l15 = lane & 15;
  const long rs3 = (long)J * 3 * C, rs1 = (long)J * C;
  const bf16* qb = qkv + ((long)b * T * J + j) * 3 * C + h * D;
  const bf16* ob = out + ((long)b * T * J + j) * C + h * D;
  const bf16* gb = dout + ((long)b * T * J + j) * C + h * D;
  // delta and log-sum-exp: 4 threads per frame, each a quarter of the head dim.  The first pass's rows (all of them when rows <= 16 waves x 16)
  // are requested BEFORE the images are staged - unconditional loads from clamped frames - so that they travel with the image loads
  // instead of costing two more memory round trips behind them.
  constexpr int DQ = D / 16;                               // 8-byte loads of dO and of O per thread
  auto delta_load = [&](int t0, uint2 (&a)[DQ], uint2 (&o)[DQ], float& l) {
    const int tcl = min(t0 + (tid >> 2), T - 1), part = tid & 3;
#pragma unroll
    for (int i = 0; i < DQ; ++i) {
      a[i] = *reinterpret_cast<const uint2*>(gb + (long)tcl * rs1 + part * (D / 4) + 4 * i);
      o[i] = *reinterpret_cast<const uint2*>(ob + (long)tcl * rs1 + part * (D / 4) + 4 * i);
    }
    l = lse[(long)unit * T + tcl];
  };
  auto delta_commit = [&](int t0, const uint2 (&a)[DQ], const uint2 (&o)[DQ], float l) {
    const int t = t0 + (tid >> 2), part = tid & 3;
    float dl = 0.f;
#pragma unroll
    for (int i = 0; i < DQ; ++i) {
      // (out_is_f16: O exists as the fp16 plane of an f16f8 pair only - mp_model_config::f16f8 = 3 - and is rounded to bf16 here, so that delta is
      // the sum a bf16 copy of O would have given)
      const float4 a4 = bf16x4_to_float4(a[i]), o4 = bf16x4_to_float4(out_is_f16 ? f16x4_to_bf16x4(o[i]) : o[i]);
      dl += (a4.x * o4.x + a4.y * o4.y) + (a4.z * o4.z + a4.w * o4.w);
    }
    if (t >= T) dl = 0.f;
    dl += __shfl_xor(dl, 1, 64);
    dl += __shfl_xor(dl, 2, 64);
    if (part == 0 && t < rows) {
      Dl[t] = -dl;                                        // stored negated: ds = p (dp + (-delta)) is a packed add + a packed multiply
      Ls[t] = (t < T) ? l * -1.4426950408889634f : -INFINITY;    // MINUS lse in log2 units
    }
  };
  {
    uint2 da[DQ], dob[DQ];
    float dlse;
    delta_load(0, da, dob, dlse);
    char* const imgs[4] = {Qs, Ks, Vs, Gs};
    const bf16* const srcs[4] = {qb, qb + C, qb + 2 * C, gb};
    const long strides[4] = {rs3, rs3, rs3, rs1};
    stage_rows_multi<D, 4>(imgs, srcs, strides, T, rows, tid, nw * 64);
    delta_commit(0, da, dob, dlse);
    for (int t0 = nw * 16; t0 < rows; t0 += nw * 16) {
      delta_load(t0, da, dob, dlse);
      delta_commit(t0, da, dob, dlse);
    }
  }
  __syncthreads();
  const int ntile = NTC ? NTC : (T + 15) >> 4;
  bf16* dq_base = dqkv + ((long)b * T * J + j) * 3 * C + h * D;
#ifdef MP_GEMM_DIAG
  if (debug & 1) return;                                 // timing ablation: staging only
#endif
  const float scale2 = scale * 1.4426950408889634f;    // p = 2^(s scale log2 e - lse log2 e)
  ImgRd<D> Qr, Kr, Vr, Gr;
  Qr.init(Qs, lane);
  Kr.init(Ks, lane);
  Vr.init(Vs, lane);
  Gr.init(Gs, lane);

  // ---- pass A: dQ (scores in the [key][query] orientation: query on the lane, 4 keys per accumulator) ----
  for (int qt = wave; qt < ntile; qt += nw) {
    const int tq = qt * 16 + l15;
    bf16x8_t bq[KS], bg[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bq[ks] = Qr.rows(ks, qt * 16);
      bg[ks] = Gr.rows(ks, qt * 16);
    }
    const float nL = Ls[tq], ndl = Dl[tq];
    f32x4 dq[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) dq[db] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll UNR
    for (int kp = 0; 2 * kp < ntile; ++kp) {
      f32x4 ds[2];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int kt = 2 * kp + hf;
        f32x4 sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Kr.rows(ks, kt * 16), bq[ks], sc, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Vr.rows(ks, kt * 16), bg[ks], dp, 0, 0, 0);
        }
        // component by component (fmaf / add / mul on single floats): the vector forms compile to v_pk_*_f32, which cost ~20 cycles
        // each beside MFMAs on this chip (MI355X_MICROARCH: packed fp32 next to the matrix cores)
        f32x4 e;
#pragma unroll
        for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(sfma(sc[r], scale2, nL));
        if (kt >= ntile - 1) {                         // key padding exists in the last tile only (and in the tile past an odd count)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (kt * 16 + 4 * g + r >= T) e[r] = 0.f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) ds[hf][r] = smul(e[r], sadd(dp[r], ndl));
      }
      const bf16x8_t bd = pack_acc(ds[0], ds[1]);
#pragma unroll
      for (int db = 0; db < DB; ++db) dq[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Kr.cols(db, kp * 32), bd, dq[db], 0, 0, 0);
    }
    if (tq < T) {
#pragma unroll
      for (int db = 0; db < DB; ++db) store4g<F16G>(dq_base + (long)tq * rs3 + 16 * db + 4 * g, dq[db], scale, gout, gcnt);
    }
  }
#ifdef MP_GEMM_DIAG
  if (debug & 2) return;                                 // timing ablation: no dK/dV pass
#endif

  // ---- pass B: dK, dV (scores in the [query][key] orientation) ----
  for (int kt = wave; kt < ntile; kt += nw) {
    const int tk = kt * 16 + l15;
    bf16x8_t bk[KS], bv[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bk[ks] = Kr.rows(ks, kt * 16);
      bv[ks] = Vr.rows(ks, kt * 16);
    }
    f32x4 dk[DB], dv[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) { dk[db] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[db] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll UNR
    for (int qp = 0; 2 * qp < ntile; ++qp) {
      f32x4 p2[2], d2[2];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int qt = 2 * qp + hf;
        f32x4 sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Qr.rows(ks, qt * 16), bk[ks], sc, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Gr.rows(ks, qt * 16), bv[ks], dp, 0, 0, 0);
        }
        // rows of this orientation are queries (4 consecutive per lane group); -lse = -inf on padding -> p = 0
        const f32x4 nL = *reinterpret_cast<const f32x4*>(Ls + qt * 16 + 4 * g), ndl = *reinterpret_cast<const f32x4*>(Dl + qt * 16 + 4 * g);
        f32x4 e;
#pragma unroll
        for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(sfma(sc[r], scale2, nL[r]));
        p2[hf] = e;
#pragma unroll
        for (int r = 0; r < 4; ++r) d2[hf][r] = smul(e[r], sadd(dp[r], ndl[r]));
      }
      const bf16x8_t bp = pack_acc(p2[0], p2[1]), bd = pack_acc(d2[0], d2[1]);
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        dv[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Gr.cols(db, qp * 32), bp, dv[db], 0, 0, 0);
        dk[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Qr.cols(db, qp * 32), bd, dk[db], 0, 0, 0);
      }
    }
    if (tk < T) {
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        store4g<F16G>(dq_base + C + (long)tk * rs3 + 16 * db + 4 * g, dk[db], scale, gout, gcnt);
        store4g<F16G>(dq_base + 2 * C + (long)tk * rs3 + 16 * db + 4 * g, dv[db], 1.0f, gout, gcnt);
      }
    }
  }
}

// =============================================================================================
// spatial attention on the matrix cores: the N = 17 joints (16 bones) of ONE frame attend to each other.
// One workgroup per frame, one wave per head: the frame's whole fused qkv block (N x 3C bf16, contiguous in HBM: 52 KB
// at C = 512) is staged into LDS in one coalesced sweep, each wave then works on its head's 64-wide (16-wide) slices.
// N is padded to 32 inside the MFMA tiles by CLAMPING the token index of the padding lanes (finite duplicate data, their
// probabilities are masked to exactly 0), so no zero rows are stored.
// =============================================================================================
template <int D>
__device__ __forceinline__ bf16x8_t frag_tok(const char* __restrict__ S, int pitch, int off, int tok0, int ks, int lane, int N) {
  const int g = lane >> 4, d0 = 32 * ks + 8 * g;
  bf16x8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
  if (D < 32 && d0 >= D) return z;
  const int tok = min(tok0 + (lane & 15), N - 1);
  return *reinterpret_cast<const bf16x8_t*>(S + tok * pitch + off + d0 * 2);
}
// A operand [d = 16 db + (lane & 15)][kappa] over the 32 (padded) tokens, kappa permuted like pack_acc's rows
template <int D>
__device__ __forceinline__ bf16x8_t frag_tokT(const char* __restrict__ S, int pitch, int off, int db, int lane, int N) {
  typedef bf16x4_t __attribute__((address_space(3))) * lds_ptr;
  const int li = lane & 15, q = li >> 2, p = li & 3, g = lane >> 4;
  const int r1 = min(4 * g + q, N - 1), r2 = min(16 + 4 * g + q, N - 1);
  const int cb = off + (16 * db + 4 * p) * 2;
  const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(S + r1 * pitch + cb));
  const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(S + r2 * pitch + cb));
  bf16x8_t f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3]; f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}

// Staging of a contiguous block of rows x row_bytes into an LDS image of `pitch` bytes per row, in two halves with NB chunks per thread in flight: issue requests the chunks tid, tid + nthreads, ... (clamped to the block's last
// chunk: always a valid address, no branch around the load), commit writes them.  A chunk-at-a-time loop (load, write, next) compiles to load / s_waitcnt vmcnt(0) /
// ds_write per chunk - one memory round trip per 16 bytes and thread, 8-9 of them per frame in the spatial backward.  Chunks past
// NB * nthreads (frames of more than ~21 tokens at head dim 64) are left to stage_block_tail.
template <int NB>
__device__ __forceinline__ void stage_block_issue(uint4 (&x)[NB], const bf16* __restrict__ src, int rows, int row_bytes, int tid, int nthreads) {
  const int total = rows * (row_bytes >> 4);
#pragma unroll
  for (int u = 0; u < NB; ++u) x[u] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(src) + (long)min(tid + u * nthreads, total - 1) * 16);
}
// the requested chunks pass through an opaque move: without it hipcc sinks every load into the conditional block of its LDS write (the
// commit below), which restores the one-round-trip-per-chunk sequence
template <int NB>
__device__ __forceinline__ void stage_block_pin(uint4 (&x)[NB]) {
#pragma unroll
  for (int u = 0; u < NB; ++u) asm volatile("" : "+v"(x[u].x), "+v"(x[u].y), "+v"(x[u].z), "+v"(x[u].w));
}
template <int NB>
__device__ __forceinline__ void stage_block_commit(char* __restrict__ S, int pitch, const uint4 (&x)[NB], int rows, int row_bytes, int tid, int nthreads) {
  const int cpr = row_bytes >> 4, total = rows * cpr;
#pragma unroll
  for (int u = 0; u < NB; ++u) {
    const int c = tid + u * nthreads;
    const int r = c / cpr, k = c - r * cpr;
    if (c < total) *reinterpret_cast<uint4*>(S + r * pitch + k * 16) = x[u];
  }
}
template <int NB>
__device__ __forceinline__ void stage_block_tail(char* __restrict__ S, int pitch, const bf16* __restrict__ src, int rows, int row_bytes, int tid, int nthreads) {
  const int cpr = row_bytes >> 4;
  for (int c = tid + NB * nthreads; c < rows * cpr; c += nthreads) {
    const int r = c / cpr, k = c - r * cpr;
    *reinterpret_cast<uint4*>(S + r * pitch + k * 16) = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(src) + (long)r * row_bytes + k * 16);
  }
}

// softmax over the (<= 32) keys of this lane's query column; s0/s1: accumulator tiles of keys 0..15 / 16..31 (raw q.k)
__device__ __forceinline__ void col_softmax(f32x4& s0, f32x4& s1, int g, int N, float scale) {
  float mx = -INFINITY;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    s0[r] = (4 * g + r < N) ? s0[r] * scale : -INFINITY;
    s1[r] = (16 + 4 * g + r < N) ? s1[r] * scale : -INFINITY;
    mx = fmaxf(mx, fmaxf(s0[r], s1[r]));
  }
  mx = group_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    s0[r] = __expf(s0[r] - mx);
    s1[r] = __expf(s1[r] - mx);
    sum += s0[r] + s1[r];
  }
  const float inv = 1.0f / group_sum(sum);
#pragma unroll
  for (int r = 0; r < 4; ++r) { s0[r] *= inv; s1[r] *= inv; }
}

template <int D>
__global__ __launch_bounds__(512) void attn_smfma_fwd_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out, int N, int C, int H,
                                                               float scale) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  constexpr int KS = ACfg<D>::KS, DB = ACfg<D>::DB;
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, h = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  const int pitch = 6 * C + 16;
  {
    constexpr int NBQ = D == 64 ? 8 : 2;                    // 16-byte chunks of the frame's qkv block per thread: 6 N D / 1024 (17 tokens: 6.4 / 1.6)
    uint4 xq[NBQ];
    stage_block_issue<NBQ>(xq, qkv + (long)f * N * 3 * C, N, 6 * C, tid, blockDim.x);
    stage_block_pin<NBQ>(xq);
    stage_block_commit<NBQ>(sm, pitch, xq, N, 6 * C, tid, blockDim.x);
    stage_block_tail<NBQ>(sm, pitch, qkv + (long)f * N * 3 * C, N, 6 * C, tid, blockDim.x);
  }
  __syncthreads();
  const int oq = h * D * 2, ok = 2 * C + oq, ov = 4 * C + oq;
  bf16x8_t qf[2][KS], kf[2][KS];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qf[t][ks] = frag_tok<D>(sm, pitch, oq, 16 * t, ks, lane, N);
      kf[t][ks] = frag_tok<D>(sm, pitch, ok, 16 * t, ks, lane, N);
    }
  bf16x8_t vT[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db) vT[db] = frag_tokT<D>(sm, pitch, ov, db, lane, N);
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    if (qt * 16 >= N) break;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};      // S^T[key][query]: keys 0..15 / 16..31
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[0][ks], qf[qt][ks], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[1][ks], qf[qt][ks], s1, 0, 0, 0);
    }
    col_softmax(s0, s1, g, N, scale);
    const bf16x8_t bp = pack_acc(s0, s1);
    const int tq = qt * 16 + l15;
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vT[db], bp, o, 0, 0, 0);
      if (tq < N) store4(out + ((long)f * N + tq) * C + h * D + 16 * db + 4 * g, o, 1.0f);
    }
  }
}

template <int D, bool F16G>
__global__ __launch_bounds__(512) void attn_smfma_bwd_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ dout,
                                                               bf16* __restrict__ dqkv, int N, int C, int H, float scale, const float* __restrict__ gout_p) {
  const float gout = F16G ? *gout_p : 0.f;
  unsigned* const gcnt = F16G ? reinterpret_cast<unsigned*>(const_cast<float*>(gout_p)) + 4 : nullptr;
  extern __shared__ __attribute__((aligned(16))) char sm[];
  constexpr int KS = ACfg<D>::KS, DB = ACfg<D>::DB;
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, h = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  const int pitch = 6 * C + 16, gpitch = 2 * C + 16;
  char* gs = sm + N * pitch;                               // dO block of the frame
  {
    // the frame's qkv block (8 chunks per thread) and its dO block (3) are requested together, then written
    constexpr int NBQ = D == 64 ? 8 : 2, NBG = D == 64 ? 3 : 1;
    uint4 xq[NBQ], xg[NBG];
    stage_block_issue<NBQ>(xq, qkv + (long)f * N * 3 * C, N, 6 * C, tid, blockDim.x);
    stage_block_issue<NBG>(xg, dout + (long)f * N * C, N, 2 * C, tid, blockDim.x);
    stage_block_pin<NBQ>(xq);
    stage_block_pin<NBG>(xg);
    stage_block_commit<NBQ>(sm, pitch, xq, N, 6 * C, tid, blockDim.x);
    stage_block_commit<NBG>(gs, gpitch, xg, N, 2 * C, tid, blockDim.x);
    stage_block_tail<NBQ>(sm, pitch, qkv + (long)f * N * 3 * C, N, 6 * C, tid, blockDim.x);
    stage_block_tail<NBG>(gs, gpitch, dout + (long)f * N * C, N, 2 * C, tid, blockDim.x);
  }
  __syncthreads();
  const int oq = h * D * 2, ok = 2 * C + oq, ov = 4 * C + oq, og = oq;
  bf16x8_t qf[2][KS], kf[2][KS], vf[2][KS], gf[2][KS];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qf[t][ks] = frag_tok<D>(sm, pitch, oq, 16 * t, ks, lane, N);
      kf[t][ks] = frag_tok<D>(sm, pitch, ok, 16 * t, ks, lane, N);
      vf[t][ks] = frag_tok<D>(sm, pitch, ov, 16 * t, ks, lane, N);
      gf[t][ks] = frag_tok<D>(gs, gpitch, og, 16 * t, ks, lane, N);
    }
  bf16* dbase = dqkv + (long)f * N * 3 * C + h * D;
  // ---- pass A ([key][query] orientation): softmax statistics per query column, dQ ----
  float mq[2], lq[2], dlq[2];                              // per query column (tile qt, lane & 15): max, 1/sum handled inside
  f32x4 pA[2][2];                                          // normalised P^T tiles [qt][kt]
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f}, d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[0][ks], qf[qt][ks], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[1][ks], qf[qt][ks], s1, 0, 0, 0);
      d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[0][ks], gf[qt][ks], d0, 0, 0, 0);   // dP^T[key][query]
      d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[1][ks], gf[qt][ks], d1, 0, 0, 0);
    }
    // raw maxima / sums are needed again in pass B: recompute them there from the same scores via these two numbers
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s0[r] = (4 * g + r < N) ? s0[r] * scale : -INFINITY;
      s1[r] = (16 + 4 * g + r < N) ? s1[r] * scale : -INFINITY;
      mx = fmaxf(mx, fmaxf(s0[r], s1[r]));
    }
    mx = group_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s0[r] = __expf(s0[r] - mx);
      s1[r] = __expf(s1[r] - mx);
      sum += s0[r] + s1[r];
    }
    sum = group_sum(sum);
    const float inv = 1.0f / sum;
    float dl = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s0[r] *= inv; s1[r] *= inv;
      dl += s0[r] * d0[r] + s1[r] * d1[r];
    }
    dl = group_sum(dl);
    mq[qt] = mx; lq[qt] = inv; dlq[qt] = dl;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      d0[r] = s0[r] * (d0[r] - dl);                        // dS^T
      d1[r] = s1[r] * (d1[r] - dl);
    }
    pA[qt][0] = s0; pA[qt][1] = s1;
    const bf16x8_t bds = pack_acc(d0, d1);
    const int tq = qt * 16 + l15;
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      f32x4 dq = {0.f, 0.f, 0.f, 0.f};
      dq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tokT<D>(sm, pitch, ok, db, lane, N), bds, dq, 0, 0, 0);
      if (tq < N) store4g<F16G>(dbase + (long)tq * 3 * C + 16 * db + 4 * g, dq, scale, gout, gcnt);
    }
  }
  // ---- pass B ([query][key] orientation): dK, dV.  Row statistics come from pass A's column statistics by shuffle ----
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    if (kt * 16 >= N) break;
    f32x4 p2[2], ds2[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      f32x4 sa = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[qt][ks], kf[kt][ks], sa, 0, 0, 0);   // S[query][key]
        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf[qt][ks], vf[kt][ks], dp, 0, 0, 0);   // dP[query][key]
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int src = 4 * g + r;                          // lane holding query (qt, 4g + r)'s column statistics
        const float m = __shfl(mq[qt], src, 64), il = __shfl(lq[qt], src, 64), dl = __shfl(dlq[qt], src, 64);
        const bool ok2 = (qt * 16 + 4 * g + r < N) && (kt * 16 + l15 < N);
        const float p = ok2 ? __expf(sa[r] * scale - m) * il : 0.f;
        p2[qt][r] = p;
        ds2[qt][r] = p * (dp[r] - dl);
      }
    }
    const bf16x8_t bp = pack_acc(p2[0], p2[1]), bds = pack_acc(ds2[0], ds2[1]);
    const int tk = kt * 16 + l15;
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      f32x4 dv = {0.f, 0.f, 0.f, 0.f}, dk = {0.f, 0.f, 0.f, 0.f};
      dv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tokT<D>(gs, gpitch, og, db, lane, N), bp, dv, 0, 0, 0);
      dk = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tokT<D>(sm, pitch, oq, db, lane, N), bds, dk, 0, 0, 0);
      if (tk < N) {
        store4g<F16G>(dbase + C + (long)tk * 3 * C + 16 * db + 4 * g, dk, scale, gout, gcnt);
        store4g<F16G>(dbase + 2 * C + (long)tk * 3 * C + 16 * db + 4 * g, dv, 1.0f, gout, gcnt);
      }
    }
  }
  (void)pA;
}

float attn_qk_scale(int D);      // attention.hip: head_dim ** -0.5 or the engine's override (muP)
bool attn_smfma_supported(int N, int D, int H) { return N >= 16 && N <= 32 && (D == 64 || D == 16) && H >= 1 && H <= 8; }

int attn_smfma_fwd(const bf16* qkv, bf16* out, int B, int T, int J, int C, int H, hipStream_t st) {
  const int D = C / H;
  MP_CHECK(attn_smfma_supported(J, D, H) && C % 8 == 0, MP_ERR_ARG, "attn_smfma_fwd: J=%d D=%d H=%d unsupported", J, D, H);
  const float scale = attn_qk_scale(D);
  const size_t lds = (size_t)J * (6 * C + 16);
  if (D == 64) {
    static bool attr_set = false;
    if (!attr_set) {
      MP_HIP(hipFuncSetAttribute((const void*)attn_smfma_fwd_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_set = true;
    }
    hipLaunchKernelGGL(attn_smfma_fwd_kernel<64>, dim3(B * T), dim3(H * 64), lds, st, qkv, out, J, C, H, scale);
  } else {
    hipLaunchKernelGGL(attn_smfma_fwd_kernel<16>, dim3(B * T), dim3(H * 64), lds, st, qkv, out, J, C, H, scale);
  }
  MP_LAUNCH_CHECK();
  return MP_OK;
}

int attn_smfma_bwd(const bf16* qkv, const bf16* dout, bf16* dqkv, int B, int T, int J, int C, int H, hipStream_t st) {
  const int D = C / H;
  MP_CHECK(attn_smfma_supported(J, D, H) && C % 8 == 0, MP_ERR_ARG, "attn_smfma_bwd: J=%d D=%d H=%d unsupported", J, D, H);
  const float scale = attn_qk_scale(D);
  const size_t lds = (size_t)J * (6 * C + 16) + (size_t)J * (2 * C + 16);
  if (D == 64) {
    static bool attr_set = false;
    if (!attr_set) {
      MP_HIP(hipFuncSetAttribute((const void*)attn_smfma_bwd_kernel<64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      MP_HIP(hipFuncSetAttribute((const void*)attn_smfma_bwd_kernel<64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_set = true;
    }
    if (g_grad_f16 != nullptr) hipLaunchKernelGGL((attn_smfma_bwd_kernel<64, true>), dim3(B * T), dim3(H * 64), lds, st, qkv, dout, dqkv, J, C, H, scale, g_grad_f16);
    else hipLaunchKernelGGL((attn_smfma_bwd_kernel<64, false>), dim3(B * T), dim3(H * 64), lds, st, qkv, dout, dqkv, J, C, H, scale, g_grad_f16);
  } else {
    if (g_grad_f16 != nullptr) hipLaunchKernelGGL((attn_smfma_bwd_kernel<16, true>), dim3(B * T), dim3(H * 64), lds, st, qkv, dout, dqkv, J, C, H, scale, g_grad_f16);
    else hipLaunchKernelGGL((attn_smfma_bwd_kernel<16, false>), dim3(B * T), dim3(H * 64), lds, st, qkv, dout, dqkv, J, C, H, scale, g_grad_f16);
  }
  MP_LAUNCH_CHECK();
  return MP_OK;
}


// =============================================================================================
// Split precision ("bf16x3", common.h): q, k, v and the output are planar hi/lo bf16 pairs; every matrix product is evaluated as
// lo.hi + hi.lo + hi.hi on the bf16 matrix cores (fp32 accumulate), the probabilities are split into hi/lo in registers.
// Same structure as the bf16 kernels above with twice the LDS images.
// =============================================================================================
__device__ __forceinline__ void pack_acc_x2(const f32x4& a, const f32x4& b, bf16x8_t& hi, bf16x8_t& lo) {
  union { uint4 u; bf16x8_t v; } h, l;
  split_bf16x2(a[0], a[1], h.u.x, l.u.x); split_bf16x2(a[2], a[3], h.u.y, l.u.y);
  split_bf16x2(b[0], b[1], h.u.z, l.u.z); split_bf16x2(b[2], b[3], h.u.w, l.u.w);
  hi = h.v; lo = l.v;
}
__device__ __forceinline__ void store4_x2(bf16* ph, bf16* pl, const f32x4& v, float s) {
  uint2 h, l;
  split_bf16x2(v[0] * s, v[1] * s, h.x, l.x);
  split_bf16x2(v[2] * s, v[3] * s, h.y, l.y);
  *reinterpret_cast<uint2*>(ph) = h;
  *reinterpret_cast<uint2*>(pl) = l;
}
// F8O (mp_model_config::f16f8 = 3): the attention output leaves as the "f16f8" planes the proj GEMM reads (common.h: ph = the fp16 plane, pl = the
// 8-bit correction plane - the same 2 bytes per element and the same 8 bytes per four channels as a bf16 hi / lo pair)
template <bool F8O>
__device__ __forceinline__ void store4_o(bf16* ph, bf16* pl, const f32x4& v, float s) {
  if constexpr (F8O) {
    uint2 h, c;
    pack4_f16f8(make_float4(v[0] * s, v[1] * s, v[2] * s, v[3] * s), h, c);
    *reinterpret_cast<uint2*>(ph) = h;
    *reinterpret_cast<uint2*>(pl) = c;
  } else store4_x2(ph, pl, v, s);
}
__device__ __forceinline__ int img_off(int img, int img_bytes) { return img * img_bytes; }
#define MP_MFMA3(acc, a_hi, a_lo, b_hi, b_lo)                                   \
  do {                                                                          \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo, b_hi, acc, 0, 0, 0);    \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, b_lo, acc, 0, 0, 0);    \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, b_hi, acc, 0, 0, 0);    \
  } while (0)

// temporal.  PERSISTENT: a workgroup (<= 8 waves, two 16-query strips per wave) walks the (window, joint, head) units u, u + G, ...;
// the four LDS images (K_hi, K_lo, V_hi, V_lo: 128 KiB at T = 243, one workgroup per CU) of unit u + G and its Q fragments are
// PREFETCHED INTO REGISTERS while unit u is computed (the loads are issued right after the images of u were written to LDS), so the
// HBM / L2 latency of the staging - as long as the compute of a unit when it is not overlapped - is hidden behind the MFMAs of the
// previous unit.  (The bf16 kernel above overlaps staging and compute by running two 64 KiB workgroups per CU instead.)
template <int D, bool FULL, bool F8O = false>      // FULL: all NTILE key tiles are in the window (T > 240): compile-time tile count, no per-tile branches
__global__ __launch_bounds__(512) void attn_tmfma_fwd_x3_kernel(const bf16* __restrict__ qkv_hi, const bf16* __restrict__ qkv_lo,
                                                                 bf16* __restrict__ out_hi, bf16* __restrict__ out_lo,
                                                                 float* __restrict__ lse, int nunits, int T, int J, int C, int H, float scale, int debug) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  // timing ablations, diagnostics build only (results wrong by design): 1 = images staged for the first unit only, 2 = one K / V fragment
  // read per strip instead of one per tile (no LDS traffic), 4 = no softmax arithmetic, 8 = no matrix-core work, 16 = no output stores,
  // 32 = no Q loads
#ifdef MP_GEMM_DIAG
#define MP_ADBG(bit) (debug & (bit))
#else
#define MP_ADBG(bit) 0
#endif
  constexpr int ROWB = ACfg<D>::ROWB, KS = ACfg<D>::KS, DB = ACfg<D>::DB, CH = ACfg<D>::CH, PER = 4;
  const int rows = FULL ? TP : (T + 31) & ~31, nw = (int)(blockDim.x >> 6), nthreads = (int)blockDim.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l15 = lane & 15;
  const long rs3 = (long)J * 3 * C;
  const int ntile = FULL ? NTILE : (T + 15) >> 4;
  const int img_bytes = rows * ROWB, nchunk = rows * CH;      // per image; nchunk <= PER * nthreads (launcher)
  ImgRd<D> Khr, Vhr;
  Khr.init(sm, lane);
  Vhr.init(sm + 2 * img_bytes, lane);
  const int dlo = __builtin_amdgcn_readfirstlane(img_bytes);  // hi image -> lo image of K and of V
  const float scale2 = scale * 1.4426950408889634f;

  uint4 pre[4][PER];                                          // next unit's images: K_hi, K_lo, V_hi, V_lo
  bf16x8_t qn_h[2][KS], qn_l[2][KS];                          // Q fragments of this wave's two strips: the current unit's until its last
                                                              // score strip is done, then (re-requested) the next unit's
  auto unit_base = [&](int unit) -> long {
    const int h = unit % H, bj = unit / H, j = bj % J, b = bj / J;
    return ((long)b * T * J + j) * 3 * C + h * D;
  };
  auto fetch_q = [&](int unit) {
    const long qoff = unit_base(unit);
    const bf16* qh = qkv_hi + qoff;
    const bf16* ql = qkv_lo + qoff;
#pragma unroll
    for (int sidx = 0; sidx < 2; ++sidx) {
      const int tq = (wave + sidx * nw) * 16 + l15;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int d0 = 32 * ks + 8 * g;
        bf16x8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
        const bool ok = tq < T && d0 < D && !MP_ADBG(32);
        qn_h[sidx][ks] = ok ? *reinterpret_cast<const bf16x8_t*>(qh + (long)tq * rs3 + d0) : z;
        qn_l[sidx][ks] = ok ? *reinterpret_cast<const bf16x8_t*>(ql + (long)tq * rs3 + d0) : z;
      }
    }
  };
  auto fetch = [&](int unit) {
    const long qoff = unit_base(unit);
    const bf16* qh = qkv_hi + qoff;
    const bf16* ql = qkv_lo + qoff;
#pragma unroll
    for (int img = 0; img < 4; ++img) {
      const bf16* src = ((img & 1) ? ql : qh) + ((img >> 1) ? 2 * C : C);
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int idx = tid + i * nthreads, t = idx / CH, c = idx - t * CH;
        uint4 x = make_uint4(0u, 0u, 0u, 0u);
        if (idx < nchunk && t < T) x = *reinterpret_cast<const uint4*>(src + (long)t * rs3 + c * 8);
        pre[img][i] = x;
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int img = 0; img < 4; ++img)
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int idx = tid + i * nthreads, t = idx / CH, c = idx - t * CH;
        if (idx < nchunk) *reinterpret_cast<uint4*>(sm + img_off(img, img_bytes) + t * ROWB + ((c ^ ACfg<D>::swz(t)) << 4)) = pre[img][i];
      }
  };

  int unit = blockIdx.x;
  if (unit >= nunits) return;
  fetch_q(unit);
  fetch(unit);
  bool first_unit = true;
  while (true) {
    __syncthreads();                       // every wave is done with the previous unit's images
    if (!MP_ADBG(1) || first_unit) commit();
    __syncthreads();
    const int next = unit + gridDim.x;
    if (next < nunits && !MP_ADBG(1)) fetch(next);        // in flight during the compute below
    first_unit = false;
    bool q_fetched = false;                // the next unit's Q fragments are requested once this wave's last score strip is done
    const int h = unit % H, bj = unit / H, j = bj % J, b = bj / J;
#pragma unroll
    for (int sidx = 0; sidx < 2; ++sidx) {
      const int qt = wave + sidx * nw;
      if (qt >= ntile) break;
      const int tq = qt * 16 + l15;
      f32x4 s[NTILE];
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < NTILE; ++kt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (kt < ntile) {
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const int krow = MP_ADBG(2) ? 0 : kt * 16;
            const bf16x8_t kh = Khr.rows(ks, krow), kl = Khr.rows(ks, krow, dlo);
            if (MP_ADBG(8)) { acc[0] += __builtin_bit_cast(float, (int)kh[0] + (int)kl[1]); continue; }
            MP_MFMA3(acc, kh, kl, qn_h[sidx][ks], qn_l[sidx][ks]);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[r] = smul(acc[r], scale2);
          if (kt == ntile - 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (kt * 16 + 4 * g + r >= T) acc[r] = -INFINITY;
          }
          mx = fmaxf(fmaxf(mx, fmaxf(acc[0], acc[1])), fmaxf(acc[2], acc[3]));
        } else {
          acc = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        }
        s[kt] = acc;
      }
      if (sidx == 1 && next < nunits) { fetch_q(next); q_fetched = true; }
      mx = group_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int kt = 0; kt < NTILE; ++kt) {
        if (MP_ADBG(4)) { sum += s[kt][0]; continue; }
        if (kt < ntile) {
          f32x4 e;
#pragma unroll
          for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(ssub(s[kt][r], mx));
          s[kt] = e;
          sum += (e[0] + e[1]) + (e[2] + e[3]);
        } else {
          s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      sum = group_sum(sum);
      f32x4 o[DB];
#pragma unroll
      for (int db = 0; db < DB; ++db) o[db] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kp = 0; kp < NTILE / 2; ++kp) {
        if (2 * kp < ntile) {
          bf16x8_t bph, bpl;
          pack_acc_x2(s[2 * kp], s[2 * kp + 1], bph, bpl);
#pragma unroll
          for (int db = 0; db < DB; ++db) {
            const int vrow = MP_ADBG(2) ? 0 : kp * 32;
            const bf16x8_t vh = Vhr.cols(db, vrow), vl = Vhr.cols(db, vrow, dlo);
            if (MP_ADBG(8)) { o[db][0] += __builtin_bit_cast(float, (int)vh[0] + (int)vl[1] + (int)bph[0] + (int)bpl[0]); continue; }
            MP_MFMA3(o[db], vh, vl, bph, bpl);
          }
        }
      }
      if (tq < T && !(MP_ADBG(16) && sum != 12345.f)) {
        const float inv = 1.0f / sum;
        const long oo = ((long)(b * T + tq) * J + j) * C + h * D;
#pragma unroll
        for (int db = 0; db < DB; ++db) store4_o<F8O>(out_hi + oo + 16 * db + 4 * g, out_lo + oo + 16 * db + 4 * g, o[db], inv);
        if (g == 0) lse[(long)unit * T + tq] = (mx + __log2f(sum)) * 0.6931471805599453f;
      }
    }
    if (next >= nunits) break;
    if (!q_fetched) fetch_q(next);
    unit = next;
  }
}

// Two-phase form of the split-precision temporal forward for head dim 64 and windows of more than 128 frames (the benchmark's shape).
// The LDS holds region A = (K_hi, K_lo) and region B = (V_hi, V_lo); a unit is computed as
//   phase 1: scores + softmax of BOTH 16-query strips of a wave from region A   (meanwhile: V of this unit arrives in region B)
//   phase 2: O = P V of both strips from region B                              (meanwhile: K of the next unit arrives in region A)
// with the images fetched by direct-to-LDS DMA (global_load_lds_dwordx4; the XOR swizzle of stage_rows applied to the source address),
// so no register is spent on prefetching and the scores of both strips can stay live: every K / V fragment read from LDS now serves two
// strips (half the LDS fragment traffic of the one-strip-at-a-time kernel above), the staging of a unit is hidden behind the other
// phase, and the output tile is transposed through a wave-private LDS buffer so that every store instruction writes whole 128-byte
// rows (the accumulator layout gave 32-byte pieces of 16 rows that lie J*C*2 bytes apart).
__device__ __forceinline__ void tm_dma_region(char* __restrict__ region, const bf16* __restrict__ p_hi, const bf16* __restrict__ p_lo, long rs3, int T,
                                              int rows, int lane, int wave, int nw) {
  typedef __attribute__((address_space(3))) void* lptr;
  typedef const __attribute__((address_space(1))) void* gptr;
  const int per_img = rows >> 3;                     // DMA instructions per image: 8 rows (1 KiB) each
  for (int n = wave; n < 2 * per_img; n += nw) {     // wave-uniform instruction index
    const int img = n >= per_img, q = n - img * per_img;
    const int r = q * 8 + (lane >> 3), c = (lane & 7) ^ (r & 7);
    const bf16* src = (img ? p_lo : p_hi) + (long)min(r, T - 1) * rs3 + c * 8;      // frames past T repeat the last frame (finite; their
    __builtin_amdgcn_global_load_lds((gptr)src, (lptr)(region + n * 1024), 16, 0, 0);   // probabilities are exactly 0)
  }
}

// FULL: the window fills all NTILE key tiles (T > 240: the benchmark's 243 frames) - the tile count is a compile-time constant, so the unrolled
// tile loops carry no wave-uniform branches and each phase is one basic block for the instruction scheduler
template <bool FULL, bool F8O = false>
__global__ __launch_bounds__(512) void attn_tmfma_fwd_x3p_kernel(const bf16* __restrict__ qkv_hi, const bf16* __restrict__ qkv_lo,
                                                                  bf16* __restrict__ out_hi, bf16* __restrict__ out_lo,
                                                                  float* __restrict__ lse, int nunits, int T, int J, int C, int H, float scale) {
  // no contraction across statements: with the whole phase in one basic block (FULL) hipcc would fuse the score scaling into the subtraction of the
  // row maximum (one rounding less); kept off so that this kernel, its FULL = false form and the one-strip kernel stay bit-compatible (tested)
#pragma clang fp contract(off)
  extern __shared__ __attribute__((aligned(16))) char sm[];
  constexpr int D = 64, ROWB = 128, KS = 2, DB = 4, NW = 8, OPITCH = 144;
  const int rows = FULL ? TP : (T + 31) & ~31;
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, l15 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long rs3 = (long)J * 3 * C;
  const int ntile = FULL ? NTILE : (T + 15) >> 4;
  const int img_bytes = rows * ROWB;
  char* const regA = sm;                              // K_hi | K_lo
  char* const regB = sm + 2 * img_bytes;              // V_hi | V_lo
  char* const obuf = sm + 4 * img_bytes + wave * (16 * OPITCH);
  ImgRd<D> Khr, Vhr;
  Khr.init(regA, lane);
  Vhr.init(regB, lane);
  const int dlo = __builtin_amdgcn_readfirstlane(img_bytes);
  const float scale2 = scale * 1.4426950408889634f;
  auto unit_base = [&](int unit) -> long {
    const int h = unit % H, bj = unit / H, j = bj % J, b = bj / J;
    return ((long)b * T * J + j) * 3 * C + h * D;
  };
  bf16x8_t qh[2][KS], ql[2][KS];
  auto fetch_q = [&](int unit) {
    const long qoff = unit_base(unit);
#pragma unroll
    for (int sidx = 0; sidx < 2; ++sidx) {
      const int tq = min((wave + sidx * NW) * 16 + l15, T - 1);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        qh[sidx][ks] = *reinterpret_cast<const bf16x8_t*>(qkv_hi + qoff + (long)tq * rs3 + 32 * ks + 8 * g);
        ql[sidx][ks] = *reinterpret_cast<const bf16x8_t*>(qkv_lo + qoff + (long)tq * rs3 + 32 * ks + 8 * g);
      }
    }
  };
  int unit = blockIdx.x;
  if (unit >= nunits) return;
  {
    const long qoff = unit_base(unit);
    tm_dma_region(regA, qkv_hi + qoff + C, qkv_lo + qoff + C, rs3, T, rows, lane, wave, NW);
  }
  fetch_q(unit);
  const bool two = wave + NW < ntile;                 // this wave's second strip exists
  while (true) {
    const long qoff = unit_base(unit);
    const int h = unit % H, bj = unit / H, j = bj % J, b = bj / J;
    // K(unit) has landed for every wave; nobody reads region B (V of the previous unit) any more
    __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_s_barrier();
    tm_dma_region(regB, qkv_hi + qoff + 2 * C, qkv_lo + qoff + 2 * C, rs3, T, rows, lane, wave, NW);
    // ---- phase 1: scores of both strips (every K fragment serves both), exact softmax ----
    f32x4 s[2][NTILE];
    float mx[2] = {-INFINITY, -INFINITY};
#pragma unroll
    for (int kt = 0; kt < NTILE; ++kt) {
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
      if (kt < ntile) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const bf16x8_t kh = Khr.rows(ks, kt * 16), kl = Khr.rows(ks, kt * 16, dlo);
          MP_MFMA3(a0, kh, kl, qh[0][ks], ql[0][ks]);
          MP_MFMA3(a1, kh, kl, qh[1][ks], ql[1][ks]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { a0[r] = smul(a0[r], scale2); a1[r] = smul(a1[r], scale2); }      // not a0 *= scale2: see sfma()
        if (kt == ntile - 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (kt * 16 + 4 * g + r >= T) { a0[r] = -INFINITY; a1[r] = -INFINITY; }
        }
        mx[0] = fmaxf(fmaxf(mx[0], fmaxf(a0[0], a0[1])), fmaxf(a0[2], a0[3]));
        mx[1] = fmaxf(fmaxf(mx[1], fmaxf(a1[0], a1[1])), fmaxf(a1[2], a1[3]));
      } else {
        a0 = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        a1 = a0;
      }
      s[0][kt] = a0;
      s[1][kt] = a1;
    }
    float sum[2];
#pragma unroll
    for (int sidx = 0; sidx < 2; ++sidx) {
      mx[sidx] = group_max(mx[sidx]);
      float sm_ = 0.f;
#pragma unroll
      for (int kt = 0; kt < NTILE; ++kt) {
        if (kt < ntile) {
          f32x4 e;
#pragma unroll
          for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(ssub(s[sidx][kt][r], mx[sidx]));
          s[sidx][kt] = e;
          sm_ += (e[0] + e[1]) + (e[2] + e[3]);
        } else {
          s[sidx][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      sum[sidx] = group_sum(sm_);
    }
    // V(unit) has landed for every wave; nobody reads region A (K of this unit) any more
    __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_s_barrier();
    const int next = unit + gridDim.x;
    if (next < nunits) {
      const long noff = unit_base(next);
      tm_dma_region(regA, qkv_hi + noff + C, qkv_lo + noff + C, rs3, T, rows, lane, wave, NW);
      fetch_q(next);
    }
    // ---- phase 2: O = P V of both strips (every V fragment serves both) ----
    f32x4 o[2][DB];
#pragma unroll
    for (int sidx = 0; sidx < 2; ++sidx)
#pragma unroll
      for (int db = 0; db < DB; ++db) o[sidx][db] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kp = 0; kp < NTILE / 2; ++kp) {
      if (2 * kp < ntile) {
        bf16x8_t p0h, p0l, p1h, p1l;
        pack_acc_x2(s[0][2 * kp], s[0][2 * kp + 1], p0h, p0l);
        pack_acc_x2(s[1][2 * kp], s[1][2 * kp + 1], p1h, p1l);
#pragma unroll
        for (int db = 0; db < DB; ++db) {
          const bf16x8_t vh = Vhr.cols(db, kp * 32), vl = Vhr.cols(db, kp * 32, dlo);
          MP_MFMA3(o[0][db], vh, vl, p0h, p0l);
          MP_MFMA3(o[1][db], vh, vl, p1h, p1l);
        }
      }
    }
    // ---- output: per strip and plane, the 16 x 64 tile goes through the wave-private buffer and leaves as whole 128-byte rows ----
#pragma unroll
    for (int sidx = 0; sidx < 2; ++sidx) {
      if (sidx == 1 && !two) break;
      const int qt = wave + sidx * NW;
      const float inv = 1.0f / sum[sidx];
      uint2 hv[DB], lv[DB];
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        if constexpr (F8O) pack4_f16f8(make_float4(o[sidx][db][0] * inv, o[sidx][db][1] * inv, o[sidx][db][2] * inv, o[sidx][db][3] * inv), hv[db], lv[db]);
        else {
          split_bf16x2(o[sidx][db][0] * inv, o[sidx][db][1] * inv, hv[db].x, lv[db].x);
          split_bf16x2(o[sidx][db][2] * inv, o[sidx][db][3] * inv, hv[db].y, lv[db].y);
        }
      }
#pragma unroll
      for (int plane = 0; plane < 2; ++plane) {
#pragma unroll
        for (int db = 0; db < DB; ++db) *reinterpret_cast<uint2*>(obuf + l15 * OPITCH + 32 * db + 8 * g) = plane ? lv[db] : hv[db];
        bf16* const dst = plane ? out_lo : out_hi;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
          const int row = pass * 8 + (lane >> 3), tq = qt * 16 + row;
          const uint4 v = *reinterpret_cast<const uint4*>(obuf + row * OPITCH + (lane & 7) * 16);
          if (tq < T) *reinterpret_cast<uint4*>(dst + ((long)(b * T + tq) * J + j) * C + h * D + (lane & 7) * 8) = v;
        }
      }
      const int tq = qt * 16 + l15;
      if (g == 0 && tq < T) lse[(long)unit * T + tq] = (mx[sidx] + __log2f(sum[sidx])) * 0.6931471805599453f;
    }
    if (next >= nunits) break;
    unit = next;
  }
}

// (Round 4, the idea round 3 left open - matrix work of one part of a wave's unit under the softmax arithmetic of another: the cheap form of it,
// taking the exponentials tile pair by tile pair inside phase 2, ahead of that pair's P V products (same arithmetic, same order, bit-identical
// outputs; where this wave's and its SIMD partner's matrix instructions are in flight instead of in a block of their own) was built and measured:
// 909.7 / 878.1 us -> 887.3 / 879.4 us per layer at the benchmark's batch (alternating, same box; profiles/r04_probes/attn_exp_ab.log) - nothing.
// Where the vector work is issued does not matter to this kernel; with the SQ counters of round 3 (waves parked 32 %, issue-stalled 34 %) that
// leaves the two barriers per unit and the exposed LDS / DMA round trips, not the instruction mix.  Not kept.)
// (A two-phase form of the bf16 temporal BACKWARD - (K, V) and (Q, dO) regions by DMA behind the other pass, eight waves with two strips
// each so that every LDS fragment serves both - was built and measured in round 3: bit-compatible, half the LDS fragment traffic, and no
// faster: 1100-1130 us against 1123 us isolated at the benchmark's batch, 178.9 against 177.7 ms per step.  The SQ counters of both
// forms show why: waves parked on s_waitcnt / barriers 45 % of their cycles, issue stalls 32 %, LDS busy 3 %, matrix cores 27 % - the
// kernel waits for its 128-byte rows, which lie J * 3C * 2 bytes apart in the token-major qkv layout (2.4 TB/s effective for every
// temporal kernel, where the spatial kernels - whole contiguous frames - reach 5 TB/s).  Not kept.)
static int g_attn_two_phase = 1;           // mp_set_option("attn_two_phase", 0): the one-strip-at-a-time kernel for every shape (A/B timing, tests)
void attn_two_phase(int on) { g_attn_two_phase = on; }

// spatial: workgroup per frame, wave per head; the frame's hi and lo qkv blocks (2 x N x 6C bytes, contiguous in HBM) are both staged.
// Persistent with register prefetch of the next frame's blocks like the temporal kernel (105 KiB of LDS at C = 512: one workgroup per CU).
template <int D, bool F8O = false>
__global__ __launch_bounds__(512) void attn_smfma_fwd_x3_kernel(const bf16* __restrict__ qkv_hi, const bf16* __restrict__ qkv_lo,
                                                                 bf16* __restrict__ out_hi, bf16* __restrict__ out_lo, int nframes, int N, int C,
                                                                 int H, float scale) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  constexpr int KS = ACfg<D>::KS, DB = ACfg<D>::DB, PER = 7;            // 2 x PER 16-byte chunks per thread: N * 6C / 16 <= PER * threads (launcher)
  const int tid = threadIdx.x, lane = tid & 63, h = __builtin_amdgcn_readfirstlane(tid >> 6), nthreads = (int)blockDim.x;
  const int g = lane >> 4, l15 = lane & 15;
  const int pitch = 6 * C + 16, cpr = (6 * C) >> 4, nchunk = N * cpr;
  char* sh = sm;
  char* sl = sm + N * pitch;
  uint4 pre[2][PER];
  auto fetch = [&](int f) {
    const char* bh = reinterpret_cast<const char*>(qkv_hi + (long)f * N * 3 * C);
    const char* bl = reinterpret_cast<const char*>(qkv_lo + (long)f * N * 3 * C);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int idx = tid + i * nthreads;
      uint4 a = make_uint4(0u, 0u, 0u, 0u), c = a;
      if (idx < nchunk) { a = *reinterpret_cast<const uint4*>(bh + (long)idx * 16); c = *reinterpret_cast<const uint4*>(bl + (long)idx * 16); }
      pre[0][i] = a; pre[1][i] = c;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int idx = tid + i * nthreads, r = idx / cpr, k = idx - r * cpr;
      if (idx < nchunk) {
        *reinterpret_cast<uint4*>(sh + r * pitch + k * 16) = pre[0][i];
        *reinterpret_cast<uint4*>(sl + r * pitch + k * 16) = pre[1][i];
      }
    }
  };
  const int oq = h * D * 2, ok = 2 * C + oq, ov = 4 * C + oq;
  int f = blockIdx.x;
  if (f >= nframes) return;
  fetch(f);
  while (true) {
    __syncthreads();
    commit();
    __syncthreads();
    const int next = f + gridDim.x;
    if (next < nframes) fetch(next);
    bf16x8_t kfh[2][KS], kfl[2][KS];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        kfh[t][ks] = frag_tok<D>(sh, pitch, ok, 16 * t, ks, lane, N);
        kfl[t][ks] = frag_tok<D>(sl, pitch, ok, 16 * t, ks, lane, N);
      }
    bf16x8_t vTh[DB], vTl[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      vTh[db] = frag_tokT<D>(sh, pitch, ov, db, lane, N);
      vTl[db] = frag_tokT<D>(sl, pitch, ov, db, lane, N);
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      if (qt * 16 >= N) break;
      bf16x8_t qfh[KS], qfl[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        qfh[ks] = frag_tok<D>(sh, pitch, oq, 16 * qt, ks, lane, N);
        qfl[ks] = frag_tok<D>(sl, pitch, oq, 16 * qt, ks, lane, N);
      }
      f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        MP_MFMA3(s0, kfh[0][ks], kfl[0][ks], qfh[ks], qfl[ks]);
        MP_MFMA3(s1, kfh[1][ks], kfl[1][ks], qfh[ks], qfl[ks]);
      }
      col_softmax(s0, s1, g, N, scale);
      bf16x8_t bph, bpl;
      pack_acc_x2(s0, s1, bph, bpl);
      const int tq = qt * 16 + l15;
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        MP_MFMA3(o, vTh[db], vTl[db], bph, bpl);
        if (tq < N) {
          const long oo = ((long)f * N + tq) * C + h * D + 16 * db + 4 * g;
          store4_o<F8O>(out_hi + oo, out_lo + oo, o, 1.0f);
        }
      }
    }
    if (next >= nframes) break;
    f = next;
  }
}

// planar <-> fp32 (fallback path of the split-precision attention for shapes the MFMA kernels do not cover; small models only)
__global__ void join_planes_kernel(const bf16p* __restrict__ hi, long lo_off, float* __restrict__ out, long n4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n4) st4(out + 4 * i, ld4(hi + 4 * i, lo_off));
}
__global__ void split_planes_kernel(const float* __restrict__ in, bf16p* __restrict__ hi, long lo_off, long n4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n4) st4(hi + 4 * i, ld4(in + 4 * i), lo_off);
}
int join_planes(const bf16* hi, const bf16* lo, float* out, long n, hipStream_t st) {
  MP_CHECK(n % 4 == 0, MP_ERR_ARG, "join_planes: n %% 4");
  hipLaunchKernelGGL(join_planes_kernel, dim3(cdiv(n / 4, 256)), dim3(256), 0, st, reinterpret_cast<const bf16p*>(hi), (long)(lo - hi), out, n / 4);
  MP_LAUNCH_CHECK();
  return MP_OK;
}
int split_planes(const float* in, bf16* hi, bf16* lo, long n, hipStream_t st) {
  MP_CHECK(n % 4 == 0, MP_ERR_ARG, "split_planes: n %% 4");
  hipLaunchKernelGGL(split_planes_kernel, dim3(cdiv(n / 4, 256)), dim3(256), 0, st, in, reinterpret_cast<bf16p*>(hi), (long)(lo - hi), n / 4);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

static int num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, c = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
    n = c;
  }
  return n;
}
bool attn_tmfma_supported(int T, int D);
bool attn_x3_needs_scratch(int temporal, int T, int J, int C, int H) {
  const int D = C / H;
  if (temporal) return !(attn_tmfma_supported(T, D) && C % 8 == 0);
  // spatial: the frame's two planes must fit the LDS and the register prefetch (7 x 16 B per thread and plane)
  return !(attn_smfma_supported(J, D, H) && C % 8 == 0 && 2L * J * (6 * C + 16) <= 160 * 1024 && (long)J * (6 * C / 16) <= 7L * H * 64);
}

int attn_spatial_fwd_x3(const bf16* qkv_hi, const bf16* qkv_lo, bf16* out_hi, bf16* out_lo, float* scratch, int B, int T, int J, int C, int H,
                        hipStream_t st, int out_f16f8) {
  MP_CHECK(C % H == 0 && C % 4 == 0, MP_ERR_ARG, "attn_spatial_fwd_x3: C=%d H=%d", C, H);
  const int D = C / H;
  const long M = (long)B * T * J;
  MP_CHECK(!out_f16f8 || (D == 64 && !attn_x3_needs_scratch(0, T, J, C, H)), MP_ERR_ARG, "attn_spatial_fwd_x3: f16f8 output needs head dim 64 and the MFMA kernel (J=%d C=%d H=%d)", J, C, H);
  if (attn_x3_needs_scratch(0, T, J, C, H)) {
    MP_CHECK(scratch != nullptr, MP_ERR_ARG, "attn_spatial_fwd_x3: J=%d D=%d H=%d needs the fp32 scratch", J, D, H);
    int rc = join_planes(qkv_hi, qkv_lo, scratch, 3 * M * C, st);
    if (rc) return rc;
    rc = attn_spatial_fwd(scratch, scratch + 3 * M * C, 0, B, T, J, C, H, st);
    if (rc) return rc;
    return split_planes(scratch + 3 * M * C, out_hi, out_lo, M * C, st);
  }
  const float scale = attn_qk_scale(D);
  const size_t lds = 2 * (size_t)J * (6 * C + 16);
  MP_CHECK(lds <= 160 * 1024, MP_ERR_ARG, "attn_spatial_fwd_x3: frame block of %zu bytes exceeds the LDS", lds);
  MP_CHECK((long)J * (6 * C / 16) <= 7L * H * 64, MP_ERR_ARG, "attn_spatial_fwd_x3: frame block too large for the register prefetch (J=%d C=%d H=%d)", J, C, H);
  const int nframes = B * T;
  const int per_cu = (int)max((size_t)1, min((size_t)4, (size_t)(160 * 1024) / lds));
  const int grid = min(nframes, num_cus() * per_cu);
  if (D == 64) {
    static bool attr_set = false;
    if (!attr_set) {
      MP_HIP(hipFuncSetAttribute((const void*)attn_smfma_fwd_x3_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_set = true;
    }
    if (out_f16f8) {
      static bool attr8_set = false;
      if (!attr8_set) {
        MP_HIP(hipFuncSetAttribute((const void*)attn_smfma_fwd_x3_kernel<64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr8_set = true;
      }
      hipLaunchKernelGGL((attn_smfma_fwd_x3_kernel<64, true>), dim3(grid), dim3(H * 64), lds, st, qkv_hi, qkv_lo, out_hi, out_lo, nframes, J, C, H, scale);
    } else
    hipLaunchKernelGGL(attn_smfma_fwd_x3_kernel<64>, dim3(grid), dim3(H * 64), lds, st, qkv_hi, qkv_lo, out_hi, out_lo, nframes, J, C, H, scale);
  } else {
    static bool attr_set = false;
    if (!attr_set) {
      MP_HIP(hipFuncSetAttribute((const void*)attn_smfma_fwd_x3_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_set = true;
    }
    hipLaunchKernelGGL(attn_smfma_fwd_x3_kernel<16>, dim3(grid), dim3(H * 64), lds, st, qkv_hi, qkv_lo, out_hi, out_lo, nframes, J, C, H, scale);
  }
  MP_LAUNCH_CHECK();
  return MP_OK;
}

// (-DATTN_X3P_NO_FULL: A/B builds without the full-window specialisations)
#ifdef ATTN_X3P_NO_FULL
#define ATTN_X3P_FULL_GUARD && false
#else
#define ATTN_X3P_FULL_GUARD
#endif
template <int D, bool F8O = false>
static int launch_tmfma_fwd_x3(const bf16* qkv_hi, const bf16* qkv_lo, bf16* out_hi, bf16* out_lo, float* lse, int units, int T, int J, int C,
                               int H, float scale, hipStream_t st) {
  const int ntile = (T + 15) >> 4;
  const int rows = (T + 31) & ~31;
  // waves: two query strips each, and enough threads that an image is at most 4 chunks per thread (register prefetch)
  const int waves = min(8, max((ntile + 1) / 2, cdiv((long)rows * ACfg<D>::CH, 256)));
  const size_t lds = 4 * (size_t)rows * ACfg<D>::ROWB;
  static bool attr_set = false;
  if (!attr_set) {
    MP_HIP(hipFuncSetAttribute((const void*)attn_tmfma_fwd_x3_kernel<D, false, F8O>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * TP * ACfg<D>::ROWB)));
    MP_HIP(hipFuncSetAttribute((const void*)attn_tmfma_fwd_x3_kernel<D, true, F8O>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * TP * ACfg<D>::ROWB)));
    attr_set = true;
  }
  const int per_cu = (int)max((size_t)1, min((size_t)4, (size_t)(160 * 1024) / lds));
  const int grid = min(units, num_cus() * per_cu);
#ifdef MP_GEMM_DIAG
  static const int dbg = [] { const char* e = getenv("MANIPOSE_ATTN_DEBUG"); return e ? atoi(e) : 0; }();
#else
  constexpr int dbg = 0;
#endif
  if (ntile == NTILE ATTN_X3P_FULL_GUARD)
    hipLaunchKernelGGL((attn_tmfma_fwd_x3_kernel<D, true, F8O>), dim3(grid), dim3(64 * waves), lds, st, qkv_hi, qkv_lo, out_hi, out_lo, lse, units, T, J, C, H, scale, dbg);
  else
    hipLaunchKernelGGL((attn_tmfma_fwd_x3_kernel<D, false, F8O>), dim3(grid), dim3(64 * waves), lds, st, qkv_hi, qkv_lo, out_hi, out_lo, lse, units, T, J, C, H, scale, dbg);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

int attn_temporal_fwd_x3(const bf16* qkv_hi, const bf16* qkv_lo, bf16* out_hi, bf16* out_lo, float* lse, float* scratch, int B, int T, int J,
                         int C, int H, hipStream_t st, int out_f16f8) {
  MP_CHECK(C % H == 0 && C % 4 == 0, MP_ERR_ARG, "attn_temporal_fwd_x3: C=%d H=%d", C, H);
  const int D = C / H;
  const long M = (long)B * T * J;
  MP_CHECK(!out_f16f8 || (D == 64 && !attn_x3_needs_scratch(1, T, J, C, H)), MP_ERR_ARG, "attn_temporal_fwd_x3: f16f8 output needs head dim 64 and the MFMA kernel (T=%d C=%d H=%d)", T, C, H);
  if (attn_x3_needs_scratch(1, T, J, C, H)) {
    MP_CHECK(scratch != nullptr, MP_ERR_ARG, "attn_temporal_fwd_x3: T=%d D=%d needs the fp32 scratch", T, D);
    int rc = join_planes(qkv_hi, qkv_lo, scratch, 3 * M * C, st);
    if (rc) return rc;
    rc = attn_temporal_fwd(scratch, scratch + 3 * M * C, lse, 0, B, T, J, C, H, st);
    if (rc) return rc;
    return split_planes(scratch + 3 * M * C, out_hi, out_lo, M * C, st);
  }
  const float scale = attn_qk_scale(D);
  const int units = B * J * H;
  if (D == 64 && T > 128 && (g_attn_two_phase & 1) && (C * 2) % 128 == 0 && (3L * C * 2) % 16 == 0) {
    const int rows = (T + 31) & ~31;
    const size_t lds = 4 * (size_t)rows * 128 + 8 * 16 * 144;
    static bool attr_set = false;
    if (!attr_set) {
      MP_HIP(hipFuncSetAttribute((const void*)attn_tmfma_fwd_x3p_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * TP * 128 + 8 * 16 * 144)));
      MP_HIP(hipFuncSetAttribute((const void*)attn_tmfma_fwd_x3p_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * TP * 128 + 8 * 16 * 144)));
      MP_HIP(hipFuncSetAttribute((const void*)attn_tmfma_fwd_x3p_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * TP * 128 + 8 * 16 * 144)));
      MP_HIP(hipFuncSetAttribute((const void*)attn_tmfma_fwd_x3p_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * TP * 128 + 8 * 16 * 144)));
      attr_set = true;
    }
    const int grid = min(units, num_cus());
    const bool full = ((T + 15) >> 4) == NTILE ATTN_X3P_FULL_GUARD;
    if (out_f16f8) {
      if (full) hipLaunchKernelGGL((attn_tmfma_fwd_x3p_kernel<true, true>), dim3(grid), dim3(512), lds, st, qkv_hi, qkv_lo, out_hi, out_lo, lse, units, T, J, C, H, scale);
      else hipLaunchKernelGGL((attn_tmfma_fwd_x3p_kernel<false, true>), dim3(grid), dim3(512), lds, st, qkv_hi, qkv_lo, out_hi, out_lo, lse, units, T, J, C, H, scale);
    } else if (full)
      hipLaunchKernelGGL(attn_tmfma_fwd_x3p_kernel<true>, dim3(grid), dim3(512), lds, st, qkv_hi, qkv_lo, out_hi, out_lo, lse, units, T, J, C, H, scale);
    else
      hipLaunchKernelGGL(attn_tmfma_fwd_x3p_kernel<false>, dim3(grid), dim3(512), lds, st, qkv_hi, qkv_lo, out_hi, out_lo, lse, units, T, J, C, H, scale);
    MP_LAUNCH_CHECK();
    return MP_OK;
  }
  if (D == 64 && out_f16f8) return launch_tmfma_fwd_x3<64, true>(qkv_hi, qkv_lo, out_hi, out_lo, lse, units, T, J, C, H, scale, st);
  if (D == 64) return launch_tmfma_fwd_x3<64>(qkv_hi, qkv_lo, out_hi, out_lo, lse, units, T, J, C, H, scale, st);
  return launch_tmfma_fwd_x3<16>(qkv_hi, qkv_lo, out_hi, out_lo, lse, units, T, J, C, H, scale, st);
}

bool attn_tmfma_supported(int T, int D) { return T <= TP && (D == 64 || D == 16); }

template <int D, int NTC>
static int launch_tmfma_fwd(const bf16* qkv, bf16* out, float* lse, int units, int T, int J, int C, int H, float scale, hipStream_t st) {
  // waves per workgroup: two 16-query strips per wave up to 128 frames (more, smaller workgroups per CU overlap their load / compute /
  // store phases: T=81 602 -> 442 us, T=128 443 -> 428 us, T=27 411 -> 387 us at the bench's token count), 8 waves beyond
  const int ntile = (T + 15) >> 4;
  const int waves = NTC ? 8 : (ntile <= 8 ? (ntile + 1) / 2 : 8);
  const int rows = NTC ? TP : (T + 31) & ~31;
  const size_t lds = 2 * (size_t)rows * ACfg<D>::ROWB;
  static bool attr_set = false;
  if (!attr_set) {
    MP_HIP(hipFuncSetAttribute((const void*)attn_tmfma_fwd_kernel<D, NTC>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)(2 * TP * ACfg<D>::ROWB)));
    attr_set = true;
  }
  hipLaunchKernelGGL((attn_tmfma_fwd_kernel<D, NTC>), dim3(units), dim3(64 * waves), lds, st, qkv, out, lse, T, J, C, H, scale);
  MP_LAUNCH_CHECK();
  return MP_OK;
}
int attn_tmfma_fwd(const bf16* qkv, bf16* out, float* lse, int B, int T, int J, int C, int H, hipStream_t st) {
  const int D = C / H;
  MP_CHECK(attn_tmfma_supported(T, D) && C % 8 == 0, MP_ERR_ARG, "attn_tmfma_fwd: T=%d D=%d unsupported", T, D);
  const float scale = attn_qk_scale(D);
  const int units = B * J * H;
  // (run-time tile count only: with a compile-time count the scheduler hoists across the whole score strip and spills)
  if (D == 64) return launch_tmfma_fwd<64, 0>(qkv, out, lse, units, T, J, C, H, scale, st);
  return launch_tmfma_fwd<16, 0>(qkv, out, lse, units, T, J, C, H, scale, st);
}

template <int D, int NTC>
static int launch_tmfma_bwd(const bf16* qkv, const bf16* out, const bf16* dout, const float* lse, bf16* dqkv, int units, int T, int J, int C,
                            int H, float scale, int dbg, hipStream_t st) {
  // waves per workgroup: one per 16-frame strip, except 4 for 5-7 strips (T=81: 971 -> 906 us; 8 strips and more measured best at one each)
  const int ntile = (T + 15) >> 4;
  const int waves = NTC ? 16 : ((ntile >= 5 && ntile <= 7) ? 4 : min(16, ntile));
  const int rows = NTC ? TP : (T + 31) & ~31;
  const size_t lds = 4 * (size_t)rows * ACfg<D>::ROWB + 2 * rows * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    MP_HIP(hipFuncSetAttribute((const void*)attn_tmfma_bwd_kernel<D, NTC, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)(4 * TP * ACfg<D>::ROWB + 2 * TP * sizeof(float))));
    MP_HIP(hipFuncSetAttribute((const void*)attn_tmfma_bwd_kernel<D, NTC, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)(4 * TP * ACfg<D>::ROWB + 2 * TP * sizeof(float))));
    attr_set = true;
  }
  if (g_grad_f16 != nullptr)
    hipLaunchKernelGGL((attn_tmfma_bwd_kernel<D, NTC, true>), dim3(units), dim3(64 * waves), lds, st, qkv, out, dout, lse, dqkv, T, J, C, H, scale, dbg, g_grad_f16, g_out_f16);
  else
    hipLaunchKernelGGL((attn_tmfma_bwd_kernel<D, NTC, false>), dim3(units), dim3(64 * waves), lds, st, qkv, out, dout, lse, dqkv, T, J, C, H, scale, dbg, g_grad_f16, g_out_f16);
  MP_LAUNCH_CHECK();
  return MP_OK;
}
int attn_tmfma_bwd(const bf16* qkv, const bf16* out, const bf16* dout, const float* lse, bf16* dqkv, int B, int T, int J, int C, int H,
                   hipStream_t st) {
  const int D = C / H;
  MP_CHECK(attn_tmfma_supported(T, D) && C % 8 == 0, MP_ERR_ARG, "attn_tmfma_bwd: T=%d D=%d unsupported", T, D);
  const float scale = attn_qk_scale(D);
  const int units = B * J * H;
  // timing ablations (1 staging only, 2 no dK/dV pass: results wrong by design) exist in the diagnostics build only
#ifdef MP_GEMM_DIAG
  static const int dbg = [] { const char* e = getenv("MANIPOSE_ATTN_DEBUG"); return e ? atoi(e) : 0; }();
#else
  constexpr int dbg = 0;
#endif
  const bool full = T > 240;
  if (D == 64) return full ? launch_tmfma_bwd<64, 16>(qkv, out, dout, lse, dqkv, units, T, J, C, H, scale, dbg, st)
                           : launch_tmfma_bwd<64, 0>(qkv, out, dout, lse, dqkv, units, T, J, C, H, scale, dbg, st);
  return full ? launch_tmfma_bwd<16, 16>(qkv, out, dout, lse, dqkv, units, T, J, C, H, scale, dbg, st)
              : launch_tmfma_bwd<16, 0>(qkv, out, dout, lse, dqkv, units, T, J, C, H, scale, dbg, st);
}

}  // namespace mp
