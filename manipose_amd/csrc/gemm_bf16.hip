// bf16 matrix-core GEMM family (v_mfma_f32_16x16x32_bf16, fp32 accumulate): the throughput precision of the
// MixSTE Linear layers (forward X W^T, dgrad dY W, wgrad dY^T X) with the same fused epilogues as gemm_f32.hip.
//
// Tile 128x128x64 per 256-thread workgroup; 4 waves as 2x2, each wave 64x64 = 4x4 MFMA tiles x 2 k-steps.
// An operand is either
//   "N" (stored [out][red], reduction contiguous: activations in the forward/dgrad, weights in the forward):
//       staged as 16-byte chunks (8 consecutive k of one row) into a k8-major LDS image
//       chunk(row, kg) at kg*128 + (row ^ (kg & 7)); full 128-B lines per row from HBM, conflict-free
//       ds_write_b128 (8 lanes = 8 kg of one row) and conflict-free ds_read_b128 fragment reads
//       (lane -> row l&15, kg = 4*kstep + (l>>4): the b128 lane groups pair kg with kg+1, kg even).
//   "T" (stored [red][out], output index contiguous: weights in the dgrad, BOTH operands in the wgrad):
//       staged untransposed ([red][128 out], 288-B rows) and consumed through the gfx950 hardware transpose
//       read ds_read_b64_tr_b16 (two per fragment), so no transposed copies of activations or weights ever
//       exist in HBM.
// fp32 operands (the fp32 residual-gradient stream) are converted to bf16 while staging.
#include <stdlib.h>
#include "common.h"
#include "kernels.h"

namespace mp {

// Timing ablations (skip the MFMA stage / the operand DMA / the epilogue: results wrong by design) exist only in the diagnostics build
// (MP_DIAG=1 build.sh -> libmanipose_hip_diag.so); in the product library the bits are a compile-time zero and no environment
// variable can switch them on.
#ifdef MP_GEMM_DIAG
#define MP_DBG(g, bit) ((g).debug & (bit))
#else
#define MP_DBG(g, bit) 0
#endif

typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef short bf16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int GBM = 128, GBN = 128, GBK = 64;
constexpr int T_ROWB = 288;                    // bytes per reduction row of a "T" image (256 + 32 pad)
constexpr int OP_BYTES = GBK * T_ROWB;         // 18432 >= 16384 ("N" image)

struct Chunk { uint4 v; };                     // 8 bf16

__device__ __forceinline__ uint4 pack8(const float4& a, const float4& b) {
  uint4 r;
  r.x = pack_bf16x2(a.x, a.y); r.y = pack_bf16x2(a.z, a.w); r.z = pack_bf16x2(b.x, b.y); r.w = pack_bf16x2(b.z, b.w);
  return r;
}
__device__ __forceinline__ uint4 load_chunk(const bf16* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ uint4 load_chunk(const float* p) { return pack8(ld4(p), ld4(p + 4)); }

// ---- global -> registers: 4 chunks (16 B of bf16 each) per thread per operand ----
template <typename T, int TR>
__device__ __forceinline__ void load_op(const T* __restrict__ P, long ld, int out0, int r0, int OUT, int r_end, int tid,
                                        uint4 (&v)[4]) {
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int idx = tid + 256 * it;
    uint4 x = make_uint4(0u, 0u, 0u, 0u);
    if (TR == 0) {
      const int row = idx >> 3, kg = idx & 7;
      const int o = out0 + row, r = r0 + kg * 8;
      if (o < OUT && r < r_end) x = load_chunk(P + (long)o * ld + r);
    } else {
      const int r = r0 + (idx >> 4), o = out0 + (idx & 15) * 8;
      if (r < r_end && o < OUT) x = load_chunk(P + (long)r * ld + o);
    }
    v[it] = x;
  }
}

template <int TR>
__device__ __forceinline__ void store_op(char* __restrict__ S, int tid, const uint4 (&v)[4]) {
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int idx = tid + 256 * it;
    if (TR == 0) {
      const int row = idx >> 3, kg = idx & 7;
      *reinterpret_cast<uint4*>(S + ((kg * 128 + (row ^ kg)) << 4)) = v[it];
    } else {
      const int r = idx >> 4, oc = idx & 15;
      *reinterpret_cast<uint4*>(S + r * T_ROWB + oc * 16) = v[it];
    }
  }
}

// fragment of the 16 output indices [ob, ob+16) for k-step ks (32 reduction indices) of a staged operand
template <int TR>
__device__ __forceinline__ bf16x8_t read_frag(const char* __restrict__ S, int ob, int ks, int lane) {
  if (TR == 0) {
    const int row = ob + (lane & 15), kg = ks * 4 + (lane >> 4);
    return *reinterpret_cast<const bf16x8_t*>(S + ((kg * 128 + (row ^ kg)) << 4));
  } else {
    // ds_read_b64_tr_b16: lane 4q+p of a 16-lane group points at row q, columns 4p..4p+3 of a 4 x 16 block and
    // receives column (lane & 15) of the 4 rows.  Block rows = reduction indices, block columns = output indices.
    const int li = lane & 15, q = li >> 2, p = li & 3;
    const int kb = ks * 32 + (lane >> 4) * 8;
    typedef bf16x4_t __attribute__((address_space(3))) * lds_ptr;
    const char* a0 = S + (kb + q) * T_ROWB + (ob + 4 * p) * 2;
    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a0));
    const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a0 + 4 * T_ROWB));
    bf16x8_t f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3]; f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
  }
}

// planar (split-precision) outputs: element offset from the hi plane C to the lo plane C_lo; 0 for plain outputs
template <typename TC> __device__ __forceinline__ long c_lo_off(const GemmB16Args&) { return 0; }
template <> __device__ __forceinline__ long c_lo_off<bf16p>(const GemmB16Args& g) {
  return reinterpret_cast<const bf16p*>(g.C_lo) - reinterpret_cast<const bf16p*>(g.C);
}
template <> __device__ __forceinline__ long c_lo_off<f16f8>(const GemmB16Args& g) {      // f16f8 output planes: the same arithmetic (2 bytes per element in both)
  return reinterpret_cast<const f16f8*>(g.C_lo) - reinterpret_cast<const f16f8*>(g.C);
}
// storage of the second output gelu' (kept for the backward): plain bf16 next to a planar C (the backward runs on the hi planes)
template <typename TC> struct ZType { typedef TC type; };
template <> struct ZType<bf16p> { typedef bf16 type; };
template <> struct ZType<f16f8> { typedef bf16 type; };

template <typename TC> __device__ __forceinline__ void store_c(TC* p, float v);
template <> __device__ __forceinline__ void store_c<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void store_c<bf16>(bf16* p, float v) { *p = __float2bfloat16(v); }

template <typename TA, int TRA, typename TB, int TRB, typename TC, int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmB16Args g) {
  __shared__ __attribute__((aligned(16))) char smem[2 * OP_BYTES];
  char* As = smem;
  char* Bs = smem + OP_BYTES;
  const TA* A = reinterpret_cast<const TA*>(g.A);
  const TB* B = reinterpret_cast<const TB*>(g.B);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (each with a private L2), so give every XCD a
  // CONTIGUOUS run of tiles (n fastest): the tiles that share an activation row panel then share one L2.
  int tm, tn;
  {
    const int nwg = gridDim.x * gridDim.y, id = blockIdx.y * gridDim.x + blockIdx.x;
    const int xcd = id & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    tm = wg / gridDim.x;
    tn = wg - tm * gridDim.x;
  }
  const int m0 = tm * GBM, n0 = tn * GBN;
  const int kbeg = blockIdx.z * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  float bsum = 0.f;
  uint4 va[4], vb[4];
  load_op<TA, TRA>(A, g.lda, m0, kbeg, g.M, kend, tid, va);
  load_op<TB, TRB>(B, g.ldb, n0, kbeg, g.N, kend, tid, vb);

  for (int k0 = kbeg; k0 < kend; k0 += GBK) {
    __syncthreads();
    store_op<TRA>(As, tid, va);
    store_op<TRB>(Bs, tid, vb);
    __syncthreads();
    if (k0 + GBK < kend) {
      load_op<TA, TRA>(A, g.lda, m0, k0 + GBK, g.M, kend, tid, va);
      load_op<TB, TRB>(B, g.ldb, n0, k0 + GBK, g.N, kend, tid, vb);
    }
    if (EPI == EPI_SLAB && TRA == 1) {
      if (tn == 0 && tid < GBM) {
        for (int r = 0; r < GBK; ++r)
          bsum += __uint_as_float((unsigned)(*reinterpret_cast<const unsigned short*>(As + r * T_ROWB + tid * 2)) << 16);
      }
    }
#pragma unroll
    for (int ks = 0; ks < GBK / 32; ++ks) {
      bf16x8_t af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = read_frag<TRA>(As, wr * 64 + i * 16, ks, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = read_frag<TRB>(Bs, wc * 64 + j * 16, ks, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);   // C^T tile
    }
  }

  // ---- epilogue.  The MFMA was issued with the operands swapped, i.e. it produced the TRANSPOSED 16x16 tile: lane holds
  // C[m = tile row (lane & 15)][n = tile col 4*(lane>>4) + r], r = 0..3 -> four consecutive output columns per lane,
  // so bias / residual / pre-activation traffic and the stores are 16-byte (fp32) or 8-byte (bf16) vector accesses. ----
  TC* C = reinterpret_cast<TC*>(g.C);
  if (EPI == EPI_SLAB) {
    C += (long)blockIdx.z * g.M * g.ldc;
    if (TRA == 1 && tn == 0 && tid < GBM && m0 + tid < g.M && g.bias_slab != nullptr)
      g.bias_slab[(long)blockIdx.z * g.M + m0 + tid] = bsum;
  }
  TC* Z = reinterpret_cast<TC*>(g.Z);
  const float rsc = g.rscale != 0.f ? g.rscale : 1.0f;     // residual scale (muP: 1 / sqrt(depth))
  const int l15 = lane & 15, gq = lane >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = m0 + wr * 64 + i * 16 + l15;
    if (row >= g.M) continue;
    const float dscale = (EPI == EPI_BIAS_RESID) ? droppath_scale(g.mask, g.mask_mode, row, g.T, g.J) : 1.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = n0 + wc * 64 + j * 16 + 4 * gq;
      if (col >= g.N) continue;
      const long o = (long)row * g.ldc + col;
      float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      if (EPI != EPI_SLAB && EPI != EPI_DGELU && g.bias != nullptr) {
        const float4 b = ld4(g.bias + col);
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
      }
      if (EPI == EPI_BIAS_GELU) {
        const float4 d = gelu_fwd4_fast(v);     // v <- gelu(v); Z keeps gelu' (all the backward needs from the pre-activation); null: inference
        if (Z != nullptr) st4(Z + o, d);
      } else if (EPI == EPI_BIAS_RESID) {
        const float4 r = ld4(g.R + o);
        v = make_float4(r.x * rsc + dscale * v.x, r.y * rsc + dscale * v.y, r.z * rsc + dscale * v.z, r.w * rsc + dscale * v.w);
      } else if (EPI == EPI_DGELU) {
        const float4 z = ld4(Z + o);
        v = make_float4(v.x * z.x, v.y * z.y, v.z * z.z, v.w * z.w);      // Z holds gelu'(pre-activation)
      }
      st4(C + o, v);
    }
  }
}

// =================================================================================================================
// All-bf16 operands: direct-to-LDS staging (global_load_lds_dwordx4, no VGPR / ds_write round trip), two LDS stages,
// ONE barrier per k-tile; the DMA of tile t+1 overlaps the MFMAs of tile t.
// The DMA writes LDS linearly (wave-uniform base + lane*16), so the bank-conflict swizzles are applied to each lane's
// SOURCE address instead and undone by the fragment reads:
//   "N" operand: 128-B rows [row][8 chunks]; LDS slot s of row r holds k-chunk  s ^ (r & 7)
//   "T" operand: 256-B rows [red][16 chunks]; LDS slot s of row r holds out-chunk s ^ 2*((r&3) + 4*((r>>3)&1))
// Both keep full 128/256-byte HBM lines per row (the XOR only permutes chunks inside a row) and make every ds_read_b128 /
// ds_read_b64_tr_b16 of the fragment loads conflict-free.  Out-of-range chunks are sourced from a 16-byte zero page.
// =================================================================================================================
__device__ __attribute__((aligned(16))) unsigned int g_zero_page[4] = {0u, 0u, 0u, 0u};

__device__ __forceinline__ int t_swz(int r) { return 2 * ((r & 3) + 4 * ((r >> 3) & 1)); }

// ROWS = 128 or 256 output indices of this operand in the tile; every wave issues 4 DMA instructions (1 KiB each)
template <int TR, int ROWS, int NWAVES>
__device__ __forceinline__ void glds_tile(char* __restrict__ S, const bf16* __restrict__ P, long ld, int out0, int r0, int OUT, int r_end,
                                          int lane, int wave) {
  typedef __attribute__((address_space(3))) void* lptr;
  typedef const __attribute__((address_space(1))) void* gptr;
  constexpr int NINST = ROWS * 8 / 64 / NWAVES;      // == 4 for (128 rows, 4 waves) and (256 rows, 8 waves)
  constexpr int CPR = ROWS / 8;                      // chunks per reduction row of a "T" image
#pragma unroll
  for (int n4 = 0; n4 < NINST; ++n4) {
    const int n = wave * NINST + n4;                 // wave-uniform instruction index
    const int c = 64 * n + lane;
    const bf16* src;
    bool ok;
    if (TR == 0) {
      const int row = c >> 3, kg = (c & 7) ^ (row & 7);
      const int o = out0 + row, r = r0 + kg * 8;
      ok = o < OUT && r < r_end;
      src = P + (long)o * ld + r;
    } else {
      const int rr = c / CPR, oc = (c % CPR) ^ t_swz(rr);
      const int r = r0 + rr, o = out0 + oc * 8;
      ok = r < r_end && o < OUT;
      src = P + (long)r * ld + o;
    }
    if (!ok) src = reinterpret_cast<const bf16*>(g_zero_page);
    __builtin_amdgcn_global_load_lds((gptr)src, (lptr)(S + n * 1024), 16, 0, 0);
  }
}

template <int TR, int ROWS>
__device__ __forceinline__ bf16x8_t read_frag2(const char* __restrict__ S, int ob, int ks, int lane) {
  if (TR == 0) {
    const int row = ob + (lane & 15), kg = ks * 4 + (lane >> 4);
    return *reinterpret_cast<const bf16x8_t*>(S + row * 128 + ((kg ^ (row & 7)) << 4));
  } else {
    typedef bf16x4_t __attribute__((address_space(3))) * lds_ptr;
    constexpr int ROWB = ROWS * 2;
    const int li = lane & 15, q = li >> 2, p = li & 3;
    const int r = ks * 32 + (lane >> 4) * 8 + q, col = ob + 4 * p;
    const int oc = col >> 3, hf = (col >> 2) & 1, sw = t_swz(r);   // t_swz(r + 4) == t_swz(r) for q < 4
    const char* a0 = S + r * ROWB + ((oc ^ sw) << 4) + hf * 8;
    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a0));
    const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a0 + 4 * ROWB));
    bf16x8_t f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3]; f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
  }
}

// One k-tile (GBK = 64) of MFMAs for a wave's (BT/2) x 64 sub-tile, fragment reads software-pipelined: the ds_reads of
// step s+1 are issued before the 8 MFMAs of step s.
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef int i32x8_t __attribute__((ext_vector_type(8)));
// one 16 x 16 x 32 product on bf16 (F16 = 0) or fp16 (F16 = 1) fragments
template <int F16>
__device__ __forceinline__ f32x4 mfma16(const bf16x8_t& a, const bf16x8_t& b, const f32x4& c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

struct NoMid { __device__ __forceinline__ void operator()() const {} };
// DMA_LATE_STEP: the MFMA group (of 8) behind which the waves of group 1 (waves 4-7: the second wave of every SIMD) issue their share of the
// next k-tile's operand DMA in the plain persistent loop, instead of at the start of the step like group 0 (-1: everybody at the start).
#ifndef DMA_LATE_STEP
#define DMA_LATE_STEP -1
#endif
template <bool F16>
__device__ __forceinline__ float lds16_to_float(unsigned short b) {
  if constexpr (F16) return (float)__builtin_bit_cast(_Float16, b);
  else return __uint_as_float((unsigned)b << 16);
}
// eight fp16 values of a B fragment -> bf16 (round to nearest even), in registers: the weight-gradient GEMM of a layer whose input exists as an
// fp16 plane only (mp_model_config::f16f8 = 3; BCVT below).  12 VALU instructions per fragment, 8 fragments per 64 MFMAs.
__device__ __forceinline__ bf16x8_t frag_f16_to_bf16(const bf16x8_t& f) {
  const uint4 u = __builtin_bit_cast(uint4, f);
  const uint2 lo = f16x4_to_bf16x4(make_uint2(u.x, u.y)), hi = f16x4_to_bf16x4(make_uint2(u.z, u.w));
  return __builtin_bit_cast(bf16x8_t, make_uint4(lo.x, lo.y, hi.x, hi.y));
}
template <int TRA, int TRB, int BT, int ABL = 0, int F16 = 0, typename Mid = NoMid, int BCVT = 0>
__device__ __forceinline__ void mma_stage(const char* __restrict__ As, const char* __restrict__ Bs, f32x4 (&acc)[BT / 32][4], int wr, int wc,
                                          int lane, Mid mid = Mid()) {
  constexpr int MI = BT / 32;
  constexpr int HS = MI / 2, NSTEP = (GBK / 32) * HS;
  bf16x8_t b_cur[4], b_nxt[4], a_cur[2], a_nxt[2];
  // BCVT: b_cur keeps the RAW fp16 fragments at the start of a k-step; each is converted right in front of its first MFMA (below), so that the
  // k-step's first MFMA waits for 12 conversion instructions, not 48 (the conversions of all four fragments as one block in front of the k-tile's
  // first MFMA - where every wave of the workgroup stands at the same time, behind the barrier - cost the weight-gradient kernels 7 %)
#pragma unroll
  for (int j = 0; j < 4; ++j) b_cur[j] = read_frag2<TRB, BT>(Bs, wc * 64 + j * 16, 0, lane);
  a_cur[0] = read_frag2<TRA, BT>(As, wr * (BT / 2), 0, lane);
  a_cur[1] = read_frag2<TRA, BT>(As, wr * (BT / 2) + 16, 0, lane);
  // ABL (timing ablation, results wrong): 1 = the B fragments are read once per call instead of once per 32 reduction indices,
  // 2 = the A fragments are read once per call
#pragma unroll
  for (int step = 0; step < NSTEP; ++step) {
    const int ip = step % HS;
    // Pinned order per group of 8 MFMAs: first MFMA (hipcc puts its s_waitcnt lgkmcnt(0) for this group's fragments in front of it), then the
    // LDS reads of the next group, then the other 7 MFMAs, which cover the reads' latency.  Left to itself (or with sched_group_barrier
    // hints) hipcc issues most reads directly in front of an s_waitcnt lgkmcnt(0) and their first use - a dozen exposed LDS round trips per
    // k-tile and wave; and with the reads ahead of the group it still waits with lgkmcnt(0), i.e. for the reads it has just issued.
    __builtin_amdgcn_sched_barrier(0);
    if (BCVT && ip == 0) { b_cur[0] = frag_f16_to_bf16(b_cur[0]); __builtin_amdgcn_sched_barrier(0); }
    acc[2 * ip][0] = mfma16<F16>(b_cur[0], a_cur[0], acc[2 * ip][0]);   // C^T tile
    __builtin_amdgcn_sched_barrier(0);
    if (step + 1 < NSTEP) {
      const int nks = (step + 1) / HS, nip = (step + 1) % HS;
      if (ABL & 2) { a_nxt[0] = a_cur[1]; a_nxt[1] = a_cur[0]; }
      else {
        a_nxt[0] = read_frag2<TRA, BT>(As, wr * (BT / 2) + (2 * nip) * 16, nks, lane);
        a_nxt[1] = read_frag2<TRA, BT>(As, wr * (BT / 2) + (2 * nip + 1) * 16, nks, lane);
      }
      if (nip == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) b_nxt[j] = (ABL & 1) ? b_cur[(j + 1) & 3] : read_frag2<TRB, BT>(Bs, wc * 64 + j * 16, nks, lane);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 1; j < 4; ++j) {
      if (BCVT && ip == 0) { b_cur[j] = frag_f16_to_bf16(b_cur[j]); __builtin_amdgcn_sched_barrier(0); }      // (pinned: hipcc would gather the conversions in front of the group)
      acc[2 * ip][j] = mfma16<F16>(b_cur[j], a_cur[0], acc[2 * ip][j]);
      if (BCVT && ip == 0) __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[2 * ip + 1][j] = mfma16<F16>(b_cur[j], a_cur[1], acc[2 * ip + 1][j]);
    __builtin_amdgcn_sched_barrier(0);
    if (step == DMA_LATE_STEP) { mid(); __builtin_amdgcn_sched_barrier(0); }
    if (step + 1 < NSTEP) {
      a_cur[0] = a_nxt[0];
      a_cur[1] = a_nxt[1];
      if ((step + 1) % HS == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) b_cur[j] = b_nxt[j];      // (BCVT: raw; converted in front of their first MFMAs)
      }
    }
  }
}

// ---- "f16f8" split (SPLIT = 8): the two correction products of a split-precision Linear on block-scaled fp8 operands ----
// A value x is carried as hi = fp16(x) plus an 8-bit CORRECTION plane of the same byte geometry (2 bytes per element; common.h): per four
// reduction indices 8 bytes = 4 x e4m3(2^11 (x - hi)) | 4 x e4m3(hi) for an activation, 4 x e4m3(2^4 hi) | 4 x e4m3(2^15 (w - hi)) for a
// weight, so that ONE 128-deep fp8 product of the two rows' 128 bytes per 64 indices is 2^15 (x_lo w_hi + x_hi w_lo); v_mfma_scale_f32_16x16x128_f8f6f4 multiplies by the
// 2^-15 (its E8M0 block scales, all equal here) while accumulating into the fp32 registers that hold the fp16 product x_hi w_hi.  The
// 16 x 16 x 128 fp8 instruction takes twice the cycles of the 16 x 16 x 32 fp16 one for four times the depth: a k-tile costs two
// matrix-core steps instead of the three of the bf16 split, at fp16's 11 significand bits for the main product (oracle/precision_model.py:
// 2.3e-5 m against 6.0e-6 m for bf16x3 and 4.4e-4 m for a lone fp16 product, on the full-size model).
// Fragment of 16 rows: the lane's 32 bytes are the 16-byte chunks g and g + 4 of its row - exactly the two bf16 fragments (k-steps 0 and 1)
// of the same LDS image, so the reads are the conflict-free ones of the bf16 kernel; WHICH reduction index the hardware assigns to a byte
// does not matter as long as both operands use the same map (a dot product is invariant under a common permutation of its terms).
__device__ __forceinline__ i32x8_t read_frag8(const char* __restrict__ S, int ob, int lane) {
  struct Two { bf16x8_t a, b; } t;
  t.a = read_frag2<0, 256>(S, ob, 0, lane);
  t.b = read_frag2<0, 256>(S, ob, 1, lane);
  return __builtin_bit_cast(i32x8_t, t);
}
constexpr int F8_SCALE_A = 0x70707070;      // E8M0 2^(112 - 127) = 2^-15 in every byte
constexpr int F8_SCALE_B = 0x7F7F7F7F;      // 1
__device__ __forceinline__ f32x4 mfma_f8(const i32x8_t& a, const i32x8_t& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, F8_SCALE_A, 0, F8_SCALE_B);
}
// One 128-byte k-tile of a wave's 128 x 64 sub-tile in either mode, ONE code path: the fragments of both modes are the same 32 bytes per lane
// and row (16-byte chunks g and g + 4), so the LDS reads, their addresses and their registers are shared and only the matrix instruction
// differs - f8 = false: the chunks are two fp16 fragments (k-steps 0 and 1), two 16 x 16 x 32 MFMAs per block; f8 = true: one 16 x 16 x 128
// fp8 MFMA per block.  (Two specialised copies of the stage in the k loop keep two sets of address registers alive and spill 116 VGPRs.)
// Groups of one 16-row A fragment against the four B fragments; the next A fragment is requested behind the group's first MFMA.
typedef int i32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f16x8_t half_lo(const i32x8_t& v) { return __builtin_bit_cast(f16x8_t, __builtin_shufflevector(v, v, 0, 1, 2, 3)); }
__device__ __forceinline__ f16x8_t half_hi(const i32x8_t& v) { return __builtin_bit_cast(f16x8_t, __builtin_shufflevector(v, v, 4, 5, 6, 7)); }
__device__ __forceinline__ void mma_stage_mix(const char* __restrict__ As, const char* __restrict__ Bs, f32x4 (&acc)[8][4], int wr, int wc, int lane,
                                              bool f8) {
  i32x8_t b[4], a_cur, a_nxt;
#pragma unroll
  for (int j = 0; j < 4; ++j) b[j] = read_frag8(Bs, wc * 64 + j * 16, lane);
  a_cur = read_frag8(As, wr * 128, lane);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    __builtin_amdgcn_sched_barrier(0);
    if (f8) acc[i][0] = mfma_f8(b[0], a_cur, acc[i][0]);
    else acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(half_lo(b[0]), half_lo(a_cur), acc[i][0], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (i + 1 < 8) a_nxt = read_frag8(As, wr * 128 + (i + 1) * 16, lane);
    __builtin_amdgcn_sched_barrier(0);
    if (f8) {
#pragma unroll
      for (int j = 1; j < 4; ++j) acc[i][j] = mfma_f8(b[j], a_cur, acc[i][j]);
    } else {
#pragma unroll
      for (int j = 1; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(half_lo(b[j]), half_lo(a_cur), acc[i][j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(half_hi(b[j]), half_hi(a_cur), acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (i + 1 < 8) a_cur = a_nxt;
  }
}

// Per-lane byte offsets of the 4 DMA instructions a wave issues per operand and k-tile, relative to the (wave-uniform)
// address of the tile's first element of that k-tile; the same source swizzles as glds_tile.  Rows past the end of the
// matrix are clamped to its last row (their products are never stored), so no lane needs a different base.
template <int TR>
__device__ __forceinline__ void persist_offsets(unsigned (&off)[4], long ld, int out0, int OUT, int lane, int wave) {
#pragma unroll
  for (int n4 = 0; n4 < 4; ++n4) {
    const int c = 64 * (wave * 4 + n4) + lane;
    if (TR == 0) {
      const int row = c >> 3, kg = (c & 7) ^ (row & 7);
      off[n4] = (unsigned)(min(row, OUT - 1 - out0) * (int)ld * 2 + kg * 16);
    } else {
      const int rr = c >> 5, oc = (c & 31) ^ t_swz(rr);      // OUT % 256 == 0 (launcher): every chunk is in range
      off[n4] = (unsigned)(rr * (int)ld * 2 + oc * 16);
    }
  }
}
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
// the same for NP consecutive pieces starting at piece `first` (MP_KSTEP_YOUNG: waves 4-7 take 8 pieces of a tile each)
template <int TR, int NP>
__device__ __forceinline__ void persist_offsets_n(unsigned (&off)[NP], long ld, int out0, int OUT, int lane, int first) {
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int c = 64 * (first + i) + lane;
    if (TR == 0) {
      const int row = c >> 3, kg = (c & 7) ^ (row & 7);
      off[i] = (unsigned)(min(row, OUT - 1 - out0) * (int)ld * 2 + kg * 16);
    } else {
      const int rr = c >> 5, oc = (c & 31) ^ t_swz(rr);
      off[i] = (unsigned)(rr * (int)ld * 2 + oc * 16);
    }
  }
}
__device__ __forceinline__ void persist_dma(char* __restrict__ S, const char* __restrict__ base, const unsigned (&off)[4], int wave) {
  typedef __attribute__((address_space(3))) void* lptr;
  typedef const __attribute__((address_space(1))) void* gptr;
#pragma unroll
  for (int n4 = 0; n4 < 4; ++n4) __builtin_amdgcn_global_load_lds((gptr)(base + off[n4]), (lptr)(S + (wave * 4 + n4) * 1024), 16, 0, 0);
}

// ---- hand-scheduled k-step (MP_KLOOP_ASM; tools/gen_kloop_asm.py -> kloop_asm.inc) -------------------------------------------------------
// One 64-wide step of a wave's 128 x 64 sub-tile as a single inline-asm block: the 64 MFMAs, the 24 fragment requests (two groups of 8 MFMAs
// ahead, counted lgkmcnt waits, fragments in the fixed registers v[200:255]) and the wave's share of the step's operand DMA.  What hipcc makes
// of mma_stage above: requests one group ahead behind lgkmcnt(0) waits, the DMA instructions (and ~60 scalar instructions of address and
// predicate arithmetic) in front of the step's first fragment request.
#ifndef MP_KLOOP_ASM
#define MP_KLOOP_ASM 1      // round 5: block of four split-precision forward GEMMs 3.81 -> 3.61 ms, block of four dgrads 1.235 -> 1.215 ms
#endif
#include "kloop_asm.inc"
// schedule variant of the generated step (tools/gen_kloop_asm.py, VARIANTS), for the split-precision loop and the plain loop.  Measured (same box,
// alternating, block of four GEMMs): split 0: 3626-3635, 1: 3629-3656, 2: 3608-3615, 3: 3677-3688 us (hipcc's own schedule 3803-3812);
// dgrad 0: 1213-1218, 1: 1206-1219, 2: 1222, 3: 1244-1246 us (hipcc 1231-1240) - profiles/r05_probes/kstep_variants.log
#ifndef MP_KSTEP_VARIANT
#define MP_KSTEP_VARIANT 2
#endif
#ifndef MP_KSTEP_VARIANT_P
#define MP_KSTEP_VARIANT_P 0
#endif
// (Round 5: a five-buffer form of the split-precision loop - A_lo | A_hi | B_lo | B_hi even | B_hi odd k-tile, the epilogue images inside A_hi, TWO
// barriers per k-tile with (A_hi, B_hi) and (A_hi, B_lo) as one block of 128 MFMAs - was built, passed the parity tests and measured 1.8 % SLOWER
// (block of four 3941-3951 us against 3875 us, profiles/r05_probes/x3_five_buffer_ab.log): A_hi and B_lo can only be requested behind the k-tile's
// first barrier and are then both needed 64 MFMAs later.  With the operand DMA moved to the younger waves (MP_KSTEP_YOUNG) the two forms are equal
// (3793-3805 against 3781-3787 us, x3_five_buffer_young_ab.log).  Removed; the code is in the history, commit "five-buffer / two-barrier form".)
#define MP_KSTEP_CLOB2(v) MP_KSTEP_CLOBBERS_V##v
#define MP_KSTEP_CLOB(v) MP_KSTEP_CLOB2(v)
__device__ __forceinline__ unsigned lds_u32(const void* p) {
  return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char*)p;
}
// per-lane parts of the fragment addresses (loop-invariant): A / "N" B image: row r = block rows + (lane & 15), 16-byte chunk
// (4 ks + (lane >> 4)) ^ (r & 7) of the 128-byte row (read_frag2<0>); "T" B image: read_frag2<1, 256> for the four 16-column blocks.
// Held per LDS buffer (base + per-lane part), so that a step needs no address arithmetic behind its barrier.
struct KFragA { unsigned a[2]; };
struct KFragB { unsigned b[4]; };
__device__ __forceinline__ void kfrag_a(KFragA& f, unsigned base, int wr, int lane) {
  const int l15 = lane & 15, gq = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) f.a[ks] = base + (unsigned)((wr * 128 + l15) * 128 + (((ks * 4 + gq) ^ (lane & 7)) << 4));
}
template <int TRB>
__device__ __forceinline__ void kfrag_b(KFragB& f, unsigned base, int wc, int lane) {
  const int l15 = lane & 15, gq = lane >> 4;
  if (TRB == 0) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) f.b[ks] = base + (unsigned)((wc * 64 + l15) * 128 + (((ks * 4 + gq) ^ (lane & 7)) << 4));
    f.b[2] = f.b[3] = 0;
  } else {
    const int q = l15 >> 2, p = l15 & 3, r = gq * 8 + q;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = wc * 64 + j * 16 + 4 * p;
      f.b[j] = base + (unsigned)(r * 512 + (((col >> 3) ^ t_swz(r)) << 4) + ((col >> 2) & 1) * 8);
    }
  }
}
// MP_KSTEP_YOUNG (with MP_KLOOP_ASM): the operand DMA of the hand-scheduled loops is issued by waves 4-7 only - the younger wave of every SIMD, 8
// pieces per tile each - instead of 4 pieces by every wave (configurations Z0 / Z1 / Z2 / ZP of the generator)
#ifndef MP_KSTEP_YOUNG
#define MP_KSTEP_YOUNG 1
#endif
// Job configuration of a step = which DMA jobs its asm block can carry (tools/gen_kloop_asm.py, CONFIGS): X0 / X1 / X2 = the three steps of a
// k-tile of the split-precision loop, P = a step of the plain loop.  ONE block per step type, the jobs that are not always there behind a
// wave-uniform flag in a scalar register: two blocks on the two sides of a C++ branch make hipcc reconcile the 128 accumulator registers
// through 500-800 bytes of scratch memory per lane.
// The flags are made by the SCALAR ALU from scalar sources (kflag_*): an asm "s" operand takes nothing else, hipcc keeps wave-uniform bools as
// lane masks + v_cndmask and folds __builtin_amdgcn_readfirstlane of a value it knows to be uniform, and a hand-written v_readfirstlane of such
// a value was observed stale (round 5: one tile in a few fetched the tile behind the workgroup's last one; the mechanism was worked around, not
// identified - a plausible one is the gfx9 hazard of a VALU-written SGPR read as m0 / by VMEM within a few wait states, which the hazard
// recogniser does not see inside inline asm).  For a source hipcc holds in a vector register (e.g. `wave` below) it inserts its OWN
// v_readfirstlane in front of the block, outside the asm, where its hazard recogniser does apply; nothing here relies on a build error.
struct KJob { int en; unsigned lds; const char* base; };      // wave-uniform: enable flag (conditional jobs), LDS address of the wave's 4 KiB of the destination tile, source address
__device__ __forceinline__ int kflag_lt(int a, int b) { int r; asm volatile("s_cmp_lt_i32 %1, %2\n\ts_cselect_b32 %0, 1, 0" : "=s"(r) : "s"(a), "s"(b) : "scc"); return r; }
__device__ __forceinline__ int kflag_nonnull(const void* p) { int r; asm volatile("s_cmp_lg_u64 %1, 0\n\ts_cselect_b32 %0, 1, 0" : "=s"(r) : "s"(p) : "scc"); return r; }
// (a != b) ? 1 : x   -   "more work follows this k-tile": not the tile's last k-tile, or a next tile exists
__device__ __forceinline__ int kflag_more(int a, int b, int x) { int r; asm volatile("s_cmp_lg_u32 %1, %2\n\ts_cselect_b32 %0, 1, %3" : "=s"(r) : "s"(a), "s"(b), "s"(x) : "scc"); return r; }
// (a == b) ? x : 0
__device__ __forceinline__ int kflag_last(int a, int b, int x) { int r; asm volatile("s_cmp_eq_u32 %1, %2\n\ts_cselect_b32 %0, %3, 0" : "=s"(r) : "s"(a), "s"(b), "s"(x) : "scc"); return r; }
// MP_KLOOP_PIPE (with MP_KLOOP_ASM and MP_KSTEP_YOUNG): k-tiles 0 .. nk-2 of a tile of the split-precision forward loop run as ONE asm block with the
// loop inside and the fragment pipeline continuing across the steps - every step's barrier stands two MFMA groups before its end instead of in front
// of it (tools/gen_kloop_asm.py, pipe_loop_x3); the tile's last k-tile keeps the block-per-step form.
#ifndef MP_KLOOP_PIPE
#define MP_KLOOP_PIPE 1
#endif
// MP_KLOOP_ASM8 (with MP_KLOOP_ASM and MP_KSTEP_YOUNG): the "f16f8" loop (SPLIT = 8) in the hand-scheduled form, one block per step (0: hipcc's mma_stage_mix)
#ifndef MP_KLOOP_ASM8
#define MP_KLOOP_ASM8 1
#endif
// schedule variant of the f16f8 steps (tools/gen_kloop_asm.py, F8_VARIANTS: 0 = DMA in front of the first fragment requests, requests one group ahead;
// 1 = the same with requests two groups ahead; 2 = DMA behind the first requests, two groups ahead - the three-product loop's best)
#ifndef MP_KSTEP_VARIANT_F8
#define MP_KSTEP_VARIANT_F8 0
#endif
enum { KC_X0 = 0, KC_X1 = 1, KC_X2 = 2, KC_P = 3 };
#define MP_KSTEP_SEL2(c, t, v) MP_KSTEP_ASM_##c##_TRB##t##_V##v
#define MP_KSTEP_SEL(c, t, v) MP_KSTEP_SEL2(c, t, v)
#if MP_KSTEP_YOUNG
constexpr int KNP = 8;      // DMA pieces (1 KiB each) per operand tile and issuing wave
#define MP_KSTEP_OFFS(o, arr) [o##0] "v"(arr[0]), [o##1] "v"(arr[1]), [o##2] "v"(arr[2]), [o##3] "v"(arr[3]), [o##4] "v"(arr[4]), [o##5] "v"(arr[5]), [o##6] "v"(arr[6]), [o##7] "v"(arr[7])
#else
constexpr int KNP = 4;
#define MP_KSTEP_OFFS(o, arr) [o##0] "v"(arr[0]), [o##1] "v"(arr[1]), [o##2] "v"(arr[2]), [o##3] "v"(arr[3])
#endif
#define MP_KSTEP_JOBS                                                                                    \
  [ena] "s"(ja.en), [ldsa] "s"(ja.lds), [gba] "s"(ja.base), MP_KSTEP_OFFS(ao, aoff),                       \
  [enb] "s"(jb.en), [ldsb] "s"(jb.lds), [gbb] "s"(jb.base), MP_KSTEP_OFFS(bo, boff),                       \
  [enc] "s"(jc.en), [ldsc] "s"(jc.lds), [gbc] "s"(jc.base), [co] "v"(coff)
#define MP_KSTEP_OPS_0 [aa0] "v"(fa.a[0]), [aa1] "v"(fa.a[1]), [ba0] "v"(fb.b[0]), [ba1] "v"(fb.b[1]), MP_KSTEP_JOBS
#define MP_KSTEP_OPS_1 [aa0] "v"(fa.a[0]), [aa1] "v"(fa.a[1]), [bt0] "v"(fb.b[0]), [bt1] "v"(fb.b[1]), [bt2] "v"(fb.b[2]), [bt3] "v"(fb.b[3]), MP_KSTEP_JOBS
#define MP_KSTEP_EMIT(c, t, v) asm volatile(MP_KSTEP_SEL(c, t, v) : MP_KSTEP_ACC_OPERANDS : MP_KSTEP_OPS_##t : MP_KSTEP_CLOB(v))
// ("m0" is on the blocks' clobber lists - their DMA set-up writes it, and a compiler-made global_load_lds / readlane behind a block must
// re-initialise it; clang accepts the reserved register and warns about it)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
template <int TRB, int CFG>
__device__ __forceinline__ void kstep_asm(f32x4 (&acc)[8][4], const KFragA& fa, const KFragB& fb, const KJob& ja, const unsigned (&aoff)[KNP],
                                          const KJob& jb, const unsigned (&boff)[KNP], const KJob& jc, unsigned coff) {
  if constexpr (TRB == 0) {
#if MP_KSTEP_YOUNG
    if constexpr (CFG == KC_X0) MP_KSTEP_EMIT(Z0, 0, MP_KSTEP_VARIANT);
    else if constexpr (CFG == KC_X1) MP_KSTEP_EMIT(Z1, 0, MP_KSTEP_VARIANT);
    else if constexpr (CFG == KC_X2) MP_KSTEP_EMIT(Z2, 0, MP_KSTEP_VARIANT);
    else MP_KSTEP_EMIT(ZP, 0, MP_KSTEP_VARIANT_P);
#else
    if constexpr (CFG == KC_X0) MP_KSTEP_EMIT(X0, 0, MP_KSTEP_VARIANT);
    else if constexpr (CFG == KC_X1) MP_KSTEP_EMIT(X1, 0, MP_KSTEP_VARIANT);
    else if constexpr (CFG == KC_X2) MP_KSTEP_EMIT(X2, 0, MP_KSTEP_VARIANT);
    else MP_KSTEP_EMIT(P, 0, MP_KSTEP_VARIANT_P);
#endif
  } else {
#if MP_KSTEP_YOUNG
    if constexpr (CFG == KC_X0) MP_KSTEP_EMIT(Z0, 1, MP_KSTEP_VARIANT);
    else if constexpr (CFG == KC_X1) MP_KSTEP_EMIT(Z1, 1, MP_KSTEP_VARIANT);
    else if constexpr (CFG == KC_X2) MP_KSTEP_EMIT(Z2, 1, MP_KSTEP_VARIANT);
    else MP_KSTEP_EMIT(ZP, 1, MP_KSTEP_VARIANT_P);
#else
    if constexpr (CFG == KC_X0) MP_KSTEP_EMIT(X0, 1, MP_KSTEP_VARIANT);
    else if constexpr (CFG == KC_X1) MP_KSTEP_EMIT(X1, 1, MP_KSTEP_VARIANT);
    else if constexpr (CFG == KC_X2) MP_KSTEP_EMIT(X2, 1, MP_KSTEP_VARIANT);
    else MP_KSTEP_EMIT(P, 1, MP_KSTEP_VARIANT_P);
#endif
  }
}
#if MP_KSTEP_YOUNG
// The two steps of a k-tile of the "f16f8" loop (SPLIT = 8) in the hand-scheduled form (round 6; generator: step(.., mfma = fp16) and step_f8):
// F multiplies the fp16 tiles (64 v_mfma_f32_16x16x32_f16), E the 8-bit correction tiles (32 v_mfma_scale_f32_16x16x128_f8f6f4, block scales
// 2^-15 x 1 in two vector registers); both carry the A / B jobs of the NEXT step's tiles behind scalar flags, E the bias behind a tile's last step.
__device__ __forceinline__ void kstep_asm_f(f32x4 (&acc)[8][4], const KFragA& fa, const KFragB& fb, const KJob& ja, const unsigned (&aoff)[KNP],
                                            const KJob& jb, const unsigned (&boff)[KNP]) {
  const KJob jc = {0, 0u, nullptr};
  const unsigned coff = 0;
  asm volatile(MP_KSTEP_SEL(ZF, 0, MP_KSTEP_VARIANT_F8) : MP_KSTEP_ACC_OPERANDS : MP_KSTEP_OPS_0 : MP_KSTEP_CLOB(2));
}
#define MP_KSTEP_OPS_E [aa0] "v"(fa.a[0]), [aa1] "v"(fa.a[1]), [ba0] "v"(fb.b[0]), [ba1] "v"(fb.b[1]), [sca] "v"(sca), [scb] "v"(scb), MP_KSTEP_JOBS
__device__ __forceinline__ void kstep_asm_e(f32x4 (&acc)[8][4], const KFragA& fa, const KFragB& fb, const KJob& ja, const unsigned (&aoff)[KNP],
                                            const KJob& jb, const unsigned (&boff)[KNP], const KJob& jc, unsigned coff, int sca, int scb) {
  asm volatile(MP_KSTEP_SEL(ZE, 0, MP_KSTEP_VARIANT_F8) : MP_KSTEP_ACC_OPERANDS : MP_KSTEP_OPS_E : MP_KSTEP_CLOB(2));
}
// k-tiles 0 .. n-1 (n = nk - 1 >= 1) of a split-precision forward tile, behind the tile's first barrier (A_lo[0], B_hi[0] landed in A0, B0; A1, B1 free).
// pa / pb: the hi planes' addresses of the tile's k-tile 0.  aoff / boff come back advanced by n k-tiles (128 bytes each) in the waves that issue DMA.
__device__ __forceinline__ void kpipe_x3(f32x4 (&acc)[8][4], const KFragA (&kfa)[2], const KFragB (&kfb)[2], unsigned (&aoff)[KNP], unsigned (&boff)[KNP],
                                         int dma_wave, unsigned dma_l, const char* pa, const char* pb, long a_lo, long b_lo, int n) {
  const char* const pahi0 = pa;                    // A_hi[0], B_lo[0]: the entry's requests
  const char* const pblo0 = pb + b_lo;
  const char* const pahi1 = pa + 128;              // "the next k-tile" of each plane: the offsets advance, these stay
  const char* const palo1 = pa + 128 + a_lo;
  const char* const pbhi1 = pb + 128;
  const char* const pblo1 = pb + 128 + b_lo;
  asm volatile(MP_KPIPE_X3_ASM
               : MP_KSTEP_ACC_OPERANDS, [cnt] "+s"(n),
                 [ao0] "+v"(aoff[0]), [ao1] "+v"(aoff[1]), [ao2] "+v"(aoff[2]), [ao3] "+v"(aoff[3]), [ao4] "+v"(aoff[4]), [ao5] "+v"(aoff[5]), [ao6] "+v"(aoff[6]), [ao7] "+v"(aoff[7]),
                 [bo0] "+v"(boff[0]), [bo1] "+v"(boff[1]), [bo2] "+v"(boff[2]), [bo3] "+v"(boff[3]), [bo4] "+v"(boff[4]), [bo5] "+v"(boff[5]), [bo6] "+v"(boff[6]), [bo7] "+v"(boff[7])
               : [a0k0] "v"(kfa[0].a[0]), [a0k1] "v"(kfa[0].a[1]), [a1k0] "v"(kfa[1].a[0]), [a1k1] "v"(kfa[1].a[1]),
                 [b0k0] "v"(kfb[0].b[0]), [b0k1] "v"(kfb[0].b[1]), [b1k0] "v"(kfb[1].b[0]), [b1k1] "v"(kfb[1].b[1]),
                 [dmaw] "s"(dma_wave), [lds] "s"(dma_l), [pahi0] "s"(pahi0), [pblo0] "s"(pblo0), [pahi1] "s"(pahi1), [palo1] "s"(palo1), [pbhi1] "s"(pbhi1), [pblo1] "s"(pblo1)
               : MP_KSTEP_CLOB(MP_KSTEP_VARIANT));
}
#endif
#pragma clang diagnostic pop

// The same with an UNEVEN split of a tile's 32 DMA pieces over the waves (weight-gradient kernel): the waves of group 0 (0-3) take P0 pieces
// each, those of group 1 (4-7) 8 - P0.  Measured, not derived: group 0 - the older wave of every SIMD - with ONE piece per operand and group 1
// with seven is 3 % faster than the even split on the weight-gradient shapes; the same split does nothing for the persistent kernels.
template <int TR, int NP>
__device__ __forceinline__ void uneven_offsets(unsigned (&off)[NP], long ld, int lane, int first) {
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int c = 64 * (first + i) + lane;
    if (TR == 0) {
      const int row = c >> 3, kg = (c & 7) ^ (row & 7);
      off[i] = (unsigned)(row * (int)ld * 2 + kg * 16);
    } else {
      const int rr = c >> 5, oc = (c & 31) ^ t_swz(rr);
      off[i] = (unsigned)(rr * (int)ld * 2 + oc * 16);
    }
  }
}
template <int NP>
__device__ __forceinline__ void uneven_dma(char* __restrict__ S, const char* __restrict__ base, const unsigned (&off)[NP], int first, int np) {
  typedef __attribute__((address_space(3))) void* lptr;
  typedef const __attribute__((address_space(1))) void* gptr;
#pragma unroll
  for (int i = 0; i < NP; ++i)
    if (i < np) __builtin_amdgcn_global_load_lds((gptr)(base + off[i]), (lptr)(S + (first + i) * 1024), 16, 0, 0);
}

// BT x BT output tile (BT = 128: 4 waves, 256: 8 waves); waves laid out 2 x (BT/64); each wave (BT/2) x 64.
// SPLIT = 1 (split precision, common.h): A and B are the hi planes of planar operands; every 64-wide k-tile is multiplied three
// times - (A_lo, B_hi), (A_hi, B_hi), (A_hi, B_lo) - into the same fp32 accumulators: three main-loop steps per k-tile over four
// fetched operand tiles (see the loop).
#ifndef WG_P0
#define WG_P0 1      // measured on the four weight-gradient shapes of a block: 4 (even) 1485 us, 2: 1470, 1: 1444, 0: 1680; 6: 1535
#endif
template <int TRA, int TRB, typename TC, int EPI, int BT, int SPLIT = 0>
__global__ __launch_bounds__(BT * 2) void gemm_bf16_glds_kernel(GemmB16Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NW = BT / 32, WN = BT / 64, MI = BT / 32, OPB = BT * 128, STAGE = 2 * OPB;
  const bf16* A = reinterpret_cast<const bf16*>(g.A);
  const bf16* B = reinterpret_cast<const bf16*>(g.B);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WN, wc = wave % WN;
  // XCD-aware bijective remap over the WHOLE grid, split index slowest: every XCD gets a contiguous chunk of (split, tile) pairs,
  // so the tiles of one k-split - which all read the same token range of both operands - share that XCD's L2 (split-K wgrad:
  // the per-launch fabric traffic was 2.4x the algorithmic bytes when the splits were dealt round-robin over the XCDs).
  int tm, tn, tz;
  {
    const int tiles = gridDim.x * gridDim.y, nwg = tiles * gridDim.z;
    const int id = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const int xcd = id & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    tz = wg / tiles;
    const int t = wg - tz * tiles;
    tm = t / gridDim.x;
    tn = t - tm * gridDim.x;
  }
  const int m0 = tm * BT, n0 = tn * BT;
  const int kbeg = tz * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);
  f32x4 acc[MI][4];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  f32x4 accb[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};     // 256-wide tiles: bias-gradient column sums of this wave's 32 output rows
  // the bias gradient of a (row tile, split) is summed by the first bias_parts column tiles, k-tile kt by column tile kt % bias_parts, each
  // into a slab row of its own (reduce_slabs_b16_kernel adds them): every workgroup of the launch then carries the same extra work
  const int bias_parts = (BT == 256 && g.bias_parts > 1) ? g.bias_parts : 1;
  const bool bias_here = tn < bias_parts && g.bias_slab != nullptr;

  // neighbouring tiles issue their two operand loads in opposite order: workgroups that share a panel then ask for it at different
  // moments of the k-tile instead of all at once (measured on the weight-gradient shapes: 3-18 % fewer fabric reads, same isolated time)
  const bool b_first = (tm + tn) & 1;
  if (b_first) glds_tile<TRB, BT, NW>(smem + OPB, B, g.ldb, n0, kbeg, g.N, kend, lane, wave);
  glds_tile<TRA, BT, NW>(smem, (SPLIT & 1) ? reinterpret_cast<const bf16*>(g.A_lo) : A, g.lda, m0, kbeg, g.M, kend, lane, wave);
  if (!b_first) glds_tile<TRB, BT, NW>(smem + OPB, B, g.ldb, n0, kbeg, g.N, kend, lane, wave);
  int stage = 0;
  if constexpr (SPLIT & 1) {
    // Three steps per k-tile kt, each one mma_stage over 64 reduction indices, with FOUR operand tiles fetched (not six): the buffers are
    // A0 | B0 | A1 | B1 (the two stages of the plain kernel) and
    //   step 0 multiplies (A0 = A_lo[kt], B0 = B_hi[kt])   while A1 <- A_hi[kt] is fetched
    //   step 1 multiplies (A1 = A_hi[kt], B0 = B_hi[kt])   while B1 <- B_lo[kt]
    //   step 2 multiplies (A1 = A_hi[kt], B1 = B_lo[kt])   while A0 <- A_lo[kt+1], B0 <- B_hi[kt+1]
    // every buffer is rewritten at least one barrier after its last reader.  The prologue above fetched (A_lo[0], B_hi[0]).
    const bf16* const A_lo = reinterpret_cast<const bf16*>(g.A_lo);
    const bf16* const B_lo = reinterpret_cast<const bf16*>(g.B_lo);
    char* const A0 = smem;
    char* const B0 = smem + OPB;
    char* const A1 = smem + STAGE;
    char* const B1 = smem + STAGE + OPB;
    for (int k0 = kbeg, term = 0; k0 < kend;) {       // one loop body for the three steps (a single copy of the MFMA stage)
      __syncthreads();
      const char* As = term == 0 ? A0 : A1;
      const char* Bs = term == 2 ? B1 : B0;
      if (term == 0) {
        glds_tile<TRA, BT, NW>(A1, A, g.lda, m0, k0, g.M, kend, lane, wave);
      } else if (term == 1) {
        glds_tile<TRB, BT, NW>(B1, B_lo, g.ldb, n0, k0, g.N, kend, lane, wave);
      } else if (k0 + GBK < kend) {
        glds_tile<TRA, BT, NW>(A0, A_lo, g.lda, m0, k0 + GBK, g.M, kend, lane, wave);
        glds_tile<TRB, BT, NW>(B0, B, g.ldb, n0, k0 + GBK, g.N, kend, lane, wave);
      }
      mma_stage<TRA, TRB, BT>(As, Bs, acc, wr, wc, lane);
      if (++term == 3) { term = 0; k0 += GBK; }
    }
    (void)stage;
  } else {
  // Weight-gradient shape (both operands read transposed, 256 x 256 tile): when the tile lies inside both matrices, a k-tile that lies inside
  // the split's range is fetched from a wave-uniform base + per-lane 32-bit offsets computed once (the general glds_tile spends ~100
  // VALU instructions per wave and k-tile, a dozen of them 64-bit multiplies, on addresses and bounds ahead of the MFMA stage)
  constexpr bool FASTDMA = BT == 256 && TRA == 1 && TRB == 1;
  constexpr int P0 = WG_P0, PMAX = P0 > 8 - P0 ? P0 : 8 - P0;
  unsigned aoff[PMAX], boff[PMAX];
  bool fast_tile = false;
  const int np = wave < 4 ? P0 : 8 - P0, first = wave < 4 ? wave * P0 : 4 * P0 + (wave - 4) * (8 - P0);
  if constexpr (FASTDMA) {
    fast_tile = m0 + BT <= g.M && n0 + BT <= g.N && (long)GBK * max(g.lda, g.ldb) * 2 + 512 < (1L << 31);
    uneven_offsets<1, PMAX>(aoff, g.lda, lane, first);
    uneven_offsets<1, PMAX>(boff, g.ldb, lane, first);
  }
  for (int k0 = kbeg; k0 < kend; k0 += GBK, stage ^= 1) {
    __syncthreads();   // (vmcnt(0) + barrier): tile k0 has landed for every wave; nobody still reads the other stage
    const char* As = smem + stage * STAGE;
    const char* Bs = As + OPB;
    if (k0 + GBK < kend && !MP_DBG(g, 2)) {
      char* nx = smem + (stage ^ 1) * STAGE;
      if (FASTDMA && fast_tile && k0 + 2 * GBK <= kend) {
        const char* const ab = reinterpret_cast<const char*>(A) + ((long)(k0 + GBK) * g.lda + m0) * 2;
        const char* const bb = reinterpret_cast<const char*>(B) + ((long)(k0 + GBK) * g.ldb + n0) * 2;
        if (b_first) uneven_dma<PMAX>(nx + OPB, bb, boff, first, np);
        uneven_dma<PMAX>(nx, ab, aoff, first, np);
        if (!b_first) uneven_dma<PMAX>(nx + OPB, bb, boff, first, np);
      } else {
        if (b_first) glds_tile<TRB, BT, NW>(nx + OPB, B, g.ldb, n0, k0 + GBK, g.N, kend, lane, wave);
        glds_tile<TRA, BT, NW>(nx, A, g.lda, m0, k0 + GBK, g.M, kend, lane, wave);
        if (!b_first) glds_tile<TRB, BT, NW>(nx + OPB, B, g.ldb, n0, k0 + GBK, g.N, kend, lane, wave);
      }
    }
    if (MP_DBG(g, 1)) continue;
    if (EPI == EPI_SLAB && TRA == 1 && BT != 256 && !MP_DBG(g, 8)) {      // 128-wide tiles: column sums from the LDS image
      if (tn == 0 && tid < BT) {
        const int oc = tid >> 3, wi = tid & 7;
        for (int r = 0; r < GBK; ++r)
          bsum += lds16_to_float<SPLIT == 16>(*reinterpret_cast<const unsigned short*>(As + r * (BT * 2) + ((oc ^ t_swz(r)) << 4) + wi * 2));
      }
    }
    if (EPI == EPI_SLAB && TRA == 1 && BT == 256 && bias_here && ((k0 - kbeg) / GBK) % bias_parts == tn && !MP_DBG(g, 8)) {
      // 256-wide tiles: the bias gradient (column sums of the A operand over the reduction) on the matrix cores - ones x fragment, four
      // MFMAs and four fragment reads per wave and k-tile (wave w owns the output rows 32 w .. 32 w + 31 of the tile); the element-wise
      // LDS walk above cost 13 % of the kernel at this tile size
      union { unsigned u[4]; bf16x8_t v; } ones;
      ones.u[0] = ones.u[1] = ones.u[2] = ones.u[3] = (SPLIT == 16) ? 0x3C003C00u : 0x3F803F80u;      // 1.0 in fp16 / bf16
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int ks = 0; ks < GBK / 32; ++ks)
          accb[f] = mfma16<SPLIT == 16>(ones.v, read_frag2<1, BT>(As, (2 * wave + f) * 16, ks, lane), accb[f]);
    }
    mma_stage<TRA, TRB, BT, 0, (SPLIT == 16), NoMid, (SPLIT == 32)>(As, Bs, acc, wr, wc, lane);      // SPLIT 32: the B operand (X) is an fp16 plane, converted per fragment
  }
  }

  if (MP_DBG(g, 4) && acc[0][0][0] != 12345.678f) return;
  // ---- epilogue through LDS: the accumulators (transposed-tile layout: lane = row, 4 consecutive columns per register
  // group) are written to a wave-private 64 x 64 fp32 image (16-byte chunks XOR-swizzled by the row) and read back row-major,
  // so bias / residual / pre-activation loads and the output stores are full 128/256-byte lines, 16 lanes per row. ----
  TC* C = reinterpret_cast<TC*>(g.C);
  if (EPI == EPI_SLAB) {
    C += (long)tz * g.M * g.ldc;
    if (BT == 256) {
      if (TRA == 1 && bias_here) {
#pragma unroll
        for (int f = 0; f < 2; ++f) {          // every column of the 16 x 16 result holds the sums; lane = output row
          const int row = m0 + (2 * wave + f) * 16 + (lane & 15);
          if (lane < 16 && row < g.M) g.bias_slab[((long)tz * bias_parts + tn) * g.M + row] = accb[f][0];
        }
      }
    } else if (TRA == 1 && tn == 0 && tid < BT && m0 + tid < g.M && g.bias_slab != nullptr)
      g.bias_slab[(long)tz * g.M + m0 + tid] = bsum;
  }
  typename ZType<TC>::type* Z = reinterpret_cast<typename ZType<TC>::type*>(g.Z);
  const long lo_off = c_lo_off<TC>(g);
  const float rsc = g.rscale != 0.f ? g.rscale : 1.0f;     // residual scale (muP: 1 / sqrt(depth))
  const float gout_v = (EPI == EPI_DGELU && g.gout != nullptr) ? *g.gout : 0.f;
  __syncthreads();                                   // every wave is done with the operand stages: reuse them
  float* img = reinterpret_cast<float*>(smem + wave * 16384);
  const int l15 = lane & 15, gq = lane >> 4;
  const int col = n0 + wc * 64 + 4 * l15;            // read-back mapping: 16 lanes cover the 64 columns of a row
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (EPI != EPI_SLAB && EPI != EPI_DGELU && g.bias != nullptr && col < g.N) bias4 = ld4(g.bias + col);
  DropPathRows dp;
  dp.init(EPI == EPI_BIAS_RESID ? g.mask : nullptr, g.mask_mode, g.T, g.J, m0 + wr * (BT / 2));
  float4 rg4 = make_float4(0.f, 0.f, 0.f, 0.f), rb4 = rg4;
  if (EPI == EPI_BIAS_RESID && g.rstats != nullptr && col < g.N) { rg4 = ld4(g.rgamma + col); rb4 = ld4(g.rbeta + col); }
#pragma unroll
  for (int hp = 0; hp < MI / 4; ++hp) {              // 64 rows of the wave tile per pass
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
      const int lr = ii * 16 + l15;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ch = (4 * j + gq) ^ (lr & 15);
        *reinterpret_cast<f32x4*>(img + lr * 64 + ch * 4) = acc[hp * 4 + ii][j];
      }
    }
    // row statistics of the 64 rows of this pass, one row per lane (a single coalesced load), handed out by shuffles below
    float2 rs_lane = make_float2(0.f, 1.f);
    if (EPI == EPI_BIAS_RESID && g.rstats != nullptr)
      rs_lane = *reinterpret_cast<const float2*>(g.rstats + 2 * (long)min(m0 + wr * (BT / 2) + hp * 64 + lane, g.M - 1));
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
      const int lr = it * 4 + gq;
      const int row = m0 + wr * (BT / 2) + hp * 64 + lr;
      float4 v = *reinterpret_cast<const float4*>(img + lr * 64 + ((l15 ^ (lr & 15)) << 2));
      // (mean, rstd) of this row out of the wave's prefetched statistics - shuffled while every lane is still active (the source lane
      // of a row can belong to a lane group whose own row lies past M in the last tile)
      float mean = 0.f, rstd = 1.f;
      if (EPI == EPI_BIAS_RESID) { mean = __shfl(rs_lane.x, lr, 64); rstd = __shfl(rs_lane.y, lr, 64); }
      if (row >= g.M || col >= g.N) continue;
      const long o = (long)row * g.ldc + col;
      v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
      if (EPI == EPI_BIAS_GELU) {
        const float4 d = gelu_fwd4_fast(v);     // v <- gelu(v); Z keeps gelu' (all the backward needs from the pre-activation); null: inference
        if (Z != nullptr) st4(Z + o, d);
      } else if (EPI == EPI_BIAS_RESID) {
        const float dscale = dp.scale(row);
        float4 r = ld4(g.R + o);
        if (g.rstats != nullptr) {               // the residual is LayerNorm(R), recomputed (ln_fwd stage-1 expression)
          // (mean, rstd) of this row out of the wave's prefetched statistics (one coalesced load per 64 rows, handed out by the shuffles
          // above), not a per-row float2 load consumed on the spot: that form - compiled to a v_pk_mul_f32 with op_sel on the loaded pair -
          // was the site of round 3's wrong rows (common.h, "packed-fp32 guard": cause not identified; the form no longer exists)
          mean = lone(mean); rstd = lone(rstd);
          r = make_float4((r.x - mean) * rstd * rg4.x + rb4.x, (r.y - mean) * rstd * rg4.y + rb4.y, (r.z - mean) * rstd * rg4.z + rb4.z,
                          (r.w - mean) * rstd * rg4.w + rb4.w);
        }
        v = make_float4(r.x * rsc + dscale * v.x, r.y * rsc + dscale * v.y, r.z * rsc + dscale * v.z, r.w * rsc + dscale * v.w);
      } else if (EPI == EPI_DGELU) {
        const float4 z = ld4(Z + o);
        v = make_float4(v.x * z.x, v.y * z.y, v.z * z.z, v.w * z.w);      // Z holds gelu'(pre-activation)
      }
      if constexpr (EPI == EPI_DGELU && sizeof(TC) == 2) {
        if (g.gout != nullptr) st4_f16(C + o, v, gout_v, g.gsat);
        else st4(C + o, v, lo_off);
      } else st4(C + o, v, lo_off);
    }
  }
}

// =================================================================================================================
// Persistent form of the 256 x 256 direct-to-LDS kernel, for problems with many more tiles than CUs (the forward and
// dgrad GEMMs of the token stream: K = 512..1536 is only 8..24 k-tiles, so the per-tile fixed costs - workgroup launch,
// the latency of the first DMA, the store drain - are as long as the main loop).  One workgroup per CU walks the tiles
// id, id + G, ...; the two-stage k pipeline runs ACROSS tiles (the DMA of the next tile's first k-tile is issued during
// the last k-tile of the current one), and the epilogue transposes through a wave-private 4 KiB LDS image that lies
// outside the operand stages, so its stores drain while the next tile's first k-tile is multiplied.
// Tile order: workgroup w runs on XCD w & 7 (round-robin dispatch); in every round each XCD owns G/8 consecutive
// row-major tiles, i.e. a few complete tile rows: the A rows are shared through that XCD's L2, the weights stay in it.
// =================================================================================================================
// Epilogue of one wave's 128 x 64 sub-tile: 16 rows per pass are written to the wave-private image in the accumulator layout
// (lane = row, 4 consecutive columns; 16-byte chunks XOR-swizzled by the row) and read back row-major, so residual / gelu'
// loads and the stores are full lines, 16 lanes per row.  FULL = the tile has no rows past M: no predicates, and the
// loads of a pass are issued together ahead of its stores.
// F16G (gelu'-multiplying dgrad): dz leaves as saturating scaled fp16 (GemmB16Args::gout) - a template parameter, not a run-time branch: inlined 64 times,
// the saturating store with its slow path and counters made this kernel 11 600 instructions (93 KB, more than the 64 KB instruction cache) and left
// its matrix cores 29 % busy where the plain dgrad reaches 47 % (profiles/r04_bf16x3_B79_pmc_mfma_util.csv), although the default backward never takes it
template <typename TC, int EPI, bool FULL, bool F16G = false>
__device__ __forceinline__ void persist_epilogue(const GemmB16Args& g, const f32x4 (&acc)[8][4], float* __restrict__ img, int row0, int col,
                                                 const float4& bias4, TC* __restrict__ C, typename ZType<TC>::type* __restrict__ Z, int l15, int gq) {
  constexpr bool LOADS = (EPI == EPI_BIAS_RESID || EPI == EPI_DGELU);
  const long lo_off = c_lo_off<TC>(g);
  const float gout_v = (EPI == EPI_DGELU && F16G) ? *g.gout : 0.f;      // this backward's gradient scale (a device scalar)
  float4 in_nxt[4];
  float ds_nxt[4];
  // residual = LayerNorm(R) recomputed from R and its row statistics (GemmB16Args::rstats): per-lane gamma / beta of its 4 columns
  const bool resid_ln = EPI == EPI_BIAS_RESID && g.rstats != nullptr;
  float2 rs_nxt[4];
  float4 rg4 = make_float4(0.f, 0.f, 0.f, 0.f), rb4 = rg4;
  if (resid_ln) { rg4 = ld4(g.rgamma + (FULL ? col : min(col, g.N - 4))); rb4 = ld4(g.rbeta + (FULL ? col : min(col, g.N - 4))); }
  // the residual / gelu' rows (and DropPath scales) of pass i+1 are requested BEFORE the stores of pass i are issued: memory
  // operations retire in order, so a pass never waits for the previous pass's stores.
  DropPathRows dp;
  dp.init(EPI == EPI_BIAS_RESID ? g.mask : nullptr, g.mask_mode, g.T, g.J, row0);
  auto request = [&](int i) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = row0 + i * 16 + it * 4 + gq;
      const int rc = FULL ? row : min(row, g.M - 1);
      const long o = (long)rc * g.ldc + col;
      if (LOADS) in_nxt[it] = (EPI == EPI_DGELU) ? ld4(Z + o) : ld4(g.R + o);
      ds_nxt[it] = 1.0f;
      if (EPI == EPI_BIAS_RESID) ds_nxt[it] = dp.scale(rc);
      rs_nxt[it] = resid_ln ? *reinterpret_cast<const float2*>(g.rstats + 2 * (long)rc) : make_float2(0.f, 1.f);
    }
  };
  if (LOADS) request(0);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float4 in[4];
    float ds[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      in[it] = in_nxt[it]; ds[it] = ds_nxt[it];
      if (EPI == EPI_BIAS_RESID && resid_ln) {     // ln_fwd stage-1 expression
        const float mean = lone(rs_nxt[it].x), rstd = lone(rs_nxt[it].y);      // common.h: loaded pair, splat over packed lanes
        in[it] = make_float4((in[it].x - mean) * rstd * rg4.x + rb4.x, (in[it].y - mean) * rstd * rg4.y + rb4.y,
                             (in[it].z - mean) * rstd * rg4.z + rb4.z, (in[it].w - mean) * rstd * rg4.w + rb4.w);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(img + l15 * 64 + (((4 * j + gq) ^ l15) << 2)) = acc[i][j];
    if (LOADS && i + 1 < 8) request(i + 1);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int lr = it * 4 + gq;
      const int row = row0 + i * 16 + lr;
      float4 v = *reinterpret_cast<const float4*>(img + lr * 64 + ((l15 ^ lr) << 2));
      const long o = (long)row * g.ldc + col;
      v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
      if (EPI == EPI_BIAS_GELU) {
        const float4 d = gelu_fwd4_fast(v);     // v <- gelu(v); Z keeps gelu' (all the backward needs from the pre-activation)
        if ((FULL || row < g.M) && Z != nullptr) st4(Z + o, d);      // Z null: inference, nobody reads gelu'
      } else if (EPI == EPI_BIAS_RESID) {
        v = make_float4(in[it].x + ds[it] * v.x, in[it].y + ds[it] * v.y, in[it].z + ds[it] * v.z, in[it].w + ds[it] * v.w);
      } else if (EPI == EPI_DGELU) {
        v = make_float4(v.x * in[it].x, v.y * in[it].y, v.z * in[it].z, v.w * in[it].w);      // Z holds gelu'(pre-activation)
      }
      if (FULL || row < g.M) {
        if constexpr (EPI == EPI_DGELU && sizeof(TC) == 2 && F16G) st4_f16(C + o, v, gout_v, g.gsat);      // dz as scaled fp16 for an fc1 layer whose backward GEMMs run on fp16 operands
        else st4(C + o, v, lo_off);
      }
    }
  }
}

// bf16 output, no bias / residual / activation (the dgrad GEMMs): the accumulators are rounded to bf16 BEFORE the transposition, so the
// wave-private image is 16 rows x 128 bytes per pass (half the LDS traffic of the fp32 image) and every lane leaves with 16 bytes per
// store instruction (two dwordx4 per pass instead of four dwordx2).  Same values: the rounding is that of st4(bf16*).
// Image: row r, 8-byte chunk q (= 4 columns) at r * 128 + ((q ^ (r & 14)) << 3) - conflict-free for the 8-byte writes (lane = row, one
// chunk per instruction) and for the 16-byte row-major reads (an aligned pair of chunks stays a pair).
template <bool FULL>
__device__ __forceinline__ void persist_epilogue_bf16_packed(const GemmB16Args& g, const f32x4 (&acc)[8][4], char* __restrict__ img, int row0, int col0,
                                                             bf16* __restrict__ C, int l15, int gq, int lane) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint2 v;
      v.x = pack_bf16x2(acc[i][j][0], acc[i][j][1]);
      v.y = pack_bf16x2(acc[i][j][2], acc[i][j][3]);
      *reinterpret_cast<uint2*>(img + l15 * 128 + (((4 * j + gq) ^ (l15 & 14)) << 3)) = v;
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int r = it * 8 + (lane >> 3), c16 = lane & 7;
      const uint4 v = *reinterpret_cast<const uint4*>(img + r * 128 + ((c16 ^ ((r >> 1) & 7)) << 4));
      const int row = row0 + i * 16 + r;
      if (FULL || row < g.M) *reinterpret_cast<uint4*>(C + (long)row * g.ldc + col0 + 8 * c16) = v;
    }
  }
}

// (Round 4: the residual epilogue below executes ~2700 instructions per tile and wave - the planar bias epilogue ~1100 - a third of them 64-bit
// row-offset multiplies and the two scalar branches of each of its 32 DropPath mask lookups.  A form with running row offsets and the mask kind /
// recomputed-LayerNorm flag as template parameters of the kernel (378 -> 54 branches, 5800 -> 3200-4300 instructions in the kernel) was built,
// passed the parity tests and moved nothing: proj 543 / 544 vs 546 / 553 us, forward-GEMM class 63.8 / 64.1 vs 63.9 / 64.1 ms, step 162.7 vs
// 162.6 ms (profiles/r04_probes/ab_step5_resid_epilogue_special.log).  The epilogue is not bound by its instruction stream.  Not kept.)
// (Round 4, tools/probes/epi_shapes.hip + profiles/r04_probes/: what an epilogue costs is set by the CU's store path and by every CU storing at
// once - the same 256 KiB of planar output cost 6.5 us per tile as dwordx2 stores of 4 rows x 128 B (this epilogue), 4.4 us as dwordx4 stores
// of 8 rows x 128 B, 8 us with the MFMA columns permuted so that no LDS transposition is needed (16 rows x 64 B per instruction), 13 us in the
// natural accumulator layout (16 rows x 32 B), 3 us as contiguous 16 KiB blocks; one CU in 32 storing: 4.2 / 2.1 / 5.7 / 11.6 / 0.2 us.  The
// dwordx4 form of this epilogue for the planar outputs was built and measured in the real kernels: qkv 1356 / 1344 -> 1338 / 1341 us, fc1
// unchanged, the step unchanged - not kept.)
// (diagnostics build: per-wave cycle totals of the hand-scheduled loops - wait, barrier, MFMA stage)
#ifdef MP_GEMM_DIAG
#define MP_KDIAG_A() tk1 = __builtin_readcyclecounter()
#define MP_KDIAG_B() tk2 = __builtin_readcyclecounter(); dg_wait += tk1 - tk0; dg_bar += tk2 - tk1
#define MP_KDIAG_C() tk0 = __builtin_readcyclecounter(); dg_mma += tk0 - tk2
#else
#define MP_KDIAG_A()
#define MP_KDIAG_B()
#define MP_KDIAG_C()
#endif
// SPLIT: see gemm_bf16_glds_kernel (three steps per k-tile, the DMA source planes rotate)
template <int TRB, typename TC, int EPI, int SPLIT = 0, bool F16G = false>
__global__ __launch_bounds__(512) void gemm_bf16_persist_kernel(GemmB16Args g, int tiles_n, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BT = 256, WN = 4, MI = 8, OPB = BT * 128, STAGE = 2 * OPB;
  const char* const A = reinterpret_cast<const char*>(g.A);
  const char* const B = reinterpret_cast<const char*>(g.B);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WN, wc = wave % WN;
  const int per_xcd = gridDim.x >> 3;                       // gridDim.x is a multiple of 8
  // (Round 4: giving every XCD ONE contiguous range of whole tile rows for the whole launch - so that no activation row panel is shared between
  // two L2s - changes nothing: block of four split-precision GEMMs 3898 / 3856 us -> 3899 / 3875 us, profiles/r04_probes/gemm_block_ab.log.)
  int id = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (id >= ntiles) return;
  const int nk = g.K / GBK;                                 // K % 64 == 0, nk >= 2 (launcher)
  int m0 = (id / tiles_n) * BT, n0 = (id % tiles_n) * BT;
  constexpr bool KASM8 = MP_KLOOP_ASM && MP_KLOOP_ASM8 && MP_KSTEP_YOUNG && SPLIT == 8 && TRB == 0;      // f16f8 loop, hand-scheduled (round 6)
  constexpr bool KASM = MP_KLOOP_ASM && (SPLIT == 0 || SPLIT == 1 || KASM8);      // the hand-scheduled k-step (plain, three-product and f16f8 loops)
  constexpr bool YOUNG = KASM && MP_KSTEP_YOUNG;                         // its operand DMA comes from waves 4-7 only (8 pieces of a tile each)
  constexpr int NP = YOUNG ? 8 : 4;
  const int dma_first = YOUNG ? (wave >= 4 ? (wave - 4) * 8 : 0) : wave * 4;      // this wave's first piece of an operand tile
  unsigned aoff[NP], boff[NP];
  persist_offsets_n<0, NP>(aoff, g.lda, m0, g.M, lane, dma_first);
  persist_offsets_n<TRB, NP>(boff, g.ldb, 0, BT, lane, dma_first);       // N % 256 == 0: the same for every tile
  const unsigned smem_l = lds_u32(smem), dma_l = smem_l + dma_first * 1024;   // LDS addresses: the stages' origin, this wave's pieces of an operand tile
  const int dma_wave = YOUNG ? kflag_lt(3, wave) : 1;                    // does this wave issue operand DMA (a scalar-register flag)
  KFragA kfa[2];      // fragment addresses in the A buffer of stage 0 / 1 (split loop: A0 = A_lo, A1 = A_hi)
  KFragB kfb[2];      // ... in the B buffer of stage 0 / 1 (split loop: B0 = B_hi, B1 = B_lo)
  if constexpr (KASM) {
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) { kfrag_a(kfa[sg], smem_l + sg * STAGE, wr, lane); kfrag_b<TRB>(kfb[sg], smem_l + sg * STAGE + OPB, wc, lane); }
  }
  const unsigned img_l = smem_l + 2 * STAGE + wave * 4096;               // this wave's epilogue image (the step's bias DMA lands there)
  // byte address of (tile origin, reduction index k) of each operand
  // byte address of (tile origin, k-tile kt) of each operand plane
  auto a_base = [&](int mm, int kt) { return A + ((long)mm * g.lda + kt * GBK) * 2; };
  auto b_base = [&](int nn, int kt) { return TRB ? B + ((long)kt * GBK * g.ldb + nn) * 2 : B + ((long)nn * g.ldb + kt * GBK) * 2; };
  // SPLIT: A / B above are the hi planes; the lo planes lie at these byte distances
  constexpr bool PLANES = (SPLIT & 1) || SPLIT == 8;        // two planes per operand (SPLIT 0 / 16: one, bf16 / fp16)
  const long a_lo = PLANES ? reinterpret_cast<const char*>(g.A_lo) - A : 0, b_lo = PLANES ? reinterpret_cast<const char*>(g.B_lo) - B : 0;
  if constexpr (YOUNG) {
    if (wave >= 4) {
      uneven_dma<NP>(smem, a_base(m0, 0) + (SPLIT == 8 ? 0 : a_lo), aoff, dma_first, NP);      // SPLIT 1: the first step multiplies (A_lo, B_hi); 8: the fp16 planes
      uneven_dma<NP>(smem + OPB, b_base(n0, 0), boff, dma_first, NP);
    }
  } else {
    persist_dma(smem, a_base(m0, 0) + (SPLIT == 8 ? 0 : a_lo), aoff, wave);       // SPLIT 1: the first step multiplies (A_lo, B_hi); 8: the fp16 planes
    persist_dma(smem + OPB, b_base(n0, 0), boff, wave);
  }
  int stage = 0;
  bool landed = false;                                      // this tile's first k-tile was already waited for (before the previous epilogue)
  TC* const C = reinterpret_cast<TC*>(g.C);
  typename ZType<TC>::type* const Z = reinterpret_cast<typename ZType<TC>::type*>(g.Z);
  float* const img = reinterpret_cast<float*>(smem + 2 * STAGE + wave * 4096);
  const bool has_bias = EPI != EPI_DGELU && g.bias != nullptr;
#ifdef MP_GEMM_DIAG                                          // diagnostics build (MP_DIAG=1 build.sh -> libmanipose_hip_diag.so, tools/gemm_stamps.py)
  int tile_no = 0;
  // per-wave shader-clock totals of the main loop: cycles parked on s_waitcnt (operand DMA), on the barrier behind it, and in the MFMA stage
  unsigned long long dg_wait = 0, dg_bar = 0, dg_mma = 0, dg_epi = 0;
  if (g.stagger > 0) {                                      // start the workgroups of an XCD in four phase groups
    const int phase = (blockIdx.x >> 3) & 3;
    if (phase) {
      const long long t0 = wall_clock64(), dt = (long long)phase * g.stagger;
      while (wall_clock64() - t0 < dt) __builtin_amdgcn_s_sleep(16);
    }
  }
#endif

  while (true) {
    const int idn = id + gridDim.x;
    const bool has_next = idn < ntiles;
    const int m0n = (idn / tiles_n) * BT, n0n = (idn % tiles_n) * BT;
    // (hand-scheduled k-step) hi-plane byte addresses of this tile's and the next tile's first k-tile, and the B operand's step per k-tile
    const char* const tile_a = a_base(m0, 0);
    const char* const tile_b = b_base(n0, 0);
    const char* const tile_an = a_base(m0n, 0);
    const char* const tile_bn = b_base(n0n, 0);
    const long kstep_b = TRB ? (long)GBK * g.ldb * 2 : GBK * 2;
    const int has_next_i = KASM ? kflag_lt(idn, ntiles) : 0, has_bias_i = (KASM && EPI != EPI_DGELU) ? kflag_nonnull(g.bias) : 0;
    f32x4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if constexpr (KASM8) {
      // "f16f8", hand-scheduled: per k-tile one F block (fp16 tiles in stage 0; requests the k-tile's correction tiles into stage 1) and one E block
      // (correction tiles in stage 1; requests the fp16 tiles of the next k-tile / tile into stage 0, the bias behind the tile's last step)
      const char* pa = tile_a;                             // fp16 plane of A, k-tile kt
      const char* pb = tile_b;
      const unsigned coff = 4u * lane;
      const KFragA fa1 = {{kfa[0].a[0] + (unsigned)STAGE, kfa[0].a[1] + (unsigned)STAGE}};
      const KFragB fb1 = {{kfb[0].b[0] + (unsigned)STAGE, kfb[0].b[1] + (unsigned)STAGE, 0u, 0u}};
      const int sca = F8_SCALE_A, scb = F8_SCALE_B;
      for (int kt = 0; kt < nk; ++kt, pa += GBK * 2, pb += kstep_b) {
        const bool last = kt + 1 == nk;
        const int more_i = kflag_more(kt + 1, nk, has_next_i);
#ifdef MP_GEMM_DIAG
        unsigned long long tk0 = __builtin_readcyclecounter(), tk1, tk2;
#endif
        if (kt == 0 && landed) __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0) only
        else __builtin_amdgcn_s_waitcnt(0x0070);                        // vmcnt(0) lgkmcnt(0)
        MP_KDIAG_A();
        __builtin_amdgcn_s_barrier();
        MP_KDIAG_B();
        kstep_asm_f(acc, kfa[0], kfb[0], KJob{dma_wave, dma_l + STAGE, pa + a_lo}, aoff, KJob{dma_wave, dma_l + STAGE + OPB, pb + b_lo}, boff);
        MP_KDIAG_C();
        if (last && more_i) persist_offsets_n<0, NP>(aoff, g.lda, m0n, g.M, opaque(lane), dma_first);      // (opaque: the rows are recomputed here, not kept - spilled - across the tile)
        __builtin_amdgcn_s_waitcnt(0x0070);
        MP_KDIAG_A();
        __builtin_amdgcn_s_barrier();
        MP_KDIAG_B();
        kstep_asm_e(acc, fa1, fb1, KJob{more_i & dma_wave, dma_l, last ? tile_an : pa + GBK * 2}, aoff, KJob{more_i & dma_wave, dma_l + OPB, last ? tile_bn : pb + kstep_b}, boff,
                    KJob{kflag_last(kt + 1, nk, has_bias_i), img_l, reinterpret_cast<const char*>(g.bias + n0 + wc * 64)}, coff, sca, scb);
        MP_KDIAG_C();
      }
      (void)stage;
    } else if constexpr (SPLIT == 8) {
      // "f16f8": 2 nk steps over the two stages - even steps multiply the fp16 planes of k-tile v / 2 (fp16 MFMA), odd steps its 8-bit
      // correction planes (one 128-deep fp8 MFMA per 16 x 16 block); the plain kernel's pipeline with alternating source planes
      for (int v = 0; v < 2 * nk; ++v) {
        if (v == 0 && landed) __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0) only
        else __builtin_amdgcn_s_waitcnt(0x0070);                       // vmcnt(0) lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        const int par = v & 1;
        const char* As = smem + par * STAGE;
        const char* Bs = As + OPB;
        char* nx = smem + (par ^ 1) * STAGE;
        const bool last = v + 1 == 2 * nk;
        if (!last || has_next) {
          if (last) persist_offsets_n<0, NP>(aoff, g.lda, m0n, g.M, opaque(lane), dma_first);      // (opaque: the rows are recomputed here, not kept - spilled - across the tile)
          const int ktn = last ? 0 : (v + 1) >> 1;
          persist_dma(nx, a_base(last ? m0n : m0, ktn) + (par ? 0 : a_lo), aoff, wave);
          persist_dma(nx + OPB, b_base(last ? n0n : n0, ktn) + (par ? 0 : b_lo), boff, wave);
        }
        if (last && has_bias) {
          typedef __attribute__((address_space(3))) void* lptr;
          typedef const __attribute__((address_space(1))) void* gptr;
          __builtin_amdgcn_global_load_lds((gptr)(g.bias + n0 + wc * 64 + lane), (lptr)img, 4, 0, 0);
        }
        mma_stage_mix(As, Bs, acc, wr, wc, lane, par != 0);
      }
      (void)stage;
    } else if constexpr (SPLIT & 1) {
      // three steps per k-tile over four fetched operand tiles; buffers A0 | B0 | A1 | B1 = the two stages (see gemm_bf16_glds_kernel)
      char* const A0 = smem;
      char* const B0 = smem + OPB;
      char* const A1 = smem + STAGE;
      char* const B1 = smem + STAGE + OPB;
      // one loop body for the three steps (a single copy of the MFMA stage: three inlined copies spill at the 256-VGPR limit); the
      // buffers of a step are wave-uniform selects
      // DMA schedule: step 0 requests A_hi[kt] (used by step 1), step 1 requests B_lo[kt] (step 2) and then - A0 is free as soon as step 0
      // is over - A_lo of the NEXT k-tile / tile (used two steps later), step 2 requests the next B_hi (one step later).  Step 2 waits with
      // vmcnt(4): requests retire in order, so B_lo has landed while the four A_lo requests issued behind it may still be in flight -
      // one of the two tiles the next step 0 needs has two steps to arrive instead of one.
      bool early = false;
      if constexpr (KASM) {
        // hand-scheduled form: the same buffers, DMA schedule and waits, the three steps of a k-tile written out (one asm block each, with the
        // DMA jobs that step can carry: KC_X0 / X1 / X2); running operand addresses, fragment addresses per buffer: a step's scalar set-up
        // is a handful of instructions
        const char* pa = tile_a;                           // A_hi of k-tile kt
        const char* pb = tile_b;                           // B_hi of k-tile kt
        const KJob none = {0, 0u, nullptr};
        const unsigned coff = 4u * lane;
        constexpr bool PIPE = MP_KLOOP_PIPE && YOUNG && TRB == 0;
        if constexpr (PIPE) {
          // k-tiles 0 .. nk-2 in one block (pipeline across the steps, barriers inside), then the last k-tile block by block: its step 0 needs no
          // barrier (A0, B0 landed before the block's last barrier; it requests nothing - A_hi, B_lo of this k-tile are on their way since then)
          if (landed) __builtin_amdgcn_s_waitcnt(0xC07F);
          else __builtin_amdgcn_s_waitcnt(0x0070);
          __builtin_amdgcn_s_barrier();
#ifdef MP_GEMM_DIAG
          const unsigned long long tp0 = __builtin_readcyclecounter();
#endif
          kpipe_x3(acc, kfa, kfb, aoff, boff, dma_wave, dma_l, pa, pb, a_lo, b_lo, nk - 1);
#pragma unroll
          for (int i = 0; i < NP; ++i) boff[i] -= (unsigned)(nk - 1) * (GBK * 2);      // (the B offsets are the same for every tile: back to k-tile 0)
          pa += (long)(nk - 1) * GBK * 2; pb += (long)(nk - 1) * kstep_b;
          const int more_i = has_next_i;
          kstep_asm<TRB, KC_X0>(acc, kfa[0], kfb[0], none, aoff, none, boff, none, coff);
          if (more_i) persist_offsets_n<0, NP>(aoff, g.lda, m0n, g.M, opaque(lane), dma_first);
#ifdef MP_GEMM_DIAG
          unsigned long long tk0 = __builtin_readcyclecounter(), tk1, tk2;
          dg_mma += tk0 - tp0;
#endif
          __builtin_amdgcn_s_waitcnt(0x0078);               // vmcnt(8): A_hi has landed, B_lo behind it may still be in flight
          MP_KDIAG_A();
          __builtin_amdgcn_s_barrier();
          MP_KDIAG_B();
          kstep_asm<TRB, KC_X1>(acc, kfa[1], kfb[0], KJob{more_i & dma_wave, dma_l, tile_an + a_lo}, aoff, none, boff, none, coff);
          MP_KDIAG_C();
          if (more_i) __builtin_amdgcn_s_waitcnt(0x0078);   // B_lo has landed, the next tile's A_lo may still be in flight
          else __builtin_amdgcn_s_waitcnt(0x0070);
          MP_KDIAG_A();
          __builtin_amdgcn_s_barrier();
          MP_KDIAG_B();
          kstep_asm<TRB, KC_X2>(acc, kfa[1], kfb[1], none, aoff, KJob{more_i & dma_wave, dma_l + OPB, tile_bn}, boff,
                                KJob{has_bias_i, img_l, reinterpret_cast<const char*>(g.bias + n0 + wc * 64)}, coff);
          MP_KDIAG_C();
        } else
        for (int kt = 0; kt < nk; ++kt, pa += GBK * 2, pb += kstep_b) {
          const bool last = kt + 1 == nk;
          const int more_i = kflag_more(kt + 1, nk, has_next_i);
#ifdef MP_GEMM_DIAG
          unsigned long long tk0 = __builtin_readcyclecounter(), tk1, tk2;
#endif
          // step 0: (A0 = A_lo, B0 = B_hi); requests A_hi[kt] -> A1
          if (kt == 0 && landed) __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0) only: (A_lo, B_hi) of this tile were waited for before the previous epilogue
          else __builtin_amdgcn_s_waitcnt(0x0070);
          MP_KDIAG_A();
          __builtin_amdgcn_s_barrier();
          MP_KDIAG_B();
          kstep_asm<TRB, KC_X0>(acc, kfa[0], kfb[0], KJob{dma_wave, dma_l + STAGE, pa}, aoff, none, boff, none, coff);
          MP_KDIAG_C();
          // step 1: (A1 = A_hi, B0 = B_hi); requests B_lo[kt] -> B1, then A_lo of the next k-tile / tile -> A0
          if (last && more_i) persist_offsets_n<0, NP>(aoff, g.lda, m0n, g.M, opaque(lane), dma_first);      // (opaque: the rows are recomputed here, not kept - spilled - across the tile)
          __builtin_amdgcn_s_waitcnt(0x0070);
          MP_KDIAG_A();
          __builtin_amdgcn_s_barrier();
          MP_KDIAG_B();
          kstep_asm<TRB, KC_X1>(acc, kfa[1], kfb[0], KJob{more_i & dma_wave, dma_l, (last ? tile_an : pa + GBK * 2) + a_lo}, aoff, KJob{dma_wave, dma_l + STAGE + OPB, pb + b_lo}, boff,
                                none, coff);
          MP_KDIAG_C();
          // step 2: (A1 = A_hi, B1 = B_lo); requests B_hi of the next k-tile / tile -> B0, and the bias behind the tile's last step.  It waits with
          // vmcnt(4): B_lo has landed, the four A_lo requests issued behind it may still be in flight
          if (more_i) __builtin_amdgcn_s_waitcnt(YOUNG ? 0x0078 : 0x0074);      // (younger-wave DMA: the A_lo job behind B_lo is 8 pieces)
          else __builtin_amdgcn_s_waitcnt(0x0070);
          MP_KDIAG_A();
          __builtin_amdgcn_s_barrier();
          MP_KDIAG_B();
          kstep_asm<TRB, KC_X2>(acc, kfa[1], kfb[1], none, aoff, KJob{more_i & dma_wave, dma_l + OPB, last ? tile_bn : pb + kstep_b}, boff,
                                KJob{kflag_last(kt + 1, nk, has_bias_i), img_l, reinterpret_cast<const char*>(g.bias + n0 + wc * 64)}, coff);
          MP_KDIAG_C();
        }
      } else
      for (int kt = 0, term = 0; kt < nk;) {
#ifdef MP_GEMM_DIAG
        const unsigned long long tk0 = __builtin_readcyclecounter();
#endif
        if (kt == 0 && term == 0 && landed) __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0) only: (A_lo, B_hi) of this tile were waited for before the previous epilogue
        else if (term == 2 && early) __builtin_amdgcn_s_waitcnt(0x0074);             // vmcnt(4) lgkmcnt(0)
        else __builtin_amdgcn_s_waitcnt(0x0070);
#ifdef MP_GEMM_DIAG
        const unsigned long long tk1 = __builtin_readcyclecounter();
#endif
        __builtin_amdgcn_s_barrier();
#ifdef MP_GEMM_DIAG
        const unsigned long long tk2 = __builtin_readcyclecounter();
        dg_wait += tk1 - tk0; dg_bar += tk2 - tk1;
#endif
        const char* As = term == 0 ? A0 : A1;
        const char* Bs = term == 2 ? B1 : B0;
        const bool last = kt + 1 == nk;
        if (term == 0) {
          persist_dma(A1, a_base(m0, kt), aoff, wave);                    // A_hi[kt]
        } else if (term == 1) {
          persist_dma(B1, b_base(n0, kt) + b_lo, boff, wave);             // B_lo[kt]
          early = !last || has_next;
          if (early) {
            if (last) persist_offsets_n<0, NP>(aoff, g.lda, m0n, g.M, opaque(lane), dma_first);      // (opaque: the rows are recomputed here, not kept - spilled - across the tile)
            persist_dma(A0, a_base(last ? m0n : m0, last ? 0 : kt + 1) + a_lo, aoff, wave);      // A_lo of the next k-tile / tile
          }
        } else {
          if (!last || has_next) persist_dma(B0, b_base(last ? n0n : n0, last ? 0 : kt + 1), boff, wave);   // B_hi of the next k-tile / tile
          if (last && has_bias) {
            typedef __attribute__((address_space(3))) void* lptr;
            typedef const __attribute__((address_space(1))) void* gptr;
            __builtin_amdgcn_global_load_lds((gptr)(g.bias + n0 + wc * 64 + lane), (lptr)img, 4, 0, 0);
          }
        }
        mma_stage<0, TRB, BT, (SPLIT >> 1)>(As, Bs, acc, wr, wc, lane);  // term 0: A_lo B_hi, 1: A_hi B_hi, 2: A_hi B_lo
#ifdef MP_GEMM_DIAG
        dg_mma += __builtin_readcyclecounter() - tk2;
#endif
        if (++term == 3) { term = 0; ++kt; }
      }
      (void)stage;
    } else if constexpr (KASM) {
      // plain loop, hand-scheduled form (see above): both operands of the next k-tile / tile into the other stage
      const char* pa = tile_a + GBK * 2;                   // A of k-tile ks + 1
      const char* pb = tile_b + kstep_b;
      const unsigned coff = 4u * lane;
      for (int ks = 0; ks < nk; ++ks, stage ^= 1, pa += GBK * 2, pb += kstep_b) {
        const bool last = ks + 1 == nk;
        const int fetch_i = MP_DBG(g, 2) ? 0 : kflag_more(ks + 1, nk, has_next_i);
        if (last && fetch_i) persist_offsets_n<0, NP>(aoff, g.lda, m0n, g.M, opaque(lane), dma_first);      // (opaque: the rows are recomputed here, not kept - spilled - across the tile)
        // this stage's fragment addresses (stage 0's + 64 KiB): a few vector adds in front of the barrier - NOT kfa[stage], which makes the
        // arrays scratch memory
        const unsigned so = (unsigned)stage * STAGE;
        const KFragA fa = {{kfa[0].a[0] + so, kfa[0].a[1] + so}};
        const KFragB fb = {{kfb[0].b[0] + so, kfb[0].b[1] + so, kfb[0].b[2] + so, kfb[0].b[3] + so}};
#ifdef MP_GEMM_DIAG
        unsigned long long tk0 = __builtin_readcyclecounter(), tk1, tk2;
#endif
        if (ks == 0 && landed) __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0) only
        else __builtin_amdgcn_s_waitcnt(0x0070);                        // vmcnt(0) lgkmcnt(0)
        MP_KDIAG_A();
        __builtin_amdgcn_s_barrier();
        MP_KDIAG_B();
        const unsigned nx = dma_l + (stage ^ 1) * STAGE;
        if (!MP_DBG(g, 1))
          kstep_asm<TRB, KC_P>(acc, fa, fb, KJob{fetch_i & dma_wave, nx, last ? tile_an : pa}, aoff, KJob{fetch_i & dma_wave, nx + OPB, last ? tile_bn : pb}, boff,
                               KJob{kflag_last(ks + 1, nk, has_bias_i), img_l, reinterpret_cast<const char*>(g.bias + n0 + wc * 64)}, coff);
        MP_KDIAG_C();
      }
    } else {
    for (int ks = 0; ks < nk; ++ks, stage ^= 1) {
      // k-tile ks has landed for every wave and nobody still reads the other stage.  On a tile's first k-tile the DMA was
      // waited for before the previous epilogue: do not wait for that epilogue's stores here, they drain under this k-tile.
#ifdef MP_GEMM_DIAG
      const unsigned long long tk0 = __builtin_readcyclecounter();
#endif
      if (ks == 0 && landed) __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0) only
      else __builtin_amdgcn_s_waitcnt(0x0070);                        // vmcnt(0) lgkmcnt(0)
#ifdef MP_GEMM_DIAG
      const unsigned long long tk1 = __builtin_readcyclecounter();
#endif
      __builtin_amdgcn_s_barrier();
#ifdef MP_GEMM_DIAG
      const unsigned long long tk2 = __builtin_readcyclecounter();
      dg_wait += tk1 - tk0; dg_bar += tk2 - tk1;
#endif
      const char* As = smem + stage * STAGE;
      const char* Bs = As + OPB;
      char* nx = smem + (stage ^ 1) * STAGE;
      const bool last = ks + 1 == nk;
      const bool fetch = !MP_DBG(g, 2) && (!last || has_next);
      const bool late = DMA_LATE_STEP >= 0 && wave >= 4;      // the SIMD partners of waves 0-3 multiply first and issue their DMA mid-step
      auto issue = [&]() {
        persist_dma(nx, a_base(last ? m0n : m0, last ? 0 : ks + 1), aoff, wave);
        persist_dma(nx + OPB, b_base(last ? n0n : n0, last ? 0 : ks + 1), boff, wave);
      };
      if (fetch) {
        if (last) persist_offsets_n<0, NP>(aoff, g.lda, m0n, g.M, opaque(lane), dma_first);      // (opaque: the rows are recomputed here, not kept - spilled - across the tile)
        if (!late) issue();
      }
      if (last && has_bias) {      // this wave's 64 bias values -> its (idle) epilogue image, 4 bytes per lane; covered by the vmcnt(0) below
        typedef __attribute__((address_space(3))) void* lptr;
        typedef const __attribute__((address_space(1))) void* gptr;
        __builtin_amdgcn_global_load_lds((gptr)(g.bias + n0 + wc * 64 + lane), (lptr)img, 4, 0, 0);
      }
      if (!MP_DBG(g, 1)) mma_stage<0, TRB, BT, 0, (SPLIT == 16)>(As, Bs, acc, wr, wc, lane, [&]() { if (fetch && late) issue(); });
#ifdef MP_GEMM_DIAG
      dg_mma += __builtin_readcyclecounter() - tk2;
#endif
    }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                               // vmcnt(0): the next tile's first k-tile (issued one k-tile ago)
    landed = true;

    // ---- epilogue, 16 rows of the wave's 128 x 64 sub-tile per pass through the wave-private image ----
#ifdef MP_GEMM_DIAG
    if (g.stamps != nullptr && tid == 0 && tile_no < 64) g.stamps[((long)blockIdx.x * 64 + tile_no) * 2] = wall_clock64();
    const unsigned long long te0 = __builtin_readcyclecounter();
#endif
    if (!MP_DBG(g, 4)) {
      // the lane indices pass through an opaque move so that the epilogue's address arithmetic is redone per tile instead of
      // being hoisted out of the tile loop, where it would sit in ~20 VGPRs across the main loop (the kernel runs at the
      // 256-VGPR limit of two waves per SIMD)
      int e15 = lane & 15, eq = lane >> 4;
      asm volatile("" : "+v"(e15), "+v"(eq));
      const int col = n0 + wc * 64 + 4 * e15;               // < N: N % 256 == 0
      float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (has_bias) bias4 = *reinterpret_cast<const float4*>(img + 4 * e15);
      bool done = false;
      if constexpr (sizeof(TC) == 2 && EPI == EPI_BIAS && (SPLIT == 0 || SPLIT == 16)) {      // plain bf16 output (bf16p is the planar tag: SPLIT kernels only)
        if (!has_bias) {
          int el = lane;
          asm volatile("" : "+v"(el));
          if (m0 + BT <= g.M) persist_epilogue_bf16_packed<true>(g, acc, reinterpret_cast<char*>(img), m0 + wr * 128, n0 + wc * 64, reinterpret_cast<bf16*>(C), e15, eq, el);
          else persist_epilogue_bf16_packed<false>(g, acc, reinterpret_cast<char*>(img), m0 + wr * 128, n0 + wc * 64, reinterpret_cast<bf16*>(C), e15, eq, el);
          done = true;
        }
      }
      if (!done) {
        if (m0 + BT <= g.M) persist_epilogue<TC, EPI, true, F16G>(g, acc, img, m0 + wr * 128, col, bias4, C, Z, e15, eq);
        else persist_epilogue<TC, EPI, false, F16G>(g, acc, img, m0 + wr * 128, col, bias4, C, Z, e15, eq);
      }
    }
#ifdef MP_GEMM_DIAG
    dg_epi += __builtin_readcyclecounter() - te0;
    if (g.stamps != nullptr && tid == 0 && tile_no < 64) g.stamps[((long)blockIdx.x * 64 + tile_no) * 2 + 1] = wall_clock64();
    ++tile_no;
    if (!has_next && g.stamps != nullptr && lane == 0) {      // per-wave totals behind the per-tile stamps: [workgroup][wave][wait, barrier, mma, epilogue]
      long long* o = g.stamps + 256 * 64 * 2 + ((long)blockIdx.x * 8 + wave) * 4;
      o[0] = (long long)dg_wait; o[1] = (long long)dg_bar; o[2] = (long long)dg_mma; o[3] = (long long)dg_epi;
    }
#endif
    if (!has_next) break;
    id = idn; m0 = m0n; n0 = n0n;
  }
}

static int g_persist_min_tiles = 0;        // test hook: 0 = default (2 tiles per workgroup), else the tile count from which the persistent kernel runs
void gemm_bf16_persist_min_tiles(int n) { g_persist_min_tiles = n; }

// mp_set_option("gemm_persist_mode"): 0 tiled kernels only, 1 (default) the persistent kernel where it applies (A/B timing and tests);
// mp_set_option("gemm_persist_wgs"): run the persistent GEMMs on fewer workgroups (= CUs), leaving the rest of the chip to kernels of other
// streams (a persistent workgroup takes a CU's whole LDS and register file); a multiple of 8 (one share per XCD), 0 = all CUs
static int g_persist_mode = 1;
void gemm_bf16_persist_mode(int mode) { g_persist_mode = mode; }
static int g_persist_wgs = 0;
void gemm_bf16_persist_wgs(int n) { g_persist_wgs = n; }
static int persist_workgroups() {
  static int cus8 = -1;
  if (cus8 < 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
    cus8 = (cus / 8) * 8;
  }
  int n = cus8;
  if (g_persist_wgs >= 8 && g_persist_wgs <= n) n = (g_persist_wgs / 8) * 8;
  return g_persist_mode == 0 ? 0 : n;
}

// profiling tag: 1 if the last gemm_bf16() of this thread went to the persistent kernel (the engine's per-kernel timing)
static thread_local int g_last_persist = 0;
int gemm_bf16_take_last_persist() { const int v = g_last_persist; g_last_persist = 0; return v; }

template <int TRB, typename TC, int EPI, int SPLIT, bool F16G>
static int persist_go(const GemmB16Args& g, int wgs, int tiles_n, int ntiles, hipStream_t st) {
  constexpr size_t lds = 2 * 2 * 256 * 128 + 8 * 4096;     // two operand stages + the epilogue images = 160 KiB
  static bool attr_set = false;
  if (!attr_set) {
    MP_HIP(hipFuncSetAttribute((const void*)gemm_bf16_persist_kernel<TRB, TC, EPI, SPLIT, F16G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_bf16_persist_kernel<TRB, TC, EPI, SPLIT, F16G>), dim3(wgs), dim3(512), lds, st, g, tiles_n, ntiles);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

template <int TRB, typename TC, int EPI, int SPLIT = 0>
static int launch_persist(const GemmB16Args& g_in, int wgs, hipStream_t st) {
  g_last_persist = 1;
  GemmB16Args g = g_in;
  g.stamps = nullptr; g.stagger = 0;
#ifdef MP_GEMM_DIAG
  {
    static long long* const stamps = [] { const char* e = getenv("MANIPOSE_GEMM_STAMPS"); return e ? (long long*)strtoull(e, nullptr, 0) : (long long*)nullptr; }();
    g.stamps = stamps;
    static const int stagger = [] { const char* e = getenv("MANIPOSE_GEMM_STAGGER"); return e ? atoi(e) : 0; }();
    g.stagger = stagger;
  }
#endif
  const int tiles_n = cdiv(g.N, 256), ntiles = tiles_n * cdiv(g.M, 256);
  if constexpr (EPI == EPI_DGELU && sizeof(TC) == 2) {
    if (g.gout != nullptr) return persist_go<TRB, TC, EPI, SPLIT, true>(g, wgs, tiles_n, ntiles, st);
  }
  return persist_go<TRB, TC, EPI, SPLIT, false>(g, wgs, tiles_n, ntiles, st);
}

static bool g_force_small_tile = false;    // test hook: exercise the 128x128 instantiation on big shapes too
void gemm_bf16_force_small_tile(bool on) { g_force_small_tile = on; }

template <int TRA, int TRB, typename TC, int EPI, int BT, int SPLIT = 0>
static int launch_glds_bt(const GemmB16Args& g, int splits, hipStream_t st) {
  constexpr size_t lds = 2 * 2 * BT * 128;
  static bool attr_set = false;
  if (!attr_set) {
    MP_HIP(hipFuncSetAttribute((const void*)gemm_bf16_glds_kernel<TRA, TRB, TC, EPI, BT, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  dim3 grid(cdiv(g.N, BT), cdiv(g.M, BT), splits);
  hipLaunchKernelGGL((gemm_bf16_glds_kernel<TRA, TRB, TC, EPI, BT, SPLIT>), grid, dim3(BT * 2), lds, st, g);
  MP_LAUNCH_CHECK();
  return MP_OK;
}
// 256x256 tiles (1 workgroup of 8 waves per CU: 2x the MFMA work per byte moved through the CU's vector-memory path and
// LDS) whenever the problem is wide enough; 128x128 tiles for the narrow bones-net layers.
static bool use_big_tile(const GemmB16Args& g) {
  if (g_force_small_tile) return false;
  return g.M >= 256 && g.N >= 256 && g.N % 256 == 0;
}
template <int TRA, int TRB, typename TC, int EPI, int SPLIT = 0>
static int launch_glds(const GemmB16Args& g, int splits, hipStream_t st) {
  if constexpr (TRA == 0 && EPI != EPI_SLAB) {
    const int wgs = persist_workgroups();
    // (a residual scale other than 1 - muP - is served by the tiled kernels only: the persistent residual epilogue sits at the 256-VGPR limit)
    if (wgs > 0 && splits == 1 && (g.rscale == 0.f || g.rscale == 1.0f) && use_big_tile(g) && g.K >= 2 * GBK && g.K % GBK == 0 && (long)cdiv(g.N, 256) * cdiv(g.M, 256) >= (g_persist_min_tiles > 0 ? (long)g_persist_min_tiles : 2L * wgs) &&
        256L * g.lda * 2 < (1L << 31) && 64L * g.ldb * 2 < (1L << 31) && 256L * g.ldb * 2 < (1L << 31))
      return launch_persist<TRB, TC, EPI, SPLIT>(g, wgs, st);
  }
  return use_big_tile(g) ? launch_glds_bt<TRA, TRB, TC, EPI, 256, SPLIT>(g, splits, st) : launch_glds_bt<TRA, TRB, TC, EPI, 128, SPLIT>(g, splits, st);
}

// dW += sum of the split-K slabs, db += sum of the bias slabs, ONE launch.  A block owns 64 float4 outputs; its four waves each sum
// a contiguous quarter of the slabs (eight independent loads in flight per lane), and wave 0 adds the four partial sums in fixed
// order: deterministic, and four times the loads in flight of a thread-per-output loop (the kernel is latency-bound: 1-3 MB of
// outputs, 21-64 slabs deep).
constexpr int RS_OUT = 64;
__global__ __launch_bounds__(256) void reduce_slabs_b16_kernel(const float* __restrict__ slabW, float* __restrict__ dW, long nW4,
                                                               const float* __restrict__ slabB, float* __restrict__ db, long nB4, int S, int SB,
                                                               const float* __restrict__ scale_p) {
  const float scale = scale_p != nullptr ? *scale_p : 1.0f;      // 1, or the inverse of the scale this backward's fp16 operands carried
  __shared__ float4 part[3][RS_OUT];
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  long i = (long)blockIdx.x * RS_OUT + lane;
  const float* slab = slabW;
  float* out = dW;
  long n4 = nW4;
  bool live = true;
  if (i >= nW4) {
    i -= nW4;
    live = i < nB4;
    slab = slabB; out = db; n4 = nB4; S = SB;
  }
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) {
    const int per = (S + 3) >> 2, k0 = q * per, k1 = min(S, k0 + per);
    const float4* p = reinterpret_cast<const float4*>(slab) + i;
    int k = k0;
    for (; k + 8 <= k1; k += 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(long)(k + u) * n4];
#pragma unroll
      for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; k < k1; ++k) {
      const float4 a = p[(long)k * n4];
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
  }
  if (q > 0) part[q - 1][lane] = s;
  __syncthreads();
  if (q == 0 && live) {
    float4 o = reinterpret_cast<float4*>(out)[i];
#pragma unroll
    for (int r = 0; r < 3; ++r) { s.x += part[r][lane].x; s.y += part[r][lane].y; s.z += part[r][lane].z; s.w += part[r][lane].w; }
    o.x += s.x * scale; o.y += s.y * scale; o.z += s.z * scale; o.w += s.w * scale;      // scale: 1, or what the fp16 operands carried
    reinterpret_cast<float4*>(out)[i] = o;
  }
}

template <typename TA, int TRA, typename TB, int TRB, typename TC, int EPI>
static int launch_b16(const GemmB16Args& g, int splits, hipStream_t st) {
  dim3 grid(cdiv(g.N, GBN), cdiv(g.M, GBM), splits);
  hipLaunchKernelGGL((gemm_bf16_kernel<TA, TRA, TB, TRB, TC, EPI>), grid, dim3(256), 0, st, g);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

// C[M,N] = A(i,r) B(r,j): a_f32/c_f32 select fp32 instead of bf16 storage; a_tr/b_tr select the "T" layouts.
int gemm_bf16(GemmB16Args g, int a_f32, int a_tr, int b_tr, int c_f32, int epi, hipStream_t st) {
  g.debug = 0;
#ifdef MP_GEMM_DIAG
  { static const int dbg = [] { const char* e = getenv("MANIPOSE_GEMM_DEBUG"); return e ? atoi(e) : 0; }(); g.debug = dbg; }   // timing ablations (1 no MFMA, 2 no DMA, 4 no epilogue, 8 no bias-gradient column sums in the weight-gradient kernel)
#endif
  MP_CHECK(g.M > 0 && g.N > 0 && g.K > 0, MP_ERR_ARG, "gemm_bf16: empty problem");
  MP_CHECK((a_tr ? g.M : g.K) % 8 == 0 && (b_tr ? g.N : g.K) % 8 == 0 && g.lda % 8 == 0 && g.ldb % 8 == 0 && g.N % 4 == 0 &&
               g.ldc % 4 == 0, MP_ERR_ARG,
           "gemm_bf16: contiguous operand dimensions and leading dimensions must be multiples of 8 (M=%d N=%d K=%d)", g.M, g.N, g.K);
  MP_CHECK(g.rstats == nullptr || (!a_f32 && epi == EPI_BIAS_RESID && g.rgamma && g.rbeta), MP_ERR_ARG,
           "gemm_bf16: a recomputed residual needs the direct-to-LDS residual variant and gamma / beta");
  g.k_per_split = ((g.K + GBK - 1) / GBK) * GBK;
  if (!a_f32 && !a_tr && !b_tr && !c_f32 && epi == EPI_BIAS) return launch_glds<0, 0, bf16, EPI_BIAS>(g, 1, st);
  if (!a_f32 && !a_tr && !b_tr && c_f32 && epi == EPI_BIAS) return launch_glds<0, 0, float, EPI_BIAS>(g, 1, st);
  if (!a_f32 && !a_tr && !b_tr && c_f32 && epi == EPI_BIAS_RESID) return launch_glds<0, 0, float, EPI_BIAS_RESID>(g, 1, st);
  if (!a_f32 && !a_tr && !b_tr && !c_f32 && epi == EPI_BIAS_GELU) return launch_glds<0, 0, bf16, EPI_BIAS_GELU>(g, 1, st);
  if (a_f32 && !a_tr && b_tr && !c_f32 && epi == EPI_DGELU) return launch_b16<float, 0, bf16, 1, bf16, EPI_DGELU>(g, 1, st);
  if (!a_f32 && !a_tr && b_tr && !c_f32 && epi == EPI_DGELU && g.f16) return launch_glds<0, 1, bf16, EPI_DGELU, 16>(g, 1, st);      // fp16 dY and weights
  if (!a_f32 && !a_tr && b_tr && !c_f32 && epi == EPI_DGELU) return launch_glds<0, 1, bf16, EPI_DGELU>(g, 1, st);
  if (!a_f32 && !a_tr && b_tr && !c_f32 && epi == EPI_BIAS && g.f16) {      // dgrad on fp16 operands (dY a scaled fp16 gradient, weights from the f16f8 shadow); bf16 out, which KEEPS dY's scale
    MP_CHECK(g.bias == nullptr, MP_ERR_ARG, "gemm_bf16: the fp16-operand dgrad takes no bias");
    return launch_glds<0, 1, bf16, EPI_BIAS, 16>(g, 1, st);
  }
  if (!a_f32 && !a_tr && b_tr && !c_f32 && epi == EPI_BIAS) return launch_glds<0, 1, bf16, EPI_BIAS>(g, 1, st);
  if (a_f32 && !a_tr && b_tr && !c_f32 && epi == EPI_BIAS) return launch_b16<float, 0, bf16, 1, bf16, EPI_BIAS>(g, 1, st);
  if (!a_f32 && !a_tr && b_tr && c_f32 && epi == EPI_BIAS) return launch_glds<0, 1, float, EPI_BIAS>(g, 1, st);
  if (a_f32 && !a_tr && b_tr && c_f32 && epi == EPI_BIAS) return launch_b16<float, 0, bf16, 1, float, EPI_BIAS>(g, 1, st);
  MP_CHECK(false, MP_ERR_ARG, "gemm_bf16: unsupported variant a_f32=%d a_tr=%d b_tr=%d c_f32=%d epi=%d", a_f32, a_tr, b_tr, c_f32, epi);
}

int gemm_bf16x3(GemmB16Args g, int c_f32, int epi, hipStream_t st) {
  g.debug = 0;
#ifdef MP_GEMM_DIAG
  { static const int dbg = [] { const char* e = getenv("MANIPOSE_GEMM_DEBUG"); return e ? (atoi(e) & 4) : 0; }(); g.debug = dbg; }   // timing ablation: 4 = no epilogue (the only bit the split loop looks at)
#endif
  MP_CHECK(g.M > 0 && g.N > 0 && g.K > 0, MP_ERR_ARG, "gemm_bf16x3: empty problem");
  MP_CHECK(g.K % 8 == 0 && g.lda % 8 == 0 && g.ldb % 8 == 0 && g.N % 4 == 0 && g.ldc % 4 == 0, MP_ERR_ARG,
           "gemm_bf16x3: K and the leading dimensions must be multiples of 8, N of 4 (M=%d N=%d K=%d)", g.M, g.N, g.K);
  MP_CHECK(g.A_lo && g.B_lo && (c_f32 || g.C_lo), MP_ERR_ARG, "gemm_bf16x3: lo plane missing");
  g.k_per_split = ((g.K + GBK - 1) / GBK) * GBK;
#ifdef MP_GEMM_DIAG
  {
    static const int abl = [] { const char* e = getenv("MANIPOSE_GEMM_ABL"); return e ? atoi(e) : 0; }();     // timing ablation of the fragment reads (extra kernel instantiations)
    if (abl && !c_f32 && epi == EPI_BIAS) {
      const int wgs = persist_workgroups();
      if (abl == 1) return launch_persist<0, bf16p, EPI_BIAS, 3>(g, wgs, st);
      if (abl == 2) return launch_persist<0, bf16p, EPI_BIAS, 5>(g, wgs, st);
      return launch_persist<0, bf16p, EPI_BIAS, 7>(g, wgs, st);
    }
  }
#endif
  if (!c_f32 && epi == EPI_BIAS) return launch_glds<0, 0, bf16p, EPI_BIAS, 1>(g, 1, st);
  if (!c_f32 && epi == EPI_BIAS_GELU) return launch_glds<0, 0, bf16p, EPI_BIAS_GELU, 1>(g, 1, st);
  if (c_f32 && epi == EPI_BIAS) return launch_glds<0, 0, float, EPI_BIAS, 1>(g, 1, st);
  if (c_f32 && epi == EPI_BIAS_RESID) return launch_glds<0, 0, float, EPI_BIAS_RESID, 1>(g, 1, st);
  MP_CHECK(false, MP_ERR_ARG, "gemm_bf16x3: unsupported variant c_f32=%d epi=%d", c_f32, epi);
}

// y = x W^T + b with x and W carried as fp16 hi planes + 8-bit correction planes (see mma_stage_f8): persistent kernel only
// (N % 256 == 0, K % 64 == 0, K >= 128).  A_lo / B_lo are the correction planes.  Outputs: c_f32 with EPI_BIAS: fp32; otherwise the planar
// bf16 hi / lo pair of the bf16x3 kernels (C, C_lo), EPI_BIAS or EPI_BIAS_GELU (+ Z = gelu' as plain bf16) - what the attention kernels and
// the fc2 GEMM of the engine read.
int gemm_f16f8(GemmB16Args g, int c_f32, int epi, hipStream_t st) {
  g.debug = 0;
  MP_CHECK(g.M > 0 && g.N > 0 && g.N % 256 == 0 && g.K >= 2 * GBK && g.K % GBK == 0 && g.lda % 8 == 0 && g.ldb % 8 == 0 && g.ldc % 4 == 0, MP_ERR_ARG,
           "gemm_f16f8: N must be a multiple of 256, K of 64 (M=%d N=%d K=%d)", g.M, g.N, g.K);
  MP_CHECK(g.A_lo && g.B_lo, MP_ERR_ARG, "gemm_f16f8: correction plane missing");
  MP_CHECK(256L * g.lda * 2 < (1L << 31) && 256L * g.ldb * 2 < (1L << 31), MP_ERR_ARG, "gemm_f16f8: leading dimension too large");
  g.k_per_split = g.K;
  int dev = 0, cus = 0;
  MP_HIP(hipGetDevice(&dev));
  MP_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const int wgs = max(8, (cus / 8) * 8);
  if (c_f32 && epi == EPI_BIAS) return launch_persist<0, float, EPI_BIAS, 8>(g, wgs, st);
  if (c_f32 && epi == EPI_BIAS_RESID) {
    MP_CHECK(g.R != nullptr && (g.rscale == 0.f || g.rscale == 1.0f), MP_ERR_ARG, "gemm_f16f8: residual epilogue needs R and a residual scale of 1");
    return launch_persist<0, float, EPI_BIAS_RESID, 8>(g, wgs, st);
  }
  MP_CHECK(!c_f32 && g.C_lo != nullptr, MP_ERR_ARG, "gemm_f16f8: planar output without its lo plane");
  if (epi == EPI_BIAS) return launch_persist<0, bf16p, EPI_BIAS, 8>(g, wgs, st);
  if (epi == EPI_BIAS_GELU && g.out_f16f8) return launch_persist<0, f16f8, EPI_BIAS_GELU, 8>(g, wgs, st);      // C / C_lo = fp16 + correction planes (the fc2 GEMM's f16f8 input)
  if (epi == EPI_BIAS_GELU) return launch_persist<0, bf16p, EPI_BIAS_GELU, 8>(g, wgs, st);      // Z may be null (inference: gelu' is not kept)
  MP_CHECK(false, MP_ERR_ARG, "gemm_f16f8: unsupported variant c_f32=%d epi=%d", c_f32, epi);
}

static void wgrad_split_b16(int Mtok, int Nout, int Kin, int bt, int& splits, int& kper) {
  const int tiles = cdiv(Nout, bt) * cdiv(Kin, bt);
  // 256^2 tiles run one workgroup per CU: fill the 256 CUs exactly once (a 257th workgroup would double the kernel time);
  // 128^2 tiles run 2-4 per CU.  Fewer splits also means fewer fp32 slabs to write and reduce.
  // 128-wide tiles (the bones net): up to 128 splits - with 64 its one to three output tiles gave 64-192 workgroups for 512 slots (round 4: the
  // weight-gradient class 22.54 -> 22.00 ms with every kernel on one queue, 256 splits the same; the three-queue step does not move, the bones
  // net's backward runs beside the main queue: profiles/r04_probes/ab_step2.log)
  splits = (bt == 256) ? max(1, min(64, 256 / tiles)) : max(1, min(128, (1024 + tiles - 1) / tiles));
  kper = ((cdiv(Mtok, splits) + GBK - 1) / GBK) * GBK;
  splits = cdiv(Mtok, kper);
}

// dW[N',K'] += dY[Mtok,N']^T X[Mtok,K'] (X bf16; dY bf16 or fp32) ; db += colsum(dY)
int wgrad_bf16(const void* dY, int dy_f32, long lddy, const bf16* X, long ldx, int Mtok, int Nout, int Kin, float* dW, float* db,
               float* slab, long slab_floats, hipStream_t st, int f16, const float* oscale, int x_f16) {
  MP_CHECK(Mtok > 0 && Nout % 8 == 0 && Kin % 8 == 0, MP_ERR_ARG, "wgrad_bf16: bad dims %d %d %d", Mtok, Nout, Kin);
  GemmB16Args g = {};
  g.A = dY; g.lda = lddy; g.B = X; g.ldb = ldx;
  g.M = Nout; g.N = Kin; g.K = Mtok;
  int splits, kper;
  wgrad_split_b16(Mtok, Nout, Kin, (!dy_f32 && use_big_tile(g)) ? 256 : 128, splits, kper);
  const long per = (long)Nout * Kin + Nout;
  MP_CHECK(slab_floats >= per * splits, MP_ERR_ARG, "wgrad_bf16: slab too small (%ld < %ld)", slab_floats, per * splits);
  g.C = slab; g.ldc = Kin;
  g.bias_slab = (db != nullptr) ? slab + (long)splits * Nout * Kin : nullptr;
  // 256-wide tiles: spread the bias-gradient sums over the column tiles of a row tile when the slab has room for their extra rows
  int bparts = 1;
  if (db != nullptr && !dy_f32 && use_big_tile(g)) {
    const int tn = cdiv(Kin, 256);
    if (tn > 1 && slab_floats >= ((long)Nout * Kin + (long)Nout * tn) * splits) bparts = tn;
  }
  g.bias_parts = bparts;
#ifdef MP_GEMM_DIAG
  { static const int dbg = [] { const char* e = getenv("MANIPOSE_GEMM_DEBUG"); return e ? atoi(e) : 0; }(); g.debug = dbg; }   // timing ablations: 1 no MFMA, 2 no DMA, 8 no bias column sums
#endif
  g.k_per_split = kper;
  MP_CHECK(!f16 || !dy_f32, MP_ERR_ARG, "wgrad_bf16: fp16 operands with an fp32 dY");
  MP_CHECK(!x_f16 || (!f16 && !dy_f32), MP_ERR_ARG, "wgrad_bf16: an fp16 X next to a bf16 dY is converted per fragment (x_f16): not with fp16 operands / an fp32 dY");
  int rc = dy_f32 ? launch_b16<float, 1, bf16, 1, float, EPI_SLAB>(g, splits, st)
                  : (f16 ? launch_glds<1, 1, float, EPI_SLAB, 16>(g, splits, st)
                         : (x_f16 ? launch_glds<1, 1, float, EPI_SLAB, 32>(g, splits, st) : launch_glds<1, 1, float, EPI_SLAB>(g, splits, st)));
  if (rc) return rc;
  const long nW4 = (long)Nout * Kin / 4, nB4 = db != nullptr ? Nout / 4 : 0;        // Nout, Kin are multiples of 8
  hipLaunchKernelGGL(reduce_slabs_b16_kernel, dim3(cdiv(nW4, (long)RS_OUT) + cdiv(nB4, (long)RS_OUT)), dim3(256), 0, st, slab, dW, nW4, g.bias_slab, db, nB4, splits, splits * bparts,
                     f16 ? oscale : (const float*)nullptr);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

// fp32 -> bf16 shadow copy of the flat parameter buffer (refreshed at the start of every bf16-mode forward)
__global__ void cast_bf16_kernel(const float* __restrict__ src, bf16* __restrict__ dst, long n4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n4) st4(dst + 4 * i, ld4(src + 4 * i));
}
// the same as planar hi/lo shadows (split precision)
__global__ void cast_bf16x2_kernel(const float* __restrict__ src, bf16p* __restrict__ hi, long lo_off, long n4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n4) st4(hi + 4 * i, ld4(src + 4 * i), lo_off);
}
// fp32 -> "f16f8" planes (common.h): flat, elements 4 q .. 4 q + 3 of the source become bytes 8 q .. 8 q + 7 of the correction plane
__global__ void cast_f16f8_kernel(const float* __restrict__ src, f16f8* __restrict__ hi16, char* __restrict__ cat8, long n4, int weight) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const long e = 4 * i;
  st4_f16f8(hi16 + e, cat8 + 2 * e, ld4(src + e), weight != 0);
}
int cast_to_f16f8(const float* src, void* hi16, void* cat8, long n, int weight, hipStream_t st) {
  MP_CHECK(n % 64 == 0, MP_ERR_ARG, "cast_to_f16f8: n %% 64");
  hipLaunchKernelGGL(cast_f16f8_kernel, dim3(cdiv(n / 4, 256)), dim3(256), 0, st, src, reinterpret_cast<f16f8*>(hi16), reinterpret_cast<char*>(cat8), n / 4, weight);
  MP_LAUNCH_CHECK();
  return MP_OK;
}
int cast_to_bf16x2(const float* src, bf16* hi, bf16* lo, long n, hipStream_t st) {
  MP_CHECK(n % 4 == 0, MP_ERR_ARG, "cast_to_bf16x2: n %% 4");
  hipLaunchKernelGGL(cast_bf16x2_kernel, dim3(cdiv(n / 4, 256)), dim3(256), 0, st, src, reinterpret_cast<bf16p*>(hi), (long)(lo - hi), n / 4);
  MP_LAUNCH_CHECK();
  return MP_OK;
}
int cast_to_bf16(const float* src, bf16* dst, long n, hipStream_t st) {
  MP_CHECK(n % 4 == 0, MP_ERR_ARG, "cast_to_bf16: n %% 4");
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(cdiv(n / 4, 256)), dim3(256), 0, st, src, dst, n / 4);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

}  // namespace mp
