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
#include "common.h"
#include "kernels.h"

namespace mp {

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
        st4(Z + o, v);
        v = make_float4(gelu_fast(v.x), gelu_fast(v.y), gelu_fast(v.z), gelu_fast(v.w));
      } else if (EPI == EPI_BIAS_RESID) {
        const float4 r = ld4(g.R + o);
        v = make_float4(r.x + dscale * v.x, r.y + dscale * v.y, r.z + dscale * v.z, r.w + dscale * v.w);
      } else if (EPI == EPI_DGELU) {
        const float4 z = ld4(Z + o);
        v = make_float4(v.x * gelu_grad_fast(z.x), v.y * gelu_grad_fast(z.y), v.z * gelu_grad_fast(z.z), v.w * gelu_grad_fast(z.w));
      }
      st4(C + o, v);
    }
  }
}

__global__ void reduce_slabs_b16_kernel(const float* __restrict__ slab, float* __restrict__ out, long n, int S) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < S; ++k) s += slab[(long)k * n + i];
  out[i] += s;
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
  MP_CHECK(g.M > 0 && g.N > 0 && g.K > 0, MP_ERR_ARG, "gemm_bf16: empty problem");
  MP_CHECK((a_tr ? g.M : g.K) % 8 == 0 && (b_tr ? g.N : g.K) % 8 == 0 && g.lda % 8 == 0 && g.ldb % 8 == 0 && g.N % 4 == 0 &&
               g.ldc % 4 == 0, MP_ERR_ARG,
           "gemm_bf16: contiguous operand dimensions and leading dimensions must be multiples of 8 (M=%d N=%d K=%d)", g.M, g.N, g.K);
  g.k_per_split = ((g.K + GBK - 1) / GBK) * GBK;
  if (!a_f32 && !a_tr && !b_tr && !c_f32 && epi == EPI_BIAS) return launch_b16<bf16, 0, bf16, 0, bf16, EPI_BIAS>(g, 1, st);
  if (!a_f32 && !a_tr && !b_tr && c_f32 && epi == EPI_BIAS) return launch_b16<bf16, 0, bf16, 0, float, EPI_BIAS>(g, 1, st);
  if (!a_f32 && !a_tr && !b_tr && c_f32 && epi == EPI_BIAS_RESID) return launch_b16<bf16, 0, bf16, 0, float, EPI_BIAS_RESID>(g, 1, st);
  if (!a_f32 && !a_tr && !b_tr && !c_f32 && epi == EPI_BIAS_GELU) return launch_b16<bf16, 0, bf16, 0, bf16, EPI_BIAS_GELU>(g, 1, st);
  if (a_f32 && !a_tr && b_tr && !c_f32 && epi == EPI_DGELU) return launch_b16<float, 0, bf16, 1, bf16, EPI_DGELU>(g, 1, st);
  if (!a_f32 && !a_tr && b_tr && !c_f32 && epi == EPI_DGELU) return launch_b16<bf16, 0, bf16, 1, bf16, EPI_DGELU>(g, 1, st);
  if (!a_f32 && !a_tr && b_tr && !c_f32 && epi == EPI_BIAS) return launch_b16<bf16, 0, bf16, 1, bf16, EPI_BIAS>(g, 1, st);
  if (a_f32 && !a_tr && b_tr && !c_f32 && epi == EPI_BIAS) return launch_b16<float, 0, bf16, 1, bf16, EPI_BIAS>(g, 1, st);
  if (!a_f32 && !a_tr && b_tr && c_f32 && epi == EPI_BIAS) return launch_b16<bf16, 0, bf16, 1, float, EPI_BIAS>(g, 1, st);
  if (a_f32 && !a_tr && b_tr && c_f32 && epi == EPI_BIAS) return launch_b16<float, 0, bf16, 1, float, EPI_BIAS>(g, 1, st);
  MP_CHECK(false, MP_ERR_ARG, "gemm_bf16: unsupported variant a_f32=%d a_tr=%d b_tr=%d c_f32=%d epi=%d", a_f32, a_tr, b_tr, c_f32, epi);
}

static void wgrad_split_b16(int Mtok, int Nout, int Kin, int& splits, int& kper) {
  const int tiles = cdiv(Nout, GBM) * cdiv(Kin, GBN);
  splits = max(1, min(64, (1024 + tiles - 1) / tiles));
  kper = ((cdiv(Mtok, splits) + GBK - 1) / GBK) * GBK;
  splits = cdiv(Mtok, kper);
}

// dW[N',K'] += dY[Mtok,N']^T X[Mtok,K'] (X bf16; dY bf16 or fp32) ; db += colsum(dY)
int wgrad_bf16(const void* dY, int dy_f32, long lddy, const bf16* X, long ldx, int Mtok, int Nout, int Kin, float* dW, float* db,
               float* slab, long slab_floats, hipStream_t st) {
  MP_CHECK(Mtok > 0 && Nout % 8 == 0 && Kin % 8 == 0, MP_ERR_ARG, "wgrad_bf16: bad dims %d %d %d", Mtok, Nout, Kin);
  int splits, kper;
  wgrad_split_b16(Mtok, Nout, Kin, splits, kper);
  const long per = (long)Nout * Kin + Nout;
  MP_CHECK(slab_floats >= per * splits, MP_ERR_ARG, "wgrad_bf16: slab too small (%ld < %ld)", slab_floats, per * splits);
  GemmB16Args g = {};
  g.A = dY; g.lda = lddy; g.B = X; g.ldb = ldx;
  g.M = Nout; g.N = Kin; g.K = Mtok;
  g.C = slab; g.ldc = Kin;
  g.bias_slab = (db != nullptr) ? slab + (long)splits * Nout * Kin : nullptr;
  g.k_per_split = kper;
  int rc = dy_f32 ? launch_b16<float, 1, bf16, 1, float, EPI_SLAB>(g, splits, st)
                  : launch_b16<bf16, 1, bf16, 1, float, EPI_SLAB>(g, splits, st);
  if (rc) return rc;
  const long n = (long)Nout * Kin;
  hipLaunchKernelGGL(reduce_slabs_b16_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, slab, dW, n, splits);
  MP_LAUNCH_CHECK();
  if (db != nullptr) {
    hipLaunchKernelGGL(reduce_slabs_b16_kernel, dim3(cdiv(Nout, 256)), dim3(256), 0, st, g.bias_slab, db, (long)Nout, splits);
    MP_LAUNCH_CHECK();
  }
  return MP_OK;
}

// fp32 -> bf16 shadow copy of the flat parameter buffer (refreshed at the start of every bf16-mode forward)
__global__ void cast_bf16_kernel(const float* __restrict__ src, bf16* __restrict__ dst, long n4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n4) st4(dst + 4 * i, ld4(src + 4 * i));
}
int cast_to_bf16(const float* src, bf16* dst, long n, hipStream_t st) {
  MP_CHECK(n % 4 == 0, MP_ERR_ARG, "cast_to_bf16: n %% 4");
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(cdiv(n / 4, 256)), dim3(256), 0, st, src, dst, n / 4);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

}  // namespace mp
