// Output heads on the fp32 matrix cores (v_mfma_f32_16x16x4_f32).
//
// K heads of LayerNorm(C, eps 1e-5) -> Linear(C, O) (MCLHead, rmcl_manifold_mix_ste.py:291-298; MixSTE.head, mix_ste.py:123-126) are ONE
// product once the per-head LayerNorm affine is folded into the weights:  with  G[n] = W_k[o] * gamma_k  (n = k O + o),
// s[n] = sum_c G[n][c],  c0[n] = W_k[o] . beta_k + b_k[o]  and the row statistics (mean, rstd) of x,
//     y[m][n] = xhat[m] . G[n] + c0[n],      xhat = (x - mean) rstd.
// The row kernels of heads.hip spend their time in 35 wave reductions per token; here a wave owns 16 tokens, keeps its quarter of each
// row in registers (read once: exact two-pass statistics), and the 16 x 48 outputs are 4 x 3 MFMAs per 16 channels against G staged in
// LDS.  fp32 operands and accumulation throughout: the heads feed the 6-D -> SO(3) decoder directly.
#include "common.h"
#include "kernels.h"

namespace mp {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int HM_NPAD = 48;                 // K * O <= 48 output columns (three 16-wide tiles)

// G [48][C], c0 [48] from the K heads' parameters (rows >= K O are zero)
__global__ void heads_fold_kernel(HeadParams p, int K, int O, int C, float* __restrict__ G, float* __restrict__ c0) {
  const int n = blockIdx.x;                  // one block per output column
  __shared__ float red[256];
  float cb = 0.f;
  if (n < K * O) {
    const int k = n / O, o = n - k * O;
    for (int c = threadIdx.x; c < C; c += 256) {
      const float w = p.W[k][(long)o * C + c], g = w * p.gamma[k][c];
      G[(long)n * C + c] = g;
      cb += w * p.beta[k][c];
    }
  } else {
    for (int c = threadIdx.x; c < C; c += 256) G[(long)n * C + c] = 0.f;
  }
  red[threadIdx.x] = cb;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) {
    if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
    __syncthreads();
  }
  if (threadIdx.x == 0) c0[n] = (n < K * O) ? red[0] + p.b[n / O][n % O] : 0.f;
}

// NS = C / 16 channel blocks; NT = 16-wide output tiles actually needed (ceil(K O / 16))
constexpr int HM_FWD_THREADS = 512;       // 8 waves share the 96 KB of folded weights: two per SIMD, one loads while the other multiplies
template <int NS, int NT>
__global__ __launch_bounds__(HM_FWD_THREADS) void heads_fwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ G,
                                                              const float* __restrict__ c0, int K, int O, float* __restrict__ out,
                                                              float* __restrict__ stats, int M) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int C = NS * 16, PITCH = C + 4;          // + 16 B: the 16 rows a fragment read touches fall into different banks
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
  for (int i = tid; i < NT * 16 * (C / 4); i += HM_FWD_THREADS) {
    const int r = i / (C / 4), c4 = i - r * (C / 4);
    *reinterpret_cast<float4*>(lds + r * PITCH + 4 * c4) = ld4(G + (long)r * C + 4 * c4);
  }
  __syncthreads();
  const int wave = (blockIdx.x * HM_FWD_THREADS + tid) >> 6, nwaves = (gridDim.x * HM_FWD_THREADS) >> 6;
  const int NO = K * O;
  for (int m0 = wave * 16; m0 < M; m0 += nwaves * 16) {
    const int row = min(m0 + l15, M - 1);
    const float* xr = x + (long)row * C + 4 * g;
    float4 a[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) a[s] = ld4(xr + 16 * s);
    __builtin_amdgcn_sched_barrier(0);                  // all loads of the tile in flight before anything waits (the scheduler otherwise
                                                        // serialises them behind the MFMAs to save registers)
    // exact two-pass row statistics (the row is in the registers of the four lanes l15, l15 + 16, + 32, + 48)
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) sum += (a[s].x + a[s].y) + (a[s].z + a[s].w);
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const float d0 = a[s].x - mean, d1 = a[s].y - mean, d2 = a[s].z - mean, d3 = a[s].w - mean;
      sq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    sq += __shfl_xor(sq, 16, 64);
    sq += __shfl_xor(sq, 32, 64);
    const float rstd = 1.0f / sqrtf(sq / (float)C + 1e-5f);
    if (g == 0 && m0 + l15 < M) {
      stats[2 * (long)row] = mean;
      stats[2 * (long)row + 1] = rstd;
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      a[s].x = (a[s].x - mean) * rstd; a[s].y = (a[s].y - mean) * rstd; a[s].z = (a[s].z - mean) * rstd; a[s].w = (a[s].w - mean) * rstd;
    }
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float4 b = *reinterpret_cast<const float4*>(lds + (16 * t + l15) * PITCH + 16 * s + 4 * g);
        // the four MFMAs of a 16-channel block: lane group g carries channels 16 s + 4 g + r in MFMA r (any split of the reduction index
        // works as long as A and B agree); operands: A[i = token l15][k = g], B[k = g][j = output column l15]
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s].x, b.x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s].y, b.y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s].z, b.z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s].w, b.w, acc[t], 0, 0, 0);
      }
    }
    // acc[t][r] = xhat[m0 + 4 g + r] . G[16 t + l15]
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = 16 * t + l15;
      const float cn = c0[n];
      const int k = n / O, o = n - k * O;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + 4 * g + r;
        if (n < NO && m < M) out[((long)k * M + m) * O + o] = acc[t][r] + cn;
      }
    }
  }
}

static int hm_grid(int M) {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  return max(1, min(cdiv(M, 64), cus));
}

bool heads_mfma_supported(int K, int O, int C) { return K >= 1 && O >= 1 && O <= 8 && K * O <= HM_NPAD && (C == 512 || C == 256 || C == 128); }
// mp_set_option("heads_mfma"): 0 = row kernels everywhere, 1 (default) = matrix cores wherever covered (a 16-wide tile for the bones head's single
// output wastes most of the MFMA, but the kernel is bound by reading x either way), 2 = matrix cores only from 16 outputs up
static int g_heads_mfma = 1;
void heads_mfma_mode(int mode) { g_heads_mfma = mode; }
bool heads_use_mfma(int K, int O, int C) {
  const int on = g_heads_mfma;
  return on && (on == 1 || K * O >= 16) && heads_mfma_supported(K, O, C);
}

// fold: 48 C + 48 floats, filled here with the folded weights [48][C] | c0 [48] and read again by heads_bwd_mfma
int heads_fwd_mfma(const float* x, const HeadParams& p, int K, int O, float* out, float* stats, int M, int C, float* fold, hipStream_t st) {
  MP_CHECK(heads_mfma_supported(K, O, C), MP_ERR_ARG, "heads_fwd_mfma: K=%d O=%d C=%d unsupported", K, O, C);
  float* G = fold;
  float* c0 = fold + (long)HM_NPAD * C;
  hipLaunchKernelGGL(heads_fold_kernel, dim3(HM_NPAD), dim3(256), 0, st, p, K, O, C, G, c0);
  MP_LAUNCH_CHECK();
  const int NT = cdiv(K * O, 16);
  const size_t lds = (size_t)NT * 16 * (C + 4) * sizeof(float);
#define MP_HM_FWD(NS_, NT_)                                                                                                                      \
  do {                                                                                                                                           \
    static bool attr_set = false;                                                                                                                \
    if (!attr_set) {                                                                                                                             \
      MP_HIP(hipFuncSetAttribute((const void*)heads_fwd_mfma_kernel<NS_, NT_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));       \
      attr_set = true;                                                                                                                           \
    }                                                                                                                                            \
    hipLaunchKernelGGL((heads_fwd_mfma_kernel<NS_, NT_>), dim3(hm_grid(M / 2 + 1)), dim3(HM_FWD_THREADS), lds, st, x, G, c0, K, O, out, stats, M);              \
  } while (0)
  if (C == 512)      { if (NT == 3) MP_HM_FWD(32, 3); else if (NT == 2) MP_HM_FWD(32, 2); else MP_HM_FWD(32, 1); }
  else if (C == 256) { if (NT == 3) MP_HM_FWD(16, 3); else if (NT == 2) MP_HM_FWD(16, 2); else MP_HM_FWD(16, 1); }
  else               { if (NT == 3) MP_HM_FWD(8, 3); else if (NT == 2) MP_HM_FWD(8, 2); else MP_HM_FWD(8, 1); }
#undef MP_HM_FWD
  MP_LAUNCH_CHECK();
  return MP_OK;
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
// dx = LN'(d),  d[m][c] = sum_n dY[m][n] G[n][c]:  computed transposed (tile rows = channels, columns = the wave's 16 tokens), so a
// lane ends up with d at the (token l15, channels 16 t + 4 g ..+3) positions of one float4 of x.  The two row sums of the LayerNorm
// backward do not need d:  sum_c d[m][c] = sum_n dY[m][n] sG[n]  (sG = row sums of G)  and  sum_c d[m][c] xhat[m][c] =
// sum_n dY[m][n] (y[m][n] - c0[n])  because the forward output IS xhat . G + c0.  Both come from the 35 head outputs and their
// gradients, so a channel tile is finished (12 MFMAs, one float4 of x, one float4 of dx) without keeping the row in registers, and a
// block runs 16 waves on one copy of the weights.  MFMA step q of lane group g reduces over output column n = 4 q + g (so ceil(K O / 4)
// steps cover the columns there are); G^T sits in LDS as [c][52] with column n at 12 (n & 3) + (n >> 2): a lane's A values for one
// channel tile are up to three aligned float4 reads (NQ4 of them).
constexpr int HM_GP = 52;
constexpr int HM_DX_THREADS = 1024;

__device__ __forceinline__ int hm_pos(int n) { return 12 * (n & 3) + (n >> 2); }

template <int NS, int NQ4>
__global__ __launch_bounds__(HM_DX_THREADS) void heads_bwd_dx_mfma_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                                           const float* __restrict__ G, int K, int O, const float* __restrict__ dout,
                                                                           const float* __restrict__ yout, float* __restrict__ dx, int M) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int C = NS * 16, UT = 8;
  float* sGs = lds + C * HM_GP;                          // [48] row sums of G, then [48] c0
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
  const int NO = K * O;
  for (int i = tid; i < HM_NPAD * (C / 4); i += HM_DX_THREADS) {   // transpose the folded weights (heads_fold_kernel, this step's forward) into LDS
    const int n = i / (C / 4), c = 4 * (i - n * (C / 4));
    const float4 v = ld4(G + (long)n * C + c);
    const int at = hm_pos(n);
    lds[c * HM_GP + at] = v.x; lds[(c + 1) * HM_GP + at] = v.y; lds[(c + 2) * HM_GP + at] = v.z; lds[(c + 3) * HM_GP + at] = v.w;
  }
  __syncthreads();
  if (tid < HM_NPAD * 16) {                               // 16 lanes per row sum
    const int n = tid >> 4, part = tid & 15;
    float a = 0.f;
    for (int c = part; c < C; c += 16) a += lds[c * HM_GP + hm_pos(n)];
    a += __shfl_xor(a, 8, 64); a += __shfl_xor(a, 4, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 1, 64);
    if (part == 0) { sGs[n] = a; sGs[HM_NPAD + n] = G[(long)HM_NPAD * C + n]; }
  }
  __syncthreads();
  constexpr int NQ = 4 * NQ4;
  int noff[NQ];                                          // element offset of (k, o) for n = 4 q + g in the [K][M][O] head outputs; -1 = padding column
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int n = 4 * q + g;
    const int k = n / O, o = n - k * O;
    noff[q] = n < NO ? k * M * O + o : -1;
  }
  const int wave = (blockIdx.x * HM_DX_THREADS + tid) >> 6, nwaves = (gridDim.x * HM_DX_THREADS) >> 6;
  for (int m0 = wave * 16; m0 < M; m0 += nwaves * 16) {
    const int tok = min(m0 + l15, M - 1);
    const bool live = m0 + l15 < M;
    float bq[NQ];
    float s1 = 0.f, s2 = 0.f;
    {
      float yq[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {                     // masked by a multiply: a select lets the compiler sink the load into a branch
        const long at = (long)max(noff[q], 0) + (long)tok * O;
        bq[q] = dout[at] * ((noff[q] >= 0 && live) ? 1.f : 0.f);
        yq[q] = yout[at];
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        s1 += bq[q] * sGs[4 * q + g];
        s2 += bq[q] * (yq[q] - sGs[HM_NPAD + 4 * q + g]);
      }
    }
    const float2 ms = *reinterpret_cast<const float2*>(stats + 2 * (long)tok);
    s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
    s1 /= (float)C; s2 /= (float)C;
    const float* xr = x + (long)tok * C + 4 * g;
    float* dr = dx + (long)tok * C + 4 * g;
#pragma unroll 1
    for (int t0 = 0; t0 < NS; t0 += UT) {
      float4 a[UT];
#pragma unroll
      for (int u = 0; u < UT; ++u) a[u] = ld4(xr + 16 * (t0 + u));
      __builtin_amdgcn_sched_barrier(0);                // the chunk's loads are in flight before the first wait
#pragma unroll
      for (int u = 0; u < UT; u += 2) {
        // two channel tiles x two accumulators each: four independent MFMA chains.  A[i = channel l15][k = g], B[k = g][j = token l15]
        f32x4 accA[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, accB[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        const float* gA = lds + (16 * (t0 + u) + l15) * HM_GP + 12 * g;
        const float* gB = gA + 16 * HM_GP;
#pragma unroll
        for (int q4 = 0; q4 < NQ4; ++q4) {
          const float4 wA = *reinterpret_cast<const float4*>(gA + 4 * q4), wB = *reinterpret_cast<const float4*>(gB + 4 * q4);
          accA[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wA.x, bq[4 * q4 + 0], accA[0], 0, 0, 0);
          accB[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wB.x, bq[4 * q4 + 0], accB[0], 0, 0, 0);
          accA[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wA.y, bq[4 * q4 + 1], accA[1], 0, 0, 0);
          accB[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wB.y, bq[4 * q4 + 1], accB[1], 0, 0, 0);
          accA[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wA.z, bq[4 * q4 + 2], accA[0], 0, 0, 0);
          accB[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wB.z, bq[4 * q4 + 2], accB[0], 0, 0, 0);
          accA[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wA.w, bq[4 * q4 + 3], accA[1], 0, 0, 0);
          accB[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wB.w, bq[4 * q4 + 3], accB[1], 0, 0, 0);
        }
        // acc[r] = d[token l15][channel 16 t + 4 g + r]
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x4 d = h ? accB[0] + accB[1] : accA[0] + accA[1];
          const float4 xv = a[u + h];
          const float x0 = (xv.x - ms.x) * ms.y, x1 = (xv.y - ms.x) * ms.y, x2 = (xv.z - ms.x) * ms.y, x3 = (xv.w - ms.x) * ms.y;
          if (live)
            st4(dr + 16 * (t0 + u + h),
                make_float4(ms.y * (d[0] - s1 - x0 * s2), ms.y * (d[1] - s1 - x1 * s2), ms.y * (d[2] - s1 - x2 * s2), ms.y * (d[3] - s1 - x3 * s2)));
        }
      }
    }
  }
}

// parameter gradients:  dG'[n][c] = sum_m dY[m][n] xhat[m][c]  and  db[n] = sum_m dY[m][n]  are all the token sums there are; dW, dgamma,
// dbeta follow from them per channel (heads_bwd_fin_kernel).  Wave w of a block owns channels [64 w, 64 w + 64) (column l15 of channel
// tile r' = channel 64 w + 4 l15 + r': one float4 of x per lane and MFMA step), every wave of the block walks the same tokens, four per
// step (k index = lane group).  NU = 16-column tiles that hold the K O outputs.  Partial layout per block: [48][C] | [48] (rows >= 16 NU
// are neither written nor read).
template <int UNR, int NU>
__global__ __launch_bounds__(512) void heads_bwd_param_mfma_kernel(const float* __restrict__ x, const float* __restrict__ stats, int K, int O,
                                                                    const float* __restrict__ dout, float* __restrict__ partial, int M, int C) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int NO = K * O;
  int noff[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int n = 16 * u + l15;
    const int k = n / O, o = n - k * O;
    noff[u] = n < NO ? k * M * O + o : -1;
  }
  f32x4 acc[NU][4];
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[u][r] = f32x4{0.f, 0.f, 0.f, 0.f};
  float db[NU] = {};
  const int cbase = 64 * w + 4 * l15;
  for (int m0 = blockIdx.x * 4 * UNR; m0 < M; m0 += gridDim.x * 4 * UNR) {
    float4 xv[UNR];
    float2 ms[UNR];
    float dy[UNR][NU];
#pragma unroll
    for (int q = 0; q < UNR; ++q) {
      const int m = m0 + 4 * q + g;
      const int tok = min(m, M - 1);
      xv[q] = ld4(x + (long)tok * C + cbase);
      ms[q] = *reinterpret_cast<const float2*>(stats + 2 * (long)tok);
#pragma unroll
      for (int u = 0; u < NU; ++u)                        // masked by a multiply: a select lets the compiler sink the load into a branch
        dy[q][u] = dout[(long)max(noff[u], 0) + (long)tok * O] * ((noff[u] >= 0 && m < M) ? 1.f : 0.f);
    }
    __builtin_amdgcn_sched_barrier(0);                  // every load of the batch is issued before the first wait
#pragma unroll
    for (int q = 0; q < UNR; ++q) {
      const float x0 = (xv[q].x - ms[q].x) * ms[q].y, x1 = (xv[q].y - ms[q].x) * ms[q].y, x2 = (xv[q].z - ms[q].x) * ms[q].y,
                  x3 = (xv[q].w - ms[q].x) * ms[q].y;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        // A[i = n l15][k = token g], B[k = token g][j = l15 -> channel 64 w + 4 l15 + r']
        acc[u][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(dy[q][u], x0, acc[u][0], 0, 0, 0);
        acc[u][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(dy[q][u], x1, acc[u][1], 0, 0, 0);
        acc[u][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(dy[q][u], x2, acc[u][2], 0, 0, 0);
        acc[u][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(dy[q][u], x3, acc[u][3], 0, 0, 0);
        db[u] += dy[q][u];
      }
    }
  }
  // acc[u][r'][r] = dG'[n = 16 u + 4 g + r][channel 64 w + 4 l15 + r']
  float* pr = partial + (long)blockIdx.x * (HM_NPAD * (long)C + HM_NPAD);
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      st4(pr + (long)(16 * u + 4 * g + r) * C + cbase, make_float4(acc[u][0][r], acc[u][1][r], acc[u][2][r], acc[u][3][r]));
  if (w == 0) {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      float s = db[u];
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      if (g == 0) pr[HM_NPAD * (long)C + 16 * u + l15] = s;
    }
  }
}

// block = (head k, 64 channels): 16 groups of 16 float4 lanes split the P partials, LDS-reduced; then per channel
//   dW_k[o][c] += gamma_k[c] dG'[n][c] + beta_k[c] db[n],  dgamma_k[c] += sum_o W_k[o][c] dG'[n][c],  dbeta_k[c] += sum_o W_k[o][c] db[n],  db_k[o] += db[n]
__global__ __launch_bounds__(256) void heads_bwd_fin_kernel(const float* __restrict__ partial, int P, HeadParams p, int K, int O, int C, HeadGrads gr) {
  const int k = blockIdx.y, c0 = blockIdx.x * 64, tid = threadIdx.x;
  const long stride = HM_NPAD * (long)C + HM_NPAD;
  __shared__ float red[16][8][64];
  __shared__ float rdb[32][8];
  {
    const int c4 = tid & 15, pg = tid >> 4;
    float4 s[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) s[o] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* src = partial + (long)k * O * C + c0 + 4 * c4;
    for (int q = pg; q < P; q += 16) {
#pragma unroll
      for (int o = 0; o < 8; ++o)
        if (o < O) {
          const float4 v = ld4(src + q * stride + (long)o * C);
          s[o].x += v.x; s[o].y += v.y; s[o].z += v.z; s[o].w += v.w;
        }
    }
#pragma unroll
    for (int o = 0; o < 8; ++o) *reinterpret_cast<float4*>(&red[pg][o][4 * c4]) = s[o];
  }
  {
    const int o = tid & 7, pg = tid >> 3;
    float a = 0.f;
    if (o < O)
      for (int q = pg; q < P; q += 32) a += partial[q * stride + HM_NPAD * (long)C + k * O + o];
    rdb[pg][o] = a;
  }
  __syncthreads();
  if (tid >= 64) return;
  const int c = c0 + tid;
  const float gm = p.gamma[k][c], bt = p.beta[k][c];
  float dg = 0.f, dbt = 0.f;
  for (int o = 0; o < O; ++o) {
    float v = 0.f, dbn = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) v += red[j][o][tid];
#pragma unroll
    for (int j = 0; j < 32; ++j) dbn += rdb[j][o];
    const float wv = p.W[k][(long)o * C + c];
    gr.W[k][(long)o * C + c] += gm * v + bt * dbn;
    dg += wv * v;
    dbt += wv * dbn;
    if (blockIdx.x == 0 && tid == o) gr.b[k][o] += dbn;
  }
  gr.gamma[k][c] += dg;
  gr.beta[k][c] += dbt;
}

// same contract as heads_bwd; `fold`, `out` = what this step's heads_fwd_mfma wrote (folded weights, head outputs); scratch >= 48 C + 48 floats (one partial), ideally cus times that
int heads_bwd_mfma(const float* x, const float* stats, const float* fold, const float* out, const HeadParams& p, const HeadGrads& gp, int K, int O,
                   const float* dout, float* dx, int M, int C, float* scratch, long scratch_floats, hipStream_t st, hipStream_t st_param) {
  MP_CHECK(heads_mfma_supported(K, O, C) && O <= 8, MP_ERR_ARG, "heads_bwd_mfma: K=%d O=%d C=%d unsupported", K, O, C);
  MP_CHECK((long)K * M * O < (1L << 31), MP_ERR_ARG, "heads_bwd_mfma: %d x %d x %d head outputs overflow the 32-bit offsets", K, M, O);
  const size_t lds = ((size_t)C * HM_GP + 2 * HM_NPAD) * sizeof(float);
#define MP_HM_DX(NS_, NQ4_)                                                                                                                      \
  do {                                                                                                                                           \
    static bool attr_set = false;                                                                                                                \
    if (!attr_set) {                                                                                                                             \
      MP_HIP(hipFuncSetAttribute((const void*)heads_bwd_dx_mfma_kernel<NS_, NQ4_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));   \
      attr_set = true;                                                                                                                           \
    }                                                                                                                                            \
    hipLaunchKernelGGL((heads_bwd_dx_mfma_kernel<NS_, NQ4_>), dim3(hm_grid(M / 4 + 1)), dim3(HM_DX_THREADS), lds, st, x, stats, fold, K, O, dout, \
                       out, dx, M);                                                                                                              \
  } while (0)
  const int NQ4 = cdiv(K * O, 16);
  if (C == 512)      { if (NQ4 == 3) MP_HM_DX(32, 3); else if (NQ4 == 2) MP_HM_DX(32, 2); else MP_HM_DX(32, 1); }
  else if (C == 256) { if (NQ4 == 3) MP_HM_DX(16, 3); else if (NQ4 == 2) MP_HM_DX(16, 2); else MP_HM_DX(16, 1); }
  else               { if (NQ4 == 3) MP_HM_DX(8, 3); else if (NQ4 == 2) MP_HM_DX(8, 2); else MP_HM_DX(8, 1); }
#undef MP_HM_DX
  MP_LAUNCH_CHECK();
  if (st_param != nullptr) st = st_param;
  const long stride = HM_NPAD * (long)C + HM_NPAD;
  const int P = (int)max(1L, min((long)hm_grid(M * 2), scratch_floats / stride));
  MP_CHECK(scratch_floats >= stride, MP_ERR_ARG, "heads_bwd_mfma: scratch too small");
  switch (cdiv(K * O, 16)) {
    case 1: hipLaunchKernelGGL((heads_bwd_param_mfma_kernel<8, 1>), dim3(P), dim3(C), 0, st, x, stats, K, O, dout, scratch, M, C); break;
    case 2: hipLaunchKernelGGL((heads_bwd_param_mfma_kernel<8, 2>), dim3(P), dim3(C), 0, st, x, stats, K, O, dout, scratch, M, C); break;
    default: hipLaunchKernelGGL((heads_bwd_param_mfma_kernel<8, 3>), dim3(P), dim3(C), 0, st, x, stats, K, O, dout, scratch, M, C); break;
  }
  MP_LAUNCH_CHECK();
  hipLaunchKernelGGL(heads_bwd_fin_kernel, dim3(C / 64, K), dim3(256), 0, st, scratch, P, p, K, O, C, gp);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

}  // namespace mp
