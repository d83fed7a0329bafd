// fp32 GEMM family on the gfx950 f32-input matrix cores (v_mfma_f32_32x32x2_f32): exact fp32 products with
// fp32 accumulation, i.e. the same arithmetic class as the reference's ATen mm/addmm (mix_ste.py:257-261,
// 280-281, 216-222).  This is the "parity" precision mode; the bf16 throughput GEMMs live in gemm_bf16.hip.
//
//   C[M,N] = A(i,r) * B(r,j)  reduced over r in [0,K)
//     AL = 0 : A stored [M][K] (reduction contiguous)      AL = 1 : A stored [K][M] (output-row contiguous)
//     BL = 0 : B stored [N][K] (nn.Linear weight layout)   BL = 1 : B stored [K][N]
//   forward  Y = X W^T      : AL=0, BL=0          (X tokens x in, W out x in)
//   dgrad    dX = dY W      : AL=0, BL=1          (reduction over the Linear's out features)
//   wgrad    dW = dY^T X    : AL=1, BL=1          (reduction over tokens, split over blockIdx.z into slabs)
//
// Tile: 128x128x32 per 256-thread workgroup, 4 waves as 2x2, each wave 64x64 = 2x2 MFMA 32x32 tiles.
// LDS images are [k][128+4] floats for both operands so that the MFMA fragment reads (lane -> consecutive
// output index at fixed k) are conflict-free ds_read_b32; one float per lane per operand per MFMA, so no
// contiguity requirement on the reduction axis (this is what makes all three layouts one kernel).
#include "common.h"
#include "kernels.h"

namespace mp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32, LD = 132;

template <int L>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, long ld, int out0, int r0, int OUT, int RED,
                                          int r_end, int tid, float4 (&v)[4]) {
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int idx = tid + 256 * it;
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (L == 0) {  // stored [out][red], float4 along red
      const int row = idx >> 3, kq = idx & 7;
      const int o = out0 + row, r = r0 + 4 * kq;
      if (o < OUT && r < r_end) x = ld4(P + (long)o * ld + r);
    } else {  // stored [red][out], float4 along out
      const int kk = idx >> 5, mq = idx & 31;
      const int r = r0 + kk, o = out0 + 4 * mq;
      if (r < r_end && o < OUT) x = ld4(P + (long)r * ld + o);
    }
    v[it] = x;
  }
}

template <int L>
__device__ __forceinline__ void store_tile(float* __restrict__ S, int tid, const float4 (&v)[4]) {
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int idx = tid + 256 * it;
    if (L == 0) {
      const int row = idx >> 3, kq = idx & 7;
      S[(4 * kq + 0) * LD + row] = v[it].x;
      S[(4 * kq + 1) * LD + row] = v[it].y;
      S[(4 * kq + 2) * LD + row] = v[it].z;
      S[(4 * kq + 3) * LD + row] = v[it].w;
    } else {
      const int kk = idx >> 5, mq = idx & 31;
      *reinterpret_cast<float4*>(&S[kk * LD + 4 * mq]) = v[it];
    }
  }
}

template <int AL, int BL, int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmF32Args g) {
  __shared__ __attribute__((aligned(16))) float smem[2 * BK * LD];
  float* As = smem;
  float* Bs = smem + BK * LD;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  // split-K range (only the wgrad launcher uses gridDim.z > 1)
  const int kper = g.k_per_split;
  const int kbeg = blockIdx.z * kper;
  const int kend = min(g.K, kbeg + kper);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float bsum = 0.f;  // EPI_SLAB: column sums of the A tile (bias gradient), first 128 threads of n-tile 0
  float4 va[4], vb[4];
  load_tile<AL>(g.A, g.lda, m0, kbeg, g.M, g.K, kend, tid, va);
  load_tile<BL>(g.B, g.ldb, n0, kbeg, g.N, g.K, kend, tid, vb);

  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    __syncthreads();
    store_tile<AL>(As, tid, va);
    store_tile<BL>(Bs, tid, vb);
    __syncthreads();
    if (k0 + BK < kend) {
      load_tile<AL>(g.A, g.lda, m0, k0 + BK, g.M, g.K, kend, tid, va);
      load_tile<BL>(g.B, g.ldb, n0, k0 + BK, g.N, g.K, kend, tid, vb);
    }
    if (EPI == EPI_SLAB) {
      if (blockIdx.x == 0 && tid < BM) {
#pragma unroll 8
        for (int kk = 0; kk < BK; ++kk) bsum += As[kk * LD + tid];
      }
    }
#pragma unroll 4
    for (int s = 0; s < BK / 2; ++s) {
      const float* ap = As + (2 * s + h) * LD + wr * 64 + l31;
      const float* bp = Bs + (2 * s + h) * LD + wc * 64 + l31;
      const float a0 = ap[0], a1 = ap[32];
      const float b0 = bp[0], b1 = bp[32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
  }

  // ---- epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
  float* C = g.C;
  if (EPI == EPI_SLAB) {
    C += (long)blockIdx.z * g.M * g.ldc;
    if (blockIdx.x == 0 && tid < BM && m0 + tid < g.M && g.bias_slab != nullptr)
      g.bias_slab[(long)blockIdx.z * g.M + m0 + tid] = bsum;
  }
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col = n0 + wc * 64 + ni * 32 + l31;
      if (col >= g.N) continue;
      const float bias = (EPI != EPI_SLAB && EPI != EPI_DGELU && g.bias != nullptr) ? g.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wr * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row >= g.M) continue;
        const long o = (long)row * g.ldc + col;
        float v = acc[mi][ni][r] + bias;
        if (EPI == EPI_BIAS_GELU) {
          g.Z[o] = gelu_grad_f(v);      // Z keeps gelu'(z): all the backward needs from the pre-activation
          v = gelu_f(v);
        } else if (EPI == EPI_BIAS_RESID) {
          v = g.R[o] * (g.rscale != 0.f ? g.rscale : 1.0f) + droppath_scale(g.mask, g.mask_mode, row, g.T, g.J) * v;
        } else if (EPI == EPI_DGELU) {
          v *= g.Z[o];
        }
        C[o] = v;
      }
    }
  }
}

// sum split-K slabs into (+=) the gradient buffers: dW[i] += sum_s slab[s][i]; db likewise
__global__ void reduce_slabs_kernel(const float* __restrict__ slab, float* __restrict__ out, long n, int S) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < S; ++k) s += slab[(long)k * n + i];
  out[i] += s;
}

template <int AL, int BL, int EPI>
static int launch(const GemmF32Args& g, int splits, hipStream_t st) {
  dim3 grid(cdiv(g.N, BN), cdiv(g.M, BM), splits);
  hipLaunchKernelGGL((gemm_f32_kernel<AL, BL, EPI>), grid, dim3(256), 0, st, g);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

int gemm_f32(int AL, int BL, int EPI, GemmF32Args g, hipStream_t st) {
  MP_CHECK(g.M > 0 && g.N > 0 && g.K > 0, MP_ERR_ARG, "gemm_f32: empty problem %d %d %d", g.M, g.N, g.K);
  MP_CHECK(g.K % 4 == 0 || AL == 1, MP_ERR_ARG, "gemm_f32: K=%d must be a multiple of 4", g.K);
  MP_CHECK(g.N % 4 == 0 || BL == 0, MP_ERR_ARG, "gemm_f32: N=%d must be a multiple of 4", g.N);
  MP_CHECK(AL == 0 || g.M % 4 == 0, MP_ERR_ARG, "gemm_f32: M=%d must be a multiple of 4 for AL=1", g.M);
  MP_CHECK((g.lda % 4 == 0) && (g.ldb % 4 == 0), MP_ERR_ARG, "gemm_f32: lda/ldb must be multiples of 4");
  if (EPI != EPI_SLAB) g.k_per_split = ((g.K + BK - 1) / BK) * BK;
  if (AL == 0 && BL == 0 && EPI == EPI_BIAS) return launch<0, 0, EPI_BIAS>(g, 1, st);
  if (AL == 0 && BL == 0 && EPI == EPI_BIAS_GELU) return launch<0, 0, EPI_BIAS_GELU>(g, 1, st);
  if (AL == 0 && BL == 0 && EPI == EPI_BIAS_RESID) return launch<0, 0, EPI_BIAS_RESID>(g, 1, st);
  if (AL == 0 && BL == 1 && EPI == EPI_BIAS) return launch<0, 1, EPI_BIAS>(g, 1, st);
  if (AL == 0 && BL == 1 && EPI == EPI_DGELU) return launch<0, 1, EPI_DGELU>(g, 1, st);
  MP_CHECK(false, MP_ERR_ARG, "gemm_f32: unsupported variant AL=%d BL=%d EPI=%d", AL, BL, EPI);
}

static void wgrad_split(int Mtok, int Nout, int Kin, int& splits, int& kper) {
  const int tiles = cdiv(Nout, BM) * cdiv(Kin, BN);
  splits = max(1, min(64, (1024 + tiles - 1) / tiles));   // aim at >= 4 workgroups per CU
  kper = ((cdiv(Mtok, splits) + BK - 1) / BK) * BK;
  splits = cdiv(Mtok, kper);
}
long wgrad_f32_slab_floats(int Mtok, int Nout, int Kin) {
  (void)Mtok;   // upper bound over every token count: the split count never exceeds the tile-derived target
  const int tiles = cdiv(Nout, BM) * cdiv(Kin, BN);                  // 128x128 tiles (fp32 and narrow bf16 layers)
  const int tiles256 = cdiv(Nout, 256) * cdiv(Kin, 256);             // 256x256 tiles of the bf16 direct-to-LDS kernel
  const int s128 = max(1, min(256, (1024 + tiles - 1) / tiles)), s256 = max(1, min(64, (512 + tiles256 - 1) / tiles256));      // (bounds of wgrad_split / wgrad_split_b16)
  return ((long)Nout * Kin + Nout) * max(s128, s256);
}

// dW[N',K'] += dY[Mtok,N']^T X[Mtok,K'] ; db[N'] += colsum(dY).  slab: >= splits*(N'*K' + N') floats.
int wgrad_f32(const float* dY, long lddy, const float* X, long ldx, int Mtok, int Nout, int Kin, float* dW, float* db,
              float* slab, long slab_floats, hipStream_t st) {
  MP_CHECK(Mtok > 0 && Nout % 4 == 0 && Kin % 4 == 0, MP_ERR_ARG, "wgrad_f32: bad dims %d %d %d", Mtok, Nout, Kin);
  int splits, kper;
  wgrad_split(Mtok, Nout, Kin, splits, kper);
  const long per = (long)Nout * Kin + Nout;
  MP_CHECK(slab_floats >= per * splits, MP_ERR_ARG, "wgrad_f32: slab too small (%ld < %ld)", slab_floats, per * splits);
  GemmF32Args g = {};
  g.A = dY; g.lda = lddy; g.B = X; g.ldb = ldx;
  g.M = Nout; g.N = Kin; g.K = Mtok;
  g.C = slab; g.ldc = Kin;
  g.bias_slab = (db != nullptr) ? slab + (long)splits * Nout * Kin : nullptr;
  g.k_per_split = kper;
  int rc = launch<1, 1, EPI_SLAB>(g, splits, st);
  if (rc) return rc;
  const long n = (long)Nout * Kin;
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, slab, dW, n, splits);
  MP_LAUNCH_CHECK();
  if (db != nullptr) {
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(cdiv(Nout, 256)), dim3(256), 0, st, g.bias_slab, db, (long)Nout, splits);
    MP_LAUNCH_CHECK();
  }
  return MP_OK;
}

}  // namespace mp
