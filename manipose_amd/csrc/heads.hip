// Output heads of the lifting model:
//   * K hypothesis heads, MCLHead.forward (rmcl_manifold_mix_ste.py:291-298): LayerNorm(C, eps 1e-5) -> Linear(C, 7);
//     the plain MixSTE head Sequential(LayerNorm, Linear(C, out_dim)) (mix_ste.py:123-126) is the K = 1 case.
//     One wave per token: x-hat is computed once and shared by all K heads.
//   * score head: Linear over the JOINT axis of the 7th channel + softmax over K (rmcl_manifold_mix_ste.py:294-297, :262)
//   * bones net tail: mean over T of the per-frame segment lengths (manifold_mix_ste.py:153)
#include "common.h"
#include "kernels.h"

namespace mp {

constexpr int HV = 2;        // float4 per lane -> C <= 512
constexpr int HMAXO = 8;     // out features per head <= 8

__device__ __forceinline__ void head_row_xhat(const float* __restrict__ xr, int lane, int C, float eps, bool have_stats,
                                              float& mean, float& rstd, float4 (&xh)[HV]) {
#pragma unroll
  for (int i = 0; i < HV; ++i) {
    const int c = lane * 4 + 256 * i;
    xh[i] = (c < C) ? ld4(xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (!have_stats) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < HV; ++i) s += (xh[i].x + xh[i].y) + (xh[i].z + xh[i].w);   // padding lanes hold zeros
    mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < HV; ++i)
      if (lane * 4 + 256 * i < C) {
        const float a = xh[i].x - mean, b = xh[i].y - mean, c = xh[i].z - mean, d = xh[i].w - mean;
        q += (a * a + b * b) + (c * c + d * d);
      }
    rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
  }
#pragma unroll
  for (int i = 0; i < HV; ++i) {
    if (lane * 4 + 256 * i < C) {
      xh[i].x = (xh[i].x - mean) * rstd; xh[i].y = (xh[i].y - mean) * rstd;
      xh[i].z = (xh[i].z - mean) * rstd; xh[i].w = (xh[i].w - mean) * rstd;
    }
  }
}

__device__ __forceinline__ float dot4(const float4& a, const float4& b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }

// HR token rows per wave and iteration: every weight / gamma / beta vector fetched from cache serves HR rows (the one-row form
// re-read the K * O weight vectors for every token: 70 KB of L1 traffic per row, which bounded the kernel), and the HR wave
// reductions of an output are independent chains.  Per row the arithmetic and its order are unchanged.
constexpr int HR = 4;

__global__ __launch_bounds__(256) void heads_fwd_kernel(const float* __restrict__ x, HeadParams p, int K, int O,
                                                         float* __restrict__ out, float* __restrict__ stats, int M, int C) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int m0 = wave * HR; m0 < M; m0 += nwaves * HR) {
    float4 xh[HR][HV];
#pragma unroll
    for (int r = 0; r < HR; ++r) {
      const int m = min(m0 + r, M - 1);                       // rows past the end recompute the last row and are not stored
      float mean, rstd;
      head_row_xhat(x + (long)m * C, lane, C, 1e-5f, false, mean, rstd, xh[r]);
      if (lane == 0 && m0 + r < M) {
        stats[2 * (long)m] = mean;
        stats[2 * (long)m + 1] = rstd;
      }
    }
    for (int k = 0; k < K; ++k) {
      float4 y[HR][HV];
#pragma unroll
      for (int i = 0; i < HV; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < C) {
          const float4 g = lone4(ld4(p.gamma[k] + c)), b = lone4(ld4(p.beta[k] + c));      // splat over the HR rows: common.h, lone()
#pragma unroll
          for (int r = 0; r < HR; ++r)
            y[r][i] = make_float4(xh[r][i].x * g.x + b.x, xh[r][i].y * g.y + b.y, xh[r][i].z * g.z + b.z, xh[r][i].w * g.w + b.w);
        } else {
#pragma unroll
          for (int r = 0; r < HR; ++r) y[r][i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      for (int o = 0; o < O; ++o) {
        float s[HR];
#pragma unroll
        for (int r = 0; r < HR; ++r) s[r] = 0.f;
#pragma unroll
        for (int i = 0; i < HV; ++i) {
          const int c = lane * 4 + 256 * i;
          if (c < C) {
            const float4 w = lone4(ld4(p.W[k] + (long)o * C + c));
#pragma unroll
            for (int r = 0; r < HR; ++r) s[r] += dot4(y[r][i], w);
          }
        }
#pragma unroll
        for (int r = 0; r < HR; ++r) s[r] = wave_sum(s[r]);
        if (lane == 0) {
          const float bias = p.b[k][o];
#pragma unroll
          for (int r = 0; r < HR; ++r)
            if (m0 + r < M) out[((long)k * M + m0 + r) * O + o] = s[r] + bias;
        }
      }
    }
  }
}

int heads_fwd(const float* x, const HeadParams& p, int K, int O, float* out, float* stats, int M, int C, hipStream_t st) {
  MP_CHECK(K >= 1 && K <= 8 && O >= 1 && O <= HMAXO && C % 4 == 0 && C <= 256 * HV, MP_ERR_ARG,
           "heads_fwd: K=%d O=%d C=%d unsupported", K, O, C);
  hipLaunchKernelGGL(heads_fwd_kernel, dim3(max(1, min(cdiv(M, 4 * HR), 2048))), dim3(256), 0, st, x, p, K, O, out, stats, M, C);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

// dx = LN'( sum_k gamma_k * (W_k^T dout_k) )
__global__ __launch_bounds__(256) void heads_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                            HeadParams p, int K, int O, const float* __restrict__ dout,
                                                            float* __restrict__ dx, int M, int C) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int mm = wave * HR; mm < M; mm += nwaves * HR) {
    const int m0 = __builtin_amdgcn_readfirstlane(mm);
    float4 xh[HR][HV], d[HR][HV];
    float rstd[HR];
#pragma unroll
    for (int r = 0; r < HR; ++r) {
      const int m = min(m0 + r, M - 1);
      float mean = lone(stats[2 * (long)m]);
      rstd[r] = lone(stats[2 * (long)m + 1]);
      head_row_xhat(x + (long)m * C, lane, C, 1e-5f, true, mean, rstd[r], xh[r]);
#pragma unroll
      for (int i = 0; i < HV; ++i) d[r][i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int k = 0; k < K; ++k) {
      float4 dy[HR][HV];
#pragma unroll
      for (int r = 0; r < HR; ++r)
#pragma unroll
        for (int i = 0; i < HV; ++i) dy[r][i] = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int o = 0; o < O; ++o) {
        float g[HR];
#pragma unroll
        for (int r = 0; r < HR; ++r) g[r] = dout[((long)k * M + min(m0 + r, M - 1)) * O + o];
#pragma unroll
        for (int i = 0; i < HV; ++i) {
          const int c = lane * 4 + 256 * i;
          if (c < C) {
            const float4 w = ld4(p.W[k] + (long)o * C + c);
#pragma unroll
            for (int r = 0; r < HR; ++r) {
              dy[r][i].x += g[r] * w.x; dy[r][i].y += g[r] * w.y; dy[r][i].z += g[r] * w.z; dy[r][i].w += g[r] * w.w;
            }
          }
        }
      }
#pragma unroll
      for (int i = 0; i < HV; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < C) {
          const float4 gm = ld4(p.gamma[k] + c);
#pragma unroll
          for (int r = 0; r < HR; ++r) {
            d[r][i].x += dy[r][i].x * gm.x; d[r][i].y += dy[r][i].y * gm.y; d[r][i].z += dy[r][i].z * gm.z; d[r][i].w += dy[r][i].w * gm.w;
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < HR; ++r) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < HV; ++i) {
        s1 += (d[r][i].x + d[r][i].y) + (d[r][i].z + d[r][i].w);
        s2 += dot4(d[r][i], xh[r][i]);
      }
      s1 = wave_sum(s1) / (float)C;
      s2 = wave_sum(s2) / (float)C;
      if (m0 + r < M) {
#pragma unroll
        for (int i = 0; i < HV; ++i) {
          const int c = lane * 4 + 256 * i;
          if (c < C)
            st4(dx + (long)(m0 + r) * C + c,
                make_float4(rstd[r] * (d[r][i].x - s1 - xh[r][i].x * s2), rstd[r] * (d[r][i].y - s1 - xh[r][i].y * s2),
                            rstd[r] * (d[r][i].z - s1 - xh[r][i].z * s2), rstd[r] * (d[r][i].w - s1 - xh[r][i].w * s2)));
        }
      }
    }
  }
}

// parameter gradients: a block is K waves, wave k = head k, and all K waves of a block walk the SAME token rows, so the activation
// row is fetched from HBM once (the other heads hit it in cache) instead of once per head (5 x 541 MB at the bench size).  Each
// wave writes its own partial row [dW (O*C) | db (O) | dgamma (C) | dbeta (C)].
template <int O>
__global__ __launch_bounds__(512) void heads_bwd_param_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                               HeadParams p, int K, const float* __restrict__ dout,
                                                               float* __restrict__ partial, int M, int C) {
  const int lane = threadIdx.x & 63, k = threadIdx.x >> 6;
  const int wave = blockIdx.x;                              // row stream (partial row) index
  const int nwaves = gridDim.x;
  float4 W[O][HV], dW[O][HV], gm[HV], bt[HV], dg[HV], dbt[HV];
  float db[O];
#pragma unroll
  for (int i = 0; i < HV; ++i) {
    const int c = lane * 4 + 256 * i;
    const bool ok = c < C;
    gm[i] = ok ? ld4(p.gamma[k] + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    bt[i] = ok ? ld4(p.beta[k] + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    dg[i] = dbt[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int o = 0; o < O; ++o) {
      W[o][i] = ok ? ld4(p.W[k] + (long)o * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      dW[o][i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
#pragma unroll
  for (int o = 0; o < O; ++o) db[o] = 0.f;
  for (int m0 = wave; m0 < M; m0 += nwaves) {
    const int m = __builtin_amdgcn_readfirstlane(m0);
    float mean = lone(stats[2 * (long)m]), rstd = lone(stats[2 * (long)m + 1]);
    float4 xh[HV], dy[HV];
    head_row_xhat(x + (long)m * C, lane, C, 1e-5f, true, mean, rstd, xh);
#pragma unroll
    for (int i = 0; i < HV; ++i) dy[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int o = 0; o < O; ++o) {
      const float g = lone(dout[((long)k * M + m) * O + o]);
      db[o] += g;
#pragma unroll
      for (int i = 0; i < HV; ++i) {
        const float4 y = make_float4(xh[i].x * gm[i].x + bt[i].x, xh[i].y * gm[i].y + bt[i].y, xh[i].z * gm[i].z + bt[i].z,
                                     xh[i].w * gm[i].w + bt[i].w);
        dW[o][i].x += g * y.x; dW[o][i].y += g * y.y; dW[o][i].z += g * y.z; dW[o][i].w += g * y.w;
        dy[i].x += g * W[o][i].x; dy[i].y += g * W[o][i].y; dy[i].z += g * W[o][i].z; dy[i].w += g * W[o][i].w;
      }
    }
#pragma unroll
    for (int i = 0; i < HV; ++i) {
      dg[i].x += dy[i].x * xh[i].x; dg[i].y += dy[i].y * xh[i].y; dg[i].z += dy[i].z * xh[i].z; dg[i].w += dy[i].w * xh[i].w;
      dbt[i].x += dy[i].x; dbt[i].y += dy[i].y; dbt[i].z += dy[i].z; dbt[i].w += dy[i].w;
    }
  }
  const long nk = (long)O * C + O + 2 * C;
  float* pr = partial + ((long)wave * K + k) * nk;
#pragma unroll
  for (int i = 0; i < HV; ++i) {
    const int c = lane * 4 + 256 * i;
    if (c < C) {
#pragma unroll
      for (int o = 0; o < O; ++o) st4(pr + (long)o * C + c, dW[o][i]);
      st4(pr + (long)O * C + O + c, dg[i]);
      st4(pr + (long)O * C + O + C + c, dbt[i]);
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int o = 0; o < O; ++o) pr[(long)O * C + o] = db[o];
  }
}

__global__ void heads_reduce_kernel(const float* __restrict__ partial, int P, int K, int O, int C, HeadGrads g) {
  const int k = blockIdx.y;
  const int nk = O * C + O + 2 * C;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nk) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int p = 0;
  for (; p + 3 < P; p += 4) {
    s0 += partial[((long)p * K + k) * nk + i];
    s1 += partial[((long)(p + 1) * K + k) * nk + i];
    s2 += partial[((long)(p + 2) * K + k) * nk + i];
    s3 += partial[((long)(p + 3) * K + k) * nk + i];
  }
  for (; p < P; ++p) s0 += partial[((long)p * K + k) * nk + i];
  const float s = (s0 + s1) + (s2 + s3);
  if (i < O * C) g.W[k][i] += s;
  else if (i < O * C + O) g.b[k][i - O * C] += s;
  else if (i < O * C + O + C) g.gamma[k][i - O * C - O] += s;
  else g.beta[k][i - O * C - O - C] += s;
}

constexpr int HB_GRID = 128;  // 512 waves -> 512 partial rows per head

// st_param: stream of the parameter-gradient kernels (nothing downstream on `st` needs them); null = `st`.  The caller orders it
// after the producer of `dout` and owns `scratch` until those kernels are done.
int heads_bwd(const float* x, const float* stats, const HeadParams& p, const HeadGrads& gp, int K, int O, const float* dout,
              float* dx, int M, int C, float* scratch, long scratch_floats, hipStream_t st, hipStream_t st_param) {
  MP_CHECK(K >= 1 && K <= 8 && O >= 1 && O <= HMAXO && C % 4 == 0 && C <= 256 * HV, MP_ERR_ARG,
           "heads_bwd: K=%d O=%d C=%d unsupported", K, O, C);
  hipLaunchKernelGGL(heads_bwd_dx_kernel, dim3(max(1, min(cdiv(M, 4 * HR), 2048))), dim3(256), 0, st, x, stats, p, K, O, dout, dx, M,
                     C);
  MP_LAUNCH_CHECK();
  if (st_param != nullptr) st = st_param;
  const int P = max(1, min(M, 4 * HB_GRID));                 // row streams = partial rows per head
  const long nk = (long)O * C + O + 2 * C;
  MP_CHECK(scratch_floats >= (long)P * K * nk, MP_ERR_ARG, "heads_bwd: scratch too small");
  const dim3 g2(P), b2(64 * K);
  switch (O) {
    case 1: hipLaunchKernelGGL(heads_bwd_param_kernel<1>, g2, b2, 0, st, x, stats, p, K, dout, scratch, M, C); break;
    case 3: hipLaunchKernelGGL(heads_bwd_param_kernel<3>, g2, b2, 0, st, x, stats, p, K, dout, scratch, M, C); break;
    case 4: hipLaunchKernelGGL(heads_bwd_param_kernel<4>, g2, b2, 0, st, x, stats, p, K, dout, scratch, M, C); break;
    case 5: hipLaunchKernelGGL(heads_bwd_param_kernel<5>, g2, b2, 0, st, x, stats, p, K, dout, scratch, M, C); break;
    case 6: hipLaunchKernelGGL(heads_bwd_param_kernel<6>, g2, b2, 0, st, x, stats, p, K, dout, scratch, M, C); break;
    case 7: hipLaunchKernelGGL(heads_bwd_param_kernel<7>, g2, b2, 0, st, x, stats, p, K, dout, scratch, M, C); break;
    default: MP_CHECK(false, MP_ERR_ARG, "heads_bwd: out features %d unsupported (1,3,4,5,6,7)", O);
  }
  MP_LAUNCH_CHECK();
  hipLaunchKernelGGL(heads_reduce_kernel, dim3(cdiv(nk, 256), K), dim3(256), 0, st, scratch, P, K, O, C, gp);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

// ---------------------------------------------------------------------------------------------
// score head + softmax over K
// ---------------------------------------------------------------------------------------------
__global__ void scores_fwd_kernel(const float* __restrict__ headout, ScoreParams p, int K, int O, float* __restrict__ scores,
                                  int B, int T, int J) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;   // frame (b,t)
  if (f >= B * T) return;
  const long M = (long)B * T * J;
  float lg[8];
  float mx = -INFINITY;
  for (int k = 0; k < K; ++k) {
    float s = p.b[k][0];
    for (int j = 0; j < J; ++j) s += p.w[k][j] * headout[((long)k * M + (long)f * J + j) * O + (O - 1)];
    lg[k] = s;
    mx = fmaxf(mx, s);
  }
  float sum = 0.f;
  for (int k = 0; k < K; ++k) {
    lg[k] = expf(lg[k] - mx);
    sum += lg[k];
  }
  const int b = f / T, t = f % T;
  for (int k = 0; k < K; ++k) scores[((long)b * K + k) * T + t] = lg[k] / sum;
}

int scores_fwd(const float* headout, const ScoreParams& p, int K, int O, float* scores, int B, int T, int J, hipStream_t st) {
  MP_CHECK(K >= 1 && K <= 8, MP_ERR_ARG, "scores_fwd: K=%d unsupported", K);
  hipLaunchKernelGGL(scores_fwd_kernel, dim3(cdiv(B * T, 128)), dim3(128), 0, st, headout, p, K, O, scores, B, T, J);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

// dlogit = s * (ds - sum_k s ds); d(emb) = dlogit * w_k[j] written into channel O-1 of dheadout
__global__ void scores_bwd_kernel(const float* __restrict__ scores, const float* __restrict__ dscores, ScoreParams p, int K, int O,
                                  float* __restrict__ dheadout, float* __restrict__ dlogit, int B, int T, int J) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= B * T) return;
  const long M = (long)B * T * J;
  const int b = f / T, t = f % T;
  float s[8], d[8];
  float dot = 0.f;
  for (int k = 0; k < K; ++k) {
    s[k] = scores[((long)b * K + k) * T + t];
    d[k] = dscores[((long)b * K + k) * T + t];
    dot += s[k] * d[k];
  }
  for (int k = 0; k < K; ++k) {
    const float dl = s[k] * (d[k] - dot);
    dlogit[(long)k * B * T + f] = dl;
    for (int j = 0; j < J; ++j) dheadout[((long)k * M + (long)f * J + j) * O + (O - 1)] = dl * p.w[k][j];
  }
}

// dw_k[j] += sum_f dlogit[k][f] * emb[k][f][j]; db_k += sum_f dlogit[k][f]   (J <= 32).  Two launches: grid (SP_BLOCKS, K) sums frame
// slices into partial[k][block][33] (a thread owns whole frames: the J score channels of a frame are one 4 J O-byte span), then one block
// per head adds the partials in a fixed order into the gradients.
constexpr int SP_BLOCKS = 96;

__global__ __launch_bounds__(256) void scores_param_kernel(const float* __restrict__ headout, const float* __restrict__ dlogit,
                                                            float* __restrict__ partial, int O, int B, int T, int J) {
  __shared__ float red[4][33];
  const int k = blockIdx.y;
  const long M = (long)B * T * J;
  const int F = B * T;
  float s[33];
#pragma unroll
  for (int j = 0; j < 33; ++j) s[j] = 0.f;
  for (int f = blockIdx.x * 256 + threadIdx.x; f < F; f += SP_BLOCKS * 256) {
    const float dl = dlogit[(long)k * F + f];
    const float* e = headout + ((long)k * M + (long)f * J) * O + (O - 1);
#pragma unroll
    for (int j = 0; j < 32; ++j)
      if (j < J) s[j] += dl * e[(long)j * O];
    s[32] += dl;
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < 33; ++j) {
    const float v = wave_sum(s[j]);
    if (lane == 0) red[w][j] = v;
  }
  __syncthreads();
  if (threadIdx.x < 33) partial[((long)k * SP_BLOCKS + blockIdx.x) * 33 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ __launch_bounds__(64) void scores_param_fin_kernel(const float* __restrict__ partial, ScoreGrads g, int J) {
  const int k = blockIdx.x, j = threadIdx.x;
  if (j > 32 || (j >= J && j != 32)) return;
  float a = 0.f;
  for (int b = 0; b < SP_BLOCKS; ++b) a += partial[((long)k * SP_BLOCKS + b) * 33 + j];
  if (j < J) g.w[k][j] += a;
  else g.b[k][0] += a;
}

// ev / st_param: when given, the parameter kernel runs on st_param after `ev` (recorded here on `st` behind the kernel that fills
// `scratch` and the score channel of dheadout); the caller owns `scratch` until it is done
long scores_bwd_scratch_floats(int K, int B, int T) { return (long)K * B * T + (long)K * SP_BLOCKS * 33; }

int scores_bwd(const float* headout, const float* scores, const float* dscores, const ScoreParams& p, const ScoreGrads& gp, int K,
               int O, float* dheadout, int B, int T, int J, float* scratch, long scratch_floats, hipStream_t st, hipStream_t st_param,
               hipEvent_t ev) {
  MP_CHECK(K >= 1 && K <= 8 && J <= 32, MP_ERR_ARG, "scores_bwd: K=%d J=%d unsupported", K, J);
  MP_CHECK(scratch_floats >= scores_bwd_scratch_floats(K, B, T), MP_ERR_ARG, "scores_bwd: scratch too small");
  hipLaunchKernelGGL(scores_bwd_kernel, dim3(cdiv(B * T, 128)), dim3(128), 0, st, scores, dscores, p, K, O, dheadout, scratch, B, T,
                     J);
  MP_LAUNCH_CHECK();
  if (st_param != nullptr && ev != nullptr) {
    MP_HIP(hipEventRecord(ev, st));
    MP_HIP(hipStreamWaitEvent(st_param, ev, 0));
    st = st_param;
  }
  float* partial = scratch + (long)K * B * T;
  hipLaunchKernelGGL(scores_param_kernel, dim3(SP_BLOCKS, K), dim3(256), 0, st, headout, scratch, partial, O, B, T, J);
  MP_LAUNCH_CHECK();
  hipLaunchKernelGGL(scores_param_fin_kernel, dim3(K), dim3(64), 0, st, partial, gp, J);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

// ---------------------------------------------------------------------------------------------
// bones tail: lengths[b][s] = mean_t headout[(b,t,s)]
// ---------------------------------------------------------------------------------------------
__global__ void bones_mean_fwd_kernel(const float* __restrict__ headout, float* __restrict__ lengths, int B, int T, int S) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * S) return;
  const int b = i / S, s = i % S;
  float a = 0.f;
  for (int t = 0; t < T; ++t) a += headout[((long)b * T + t) * S + s];
  lengths[i] = a / (float)T;
}

int bones_mean_fwd(const float* headout, float* lengths, int B, int T, int S, hipStream_t st) {
  hipLaunchKernelGGL(bones_mean_fwd_kernel, dim3(cdiv(B * S, 64)), dim3(64), 0, st, headout, lengths, B, T, S);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

// block per batch item: dlengths[b][s] = sum over the item's KT decoded poses; dheadout[(b,t,s)] = dlengths / T
__global__ __launch_bounds__(256) void bones_mean_bwd_kernel(const float* __restrict__ dlen_pose, int KT, float* __restrict__ dlengths,
                                                              float* __restrict__ dheadout, int T, int S) {
  __shared__ float red[256];
  __shared__ float tot[32];
  const int b = blockIdx.x, s = threadIdx.x % S, sub = threadIdx.x / S, nsub = 256 / S;
  float a = 0.f;
  if (sub < nsub) {
    const float* p = dlen_pose + (long)b * KT * S + s;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;          // four loads in flight instead of one serial chain
    int i = sub;
    for (; i + 3 * nsub < KT; i += 4 * nsub) {
      a0 += p[(long)i * S]; a1 += p[(long)(i + nsub) * S]; a2 += p[(long)(i + 2 * nsub) * S]; a3 += p[(long)(i + 3 * nsub) * S];
    }
    for (; i < KT; i += nsub) a0 += p[(long)i * S];
    a = (a0 + a1) + (a2 + a3);
  }
  red[threadIdx.x] = (sub < nsub) ? a : 0.f;
  __syncthreads();
  if (threadIdx.x < S) {
    float t = 0.f;
    for (int u = 0; u < nsub; ++u) t += red[u * S + threadIdx.x];
    tot[threadIdx.x] = t;
    if (dlengths != nullptr) dlengths[b * S + threadIdx.x] = t;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < T * S; i += 256) dheadout[(long)b * T * S + i] = tot[i % S] / (float)T;
}

int bones_mean_bwd(const float* dlen_pose, int KT, float* dlengths, float* dheadout, int B, int T, int S, hipStream_t st) {
  MP_CHECK(S <= 32, MP_ERR_ARG, "bones_mean_bwd: S=%d > 32", S);
  hipLaunchKernelGGL(bones_mean_bwd_kernel, dim3(B), dim3(256), 0, st, dlen_pose, KT, dlengths, dheadout, T, S);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

}  // namespace mp
