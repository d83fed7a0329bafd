// fp32 attention kernels of the MixSTE blocks (Attention.forward, mix_ste.py:255-282), operating in place on
// the canonical token layout m = (b*T + t)*J + j so that the reference's 16 "(B L) J C <-> (B J) L C"
// transpose copies (mix_ste.py:144,167,171) never happen:
//   spatial  : attends over the J (17 joints / 16 bones) tokens of one frame  -> contiguous rows
//   temporal : attends over the T frames of one joint                         -> rows strided by J*3C
// These are the exact-fp32 (parity mode) kernels: VALU dot products out of LDS, softmax in fp32.
#include "common.h"
#include "kernels.h"

namespace mp {

// =============================================================================================
// spatial attention: one wave per (frame, head); N = J <= 32 tokens, D = head dim (multiple of 4)
// =============================================================================================
template <bool BWD>
__device__ __forceinline__ int spatial_lds_floats(int N, int D) {
  const int DP = D + 4, NP = N + 1;
  return ((BWD ? (N * D + N * DP + N * DP + N * D + 2 * N * NP) : (N * D + N * DP + N * D + N * NP)) + 3) & ~3;
}
static int spatial_lds_floats_host(bool bwd, int N, int D) {
  const int DP = D + 4, NP = N + 1;
  return ((bwd ? (N * D + N * DP + N * DP + N * D + 2 * N * NP) : (N * D + N * DP + N * D + N * NP)) + 3) & ~3;   // 16-B aligned per wave
}

template <int WPB, typename TS>
__global__ __launch_bounds__(WPB * 64) void attn_spatial_fwd_kernel(const TS* __restrict__ qkv, TS* __restrict__ out,
                                                                     int F, int N, int C, int H, int D, float scale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int DP = D + 4, NP = N + 1, D4 = D >> 2;
  float* q = sm + wv * spatial_lds_floats<false>(N, D);
  float* k = q + N * D;
  float* v = k + N * DP;
  float* p = v + N * D;
  const int unit = blockIdx.x * WPB + wv;
  const bool active = unit < F * H;
  const int f = active ? unit / H : 0, h = active ? unit % H : 0;
  const long m0 = (long)f * N;
  const int C3 = 3 * C;
  if (active) {
    for (int idx = lane; idx < N * D4; idx += 64) {
      const int i = idx / D4, c = (idx - i * D4) * 4;
      const TS* r = qkv + (m0 + i) * C3 + h * D + c;
      *reinterpret_cast<float4*>(q + i * D + c) = ld4(r);
      *reinterpret_cast<float4*>(k + i * DP + c) = ld4(r + C);
      *reinterpret_cast<float4*>(v + i * D + c) = ld4(r + 2 * C);
    }
  }
  __syncthreads();
  if (active) {
    for (int idx = lane; idx < N * N; idx += 64) {
      const int i = idx / N, j = idx - i * N;
      float s = 0.f;
      for (int c = 0; c < D; c += 4) {
        const float4 a = *reinterpret_cast<const float4*>(q + i * D + c);
        const float4 b = *reinterpret_cast<const float4*>(k + j * DP + c);
        s += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
      }
      p[i * NP + j] = s * scale;
    }
  }
  __syncthreads();
  if (active && lane < N) {
    float* pr = p + lane * NP;
    float mx = -INFINITY;
    for (int j = 0; j < N; ++j) mx = fmaxf(mx, pr[j]);
    float sum = 0.f;
    for (int j = 0; j < N; ++j) {
      const float e = expf(pr[j] - mx);
      pr[j] = e;
      sum += e;
    }
    const float inv = 1.0f / sum;
    for (int j = 0; j < N; ++j) pr[j] *= inv;
  }
  __syncthreads();
  if (active) {
    for (int idx = lane; idx < N * D4; idx += 64) {
      const int i = idx / D4, c = (idx - i * D4) * 4;
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int j = 0; j < N; ++j) {
        const float w = p[i * NP + j];
        const float4 b = *reinterpret_cast<const float4*>(v + j * D + c);
        o.x += w * b.x; o.y += w * b.y; o.z += w * b.z; o.w += w * b.w;
      }
      st4(out + (m0 + i) * C + h * D + c, o);
    }
  }
}

template <int WPB, typename TS>
__global__ __launch_bounds__(WPB * 64) void attn_spatial_bwd_kernel(const TS* __restrict__ qkv, const TS* __restrict__ dout,
                                                                     TS* __restrict__ dqkv, int F, int N, int C, int H, int D,
                                                                     float scale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int DP = D + 4, NP = N + 1, D4 = D >> 2;
  float* q = sm + wv * spatial_lds_floats<true>(N, D);
  float* k = q + N * D;
  float* v = k + N * DP;
  float* go = v + N * DP;
  float* p = go + N * D;
  float* ds = p + N * NP;
  const int unit = blockIdx.x * WPB + wv;
  const bool active = unit < F * H;
  const int f = active ? unit / H : 0, h = active ? unit % H : 0;
  const long m0 = (long)f * N;
  const int C3 = 3 * C;
  if (active) {
    for (int idx = lane; idx < N * D4; idx += 64) {
      const int i = idx / D4, c = (idx - i * D4) * 4;
      const TS* r = qkv + (m0 + i) * C3 + h * D + c;
      *reinterpret_cast<float4*>(q + i * D + c) = ld4(r);
      *reinterpret_cast<float4*>(k + i * DP + c) = ld4(r + C);
      *reinterpret_cast<float4*>(v + i * DP + c) = ld4(r + 2 * C);
      *reinterpret_cast<float4*>(go + i * D + c) = ld4(dout + (m0 + i) * C + h * D + c);
    }
  }
  __syncthreads();
  if (active) {
    for (int idx = lane; idx < N * N; idx += 64) {
      const int i = idx / N, j = idx - i * N;
      float s = 0.f, dp = 0.f;
      for (int c = 0; c < D; c += 4) {
        const float4 a = *reinterpret_cast<const float4*>(q + i * D + c);
        const float4 b = *reinterpret_cast<const float4*>(k + j * DP + c);
        const float4 g = *reinterpret_cast<const float4*>(go + i * D + c);
        const float4 w = *reinterpret_cast<const float4*>(v + j * DP + c);
        s += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
        dp += (g.x * w.x + g.y * w.y) + (g.z * w.z + g.w * w.w);
      }
      p[i * NP + j] = s * scale;
      ds[i * NP + j] = dp;
    }
  }
  __syncthreads();
  if (active && lane < N) {
    float* pr = p + lane * NP;
    float* dr = ds + lane * NP;
    float mx = -INFINITY;
    for (int j = 0; j < N; ++j) mx = fmaxf(mx, pr[j]);
    float sum = 0.f;
    for (int j = 0; j < N; ++j) {
      const float e = expf(pr[j] - mx);
      pr[j] = e;
      sum += e;
    }
    const float inv = 1.0f / sum;
    float r = 0.f;
    for (int j = 0; j < N; ++j) {
      pr[j] *= inv;
      r += pr[j] * dr[j];
    }
    for (int j = 0; j < N; ++j) dr[j] = pr[j] * (dr[j] - r) * scale;   // d(scores) incl. the qk scale
  }
  __syncthreads();
  if (active) {
    for (int idx = lane; idx < N * D4; idx += 64) {
      const int i = idx / D4, c = (idx - i * D4) * 4;
      float4 dq = make_float4(0.f, 0.f, 0.f, 0.f), dk = dq, dv = dq;
      for (int j = 0; j < N; ++j) {
        const float sij = ds[i * NP + j], sji = ds[j * NP + i], pji = p[j * NP + i];
        const float4 kj = *reinterpret_cast<const float4*>(k + j * DP + c);
        const float4 qj = *reinterpret_cast<const float4*>(q + j * D + c);
        const float4 gj = *reinterpret_cast<const float4*>(go + j * D + c);
        dq.x += sij * kj.x; dq.y += sij * kj.y; dq.z += sij * kj.z; dq.w += sij * kj.w;
        dk.x += sji * qj.x; dk.y += sji * qj.y; dk.z += sji * qj.z; dk.w += sji * qj.w;
        dv.x += pji * gj.x; dv.y += pji * gj.y; dv.z += pji * gj.z; dv.w += pji * gj.w;
      }
      TS* r = dqkv + (m0 + i) * C3 + h * D + c;
      st4(r, dq);
      st4(r + C, dk);
      st4(r + 2 * C, dv);
    }
  }
}

// head_dim ** -0.5 (mix_ste.py:243-244) unless the model engine has set another scale for the launches it is about to issue on this
// thread (muP: 1 / head_dim, mix_ste.py:243; or an explicit qk_scale): attn_scale_override(s > 0), reset with 0
static thread_local float g_scale_override = 0.f;
void attn_scale_override(float s) { g_scale_override = s; }
float attn_qk_scale(int D) { return g_scale_override > 0.f ? g_scale_override : 1.0f / sqrtf((float)D); }
static float qk_scale(int D) { return attn_qk_scale(D); }

template <typename TS>
static int spatial_fwd_t(const TS* qkv, TS* out, int B, int T, int J, int C, int H, hipStream_t st) {
  const int D = C / H, F = B * T;
  constexpr int WPB = 4;
  const size_t lds = (size_t)WPB * spatial_lds_floats_host(false, J, D) * sizeof(float);
  MP_CHECK(lds <= 160 * 1024, MP_ERR_ARG, "attn_spatial_fwd: LDS %zu too large", lds);
  hipLaunchKernelGGL((attn_spatial_fwd_kernel<WPB, TS>), dim3(cdiv((long)F * H, WPB)), dim3(WPB * 64), lds, st, qkv, out, F, J, C, H,
                     D, qk_scale(D));
  MP_LAUNCH_CHECK();
  return MP_OK;
}
template <typename TS>
static int spatial_bwd_t(const TS* qkv, const TS* dout, TS* dqkv, int B, int T, int J, int C, int H, hipStream_t st) {
  const int D = C / H, F = B * T;
  constexpr int WPB = 2;
  const size_t lds = (size_t)WPB * spatial_lds_floats_host(true, J, D) * sizeof(float);
  MP_CHECK(lds <= 160 * 1024, MP_ERR_ARG, "attn_spatial_bwd: LDS %zu too large", lds);
  hipLaunchKernelGGL((attn_spatial_bwd_kernel<WPB, TS>), dim3(cdiv((long)F * H, WPB)), dim3(WPB * 64), lds, st, qkv, dout, dqkv, F, J,
                     C, H, D, qk_scale(D));
  MP_LAUNCH_CHECK();
  return MP_OK;
}

bool attn_smfma_supported(int N, int D, int H);
int attn_smfma_fwd(const bf16* qkv, bf16* out, int B, int T, int J, int C, int H, hipStream_t st);
int attn_smfma_bwd(const bf16* qkv, const bf16* dout, bf16* dqkv, int B, int T, int J, int C, int H, hipStream_t st);

int attn_spatial_fwd(const void* qkv, void* out, int is_bf16, int B, int T, int J, int C, int H, hipStream_t st) {
  MP_CHECK(C % H == 0 && (C / H) % 4 == 0 && J <= 32, MP_ERR_ARG, "attn_spatial_fwd: C=%d H=%d J=%d unsupported", C, H, J);
  if (is_bf16 && attn_smfma_supported(J, C / H, H)) return attn_smfma_fwd((const bf16*)qkv, (bf16*)out, B, T, J, C, H, st);
  return is_bf16 ? spatial_fwd_t<bf16>((const bf16*)qkv, (bf16*)out, B, T, J, C, H, st)
                 : spatial_fwd_t<float>((const float*)qkv, (float*)out, B, T, J, C, H, st);
}
int attn_spatial_bwd(const void* qkv, const void* dout, void* dqkv, int is_bf16, int B, int T, int J, int C, int H, hipStream_t st) {
  MP_CHECK(C % H == 0 && (C / H) % 4 == 0 && J <= 32, MP_ERR_ARG, "attn_spatial_bwd: C=%d H=%d J=%d unsupported", C, H, J);
  if (is_bf16 && attn_smfma_supported(J, C / H, H))
    return attn_smfma_bwd((const bf16*)qkv, (const bf16*)dout, (bf16*)dqkv, B, T, J, C, H, st);
  return is_bf16 ? spatial_bwd_t<bf16>((const bf16*)qkv, (const bf16*)dout, (bf16*)dqkv, B, T, J, C, H, st)
                 : spatial_bwd_t<float>((const float*)qkv, (const float*)dout, (float*)dqkv, B, T, J, C, H, st);
}

// =============================================================================================
// temporal attention: grid (B*J*H, ceil(T/256)); one thread per query (fwd, dQ) or per key (dK/dV);
// the other side streams through LDS in tiles of KT rows, read as wave-uniform broadcasts.
// =============================================================================================
constexpr int KT = 16;

template <int D, typename TS>
__device__ __forceinline__ void load_rows_tile(float* __restrict__ S, const TS* __restrict__ base, long row_stride, int r0,
                                               int R, int tid) {
  constexpr int D4 = D / 4;
  for (int idx = tid; idx < KT * D4; idx += 256) {
    const int r = idx / D4, c = (idx - r * D4) * 4;
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + r < R) x = ld4(base + (long)(r0 + r) * row_stride + c);
    *reinterpret_cast<float4*>(S + r * D + c) = x;
  }
}

template <int D, typename TS>
__global__ __launch_bounds__(256) void attn_temporal_fwd_kernel(const TS* __restrict__ qkv, TS* __restrict__ out,
                                                                 float* __restrict__ lse, int T, int J, int C, int H,
                                                                 float scale) {
  __shared__ __attribute__((aligned(16))) float Ks[KT * D];
  __shared__ __attribute__((aligned(16))) float Vs[KT * D];
  const int unit = blockIdx.x, h = unit % H, bj = unit / H, j = bj % J, b = bj / J;
  const int t = blockIdx.y * 256 + threadIdx.x;
  const bool active = t < T;
  const long rs3 = (long)J * 3 * C;
  const TS* qb = qkv + ((long)b * T * J + j) * 3 * C + h * D;   // row of frame 0
  float q[D], o[D];
#pragma unroll
  for (int c = 0; c < D; c += 4) {
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) x = ld4(qb + (long)t * rs3 + c);
    q[c] = x.x * scale; q[c + 1] = x.y * scale; q[c + 2] = x.z * scale; q[c + 3] = x.w * scale;
    o[c] = o[c + 1] = o[c + 2] = o[c + 3] = 0.f;
  }
  float mrun = -INFINITY, l = 0.f;
  for (int k0 = 0; k0 < T; k0 += KT) {
    __syncthreads();
    load_rows_tile<D>(Ks, qb + C, rs3, k0, T, threadIdx.x);
    load_rows_tile<D>(Vs, qb + 2 * C, rs3, k0, T, threadIdx.x);
    __syncthreads();
    float s[KT];
    float tmax = -INFINITY;
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) {
      float a = 0.f;
#pragma unroll
      for (int c = 0; c < D; c += 4) {
        const float4 kv = *reinterpret_cast<const float4*>(Ks + kk * D + c);
        a += (q[c] * kv.x + q[c + 1] * kv.y) + (q[c + 2] * kv.z + q[c + 3] * kv.w);
      }
      s[kk] = (k0 + kk < T) ? a : -INFINITY;
      tmax = fmaxf(tmax, s[kk]);
    }
    const float mnew = fmaxf(mrun, tmax);
    const float alpha = expf(mrun - mnew);
    l *= alpha;
#pragma unroll
    for (int c = 0; c < D; ++c) o[c] *= alpha;
#pragma unroll
    for (int kk = 0; kk < KT; ++kk) {
      const float pe = expf(s[kk] - mnew);
      l += pe;
#pragma unroll
      for (int c = 0; c < D; c += 4) {
        const float4 vv = *reinterpret_cast<const float4*>(Vs + kk * D + c);
        o[c] += pe * vv.x; o[c + 1] += pe * vv.y; o[c + 2] += pe * vv.z; o[c + 3] += pe * vv.w;
      }
    }
    mrun = mnew;
  }
  if (active) {
    const float inv = 1.0f / l;
    TS* orow = out + ((long)(b * T + t) * J + j) * C + h * D;
#pragma unroll
    for (int c = 0; c < D; c += 4) st4(orow + c, make_float4(o[c] * inv, o[c + 1] * inv, o[c + 2] * inv, o[c + 3] * inv));
    lse[(long)unit * T + t] = mrun + logf(l);
  }
}

// dQ pass (thread per query); also writes delta[unit][t] = sum_c dO*O for the dK/dV pass
template <int D, typename TS>
__global__ __launch_bounds__(256) void attn_temporal_bwd_dq_kernel(const TS* __restrict__ qkv, const TS* __restrict__ out,
                                                                    const TS* __restrict__ dout, const float* __restrict__ lse,
                                                                    float* __restrict__ delta, TS* __restrict__ dqkv, int T,
                                                                    int J, int C, int H, float scale) {
  __shared__ __attribute__((aligned(16))) float Ks[KT * D];
  __shared__ __attribute__((aligned(16))) float Vs[KT * D];
  const int unit = blockIdx.x, h = unit % H, bj = unit / H, j = bj % J, b = bj / J;
  const int t = blockIdx.y * 256 + threadIdx.x;
  const bool active = t < T;
  const long rs3 = (long)J * 3 * C;
  const TS* qb = qkv + ((long)b * T * J + j) * 3 * C + h * D;
  const long orow = ((long)(b * T + t) * J + j) * C + h * D;
  float q[D], g[D], dq[D];
  float dl = 0.f;
#pragma unroll
  for (int c = 0; c < D; c += 4) {
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f), y = x, z = x;
    if (active) {
      x = ld4(qb + (long)t * rs3 + c);
      y = ld4(dout + orow + c);
      z = ld4(out + orow + c);
    }
    q[c] = x.x * scale; q[c + 1] = x.y * scale; q[c + 2] = x.z * scale; q[c + 3] = x.w * scale;
    g[c] = y.x; g[c + 1] = y.y; g[c + 2] = y.z; g[c + 3] = y.w;
    dl += (y.x * z.x + y.y * z.y) + (y.z * z.z + y.w * z.w);
    dq[c] = dq[c + 1] = dq[c + 2] = dq[c + 3] = 0.f;
  }
  const float L = active ? lse[(long)unit * T + t] : 0.f;
  if (active) delta[(long)unit * T + t] = dl;
  for (int k0 = 0; k0 < T; k0 += KT) {
    __syncthreads();
    load_rows_tile<D>(Ks, qb + C, rs3, k0, T, threadIdx.x);
    load_rows_tile<D>(Vs, qb + 2 * C, rs3, k0, T, threadIdx.x);
    __syncthreads();
#pragma unroll 4
    for (int kk = 0; kk < KT; ++kk) {
      if (k0 + kk >= T) break;
      float a = 0.f, dp = 0.f;
#pragma unroll
      for (int c = 0; c < D; c += 4) {
        const float4 kv = *reinterpret_cast<const float4*>(Ks + kk * D + c);
        const float4 vv = *reinterpret_cast<const float4*>(Vs + kk * D + c);
        a += (q[c] * kv.x + q[c + 1] * kv.y) + (q[c + 2] * kv.z + q[c + 3] * kv.w);
        dp += (g[c] * vv.x + g[c + 1] * vv.y) + (g[c + 2] * vv.z + g[c + 3] * vv.w);
      }
      const float dsv = expf(a - L) * (dp - dl);
#pragma unroll
      for (int c = 0; c < D; c += 4) {
        const float4 kv = *reinterpret_cast<const float4*>(Ks + kk * D + c);
        dq[c] += dsv * kv.x; dq[c + 1] += dsv * kv.y; dq[c + 2] += dsv * kv.z; dq[c + 3] += dsv * kv.w;
      }
    }
  }
  if (active) {
    TS* r = dqkv + ((long)(b * T + t) * J + j) * 3 * C + h * D;
#pragma unroll
    for (int c = 0; c < D; c += 4)
      st4(r + c, make_float4(dq[c] * scale, dq[c + 1] * scale, dq[c + 2] * scale, dq[c + 3] * scale));
  }
}

// dK/dV pass (thread per key); queries, dO, lse and delta stream through LDS
template <int D, typename TS>
__global__ __launch_bounds__(256) void attn_temporal_bwd_dkv_kernel(const TS* __restrict__ qkv, const TS* __restrict__ dout,
                                                                     const float* __restrict__ lse, const float* __restrict__ delta,
                                                                     TS* __restrict__ dqkv, int T, int J, int C, int H,
                                                                     float scale) {
  __shared__ __attribute__((aligned(16))) float Qs[KT * D];
  __shared__ __attribute__((aligned(16))) float Gs[KT * D];
  __shared__ float Ls[KT], Ds[KT];
  const int unit = blockIdx.x, h = unit % H, bj = unit / H, j = bj % J, b = bj / J;
  const int t = blockIdx.y * 256 + threadIdx.x;   // key index
  const bool active = t < T;
  const long rs3 = (long)J * 3 * C, rs1 = (long)J * C;
  const TS* qb = qkv + ((long)b * T * J + j) * 3 * C + h * D;
  const TS* gb = dout + ((long)b * T * J + j) * C + h * D;
  float k[D], v[D], dk[D], dv[D];
#pragma unroll
  for (int c = 0; c < D; c += 4) {
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f), y = x;
    if (active) {
      x = ld4(qb + C + (long)t * rs3 + c);
      y = ld4(qb + 2 * C + (long)t * rs3 + c);
    }
    k[c] = x.x * scale; k[c + 1] = x.y * scale; k[c + 2] = x.z * scale; k[c + 3] = x.w * scale;
    v[c] = y.x; v[c + 1] = y.y; v[c + 2] = y.z; v[c + 3] = y.w;
    dk[c] = dk[c + 1] = dk[c + 2] = dk[c + 3] = 0.f;
    dv[c] = dv[c + 1] = dv[c + 2] = dv[c + 3] = 0.f;
  }
  for (int q0 = 0; q0 < T; q0 += KT) {
    __syncthreads();
    load_rows_tile<D>(Qs, qb, rs3, q0, T, threadIdx.x);
    load_rows_tile<D>(Gs, gb, rs1, q0, T, threadIdx.x);
    if (threadIdx.x < KT) {
      const int tq = q0 + threadIdx.x;
      Ls[threadIdx.x] = (tq < T) ? lse[(long)unit * T + tq] : 0.f;
      Ds[threadIdx.x] = (tq < T) ? delta[(long)unit * T + tq] : 0.f;
    }
    __syncthreads();
#pragma unroll 1
    for (int qq = 0; qq < KT; ++qq) {
      if (q0 + qq >= T) break;
      float a = 0.f, dp = 0.f;
#pragma unroll
      for (int c = 0; c < D; c += 4) {
        const float4 qv = *reinterpret_cast<const float4*>(Qs + qq * D + c);
        const float4 gv = *reinterpret_cast<const float4*>(Gs + qq * D + c);
        a += (k[c] * qv.x + k[c + 1] * qv.y) + (k[c + 2] * qv.z + k[c + 3] * qv.w);
        dp += (v[c] * gv.x + v[c + 1] * gv.y) + (v[c + 2] * gv.z + v[c + 3] * gv.w);
      }
      const float pe = expf(a - Ls[qq]);
      const float dsv = pe * (dp - Ds[qq]);
#pragma unroll
      for (int c = 0; c < D; c += 4) {
        const float4 qv = *reinterpret_cast<const float4*>(Qs + qq * D + c);
        const float4 gv = *reinterpret_cast<const float4*>(Gs + qq * D + c);
        dk[c] += dsv * qv.x; dk[c + 1] += dsv * qv.y; dk[c + 2] += dsv * qv.z; dk[c + 3] += dsv * qv.w;
        dv[c] += pe * gv.x; dv[c + 1] += pe * gv.y; dv[c + 2] += pe * gv.z; dv[c + 3] += pe * gv.w;
      }
    }
  }
  if (active) {
    TS* r = dqkv + ((long)(b * T + t) * J + j) * 3 * C + h * D;
#pragma unroll
    for (int c = 0; c < D; c += 4) {
      st4(r + C + c, make_float4(dk[c] * scale, dk[c + 1] * scale, dk[c + 2] * scale, dk[c + 3] * scale));
      st4(r + 2 * C + c, make_float4(dv[c], dv[c + 1], dv[c + 2], dv[c + 3]));
    }
  }
}

#define MP_DISPATCH_D(D, CALL)                     \
  switch (D) {                                     \
    case 4:  { constexpr int DD = 4;  CALL; } break; \
    case 8:  { constexpr int DD = 8;  CALL; } break; \
    case 16: { constexpr int DD = 16; CALL; } break; \
    case 32: { constexpr int DD = 32; CALL; } break; \
    case 64: { constexpr int DD = 64; CALL; } break; \
    default: MP_CHECK(false, MP_ERR_ARG, "temporal attention: head dim %d unsupported (4,8,16,32,64)", D); \
  }

template <typename TS>
static int temporal_fwd_t(const TS* qkv, TS* out, float* lse, int B, int T, int J, int C, int H, hipStream_t st) {
  const int D = C / H;
  dim3 grid(B * J * H, cdiv(T, 256));
  MP_DISPATCH_D(D, hipLaunchKernelGGL((attn_temporal_fwd_kernel<DD, TS>), grid, dim3(256), 0, st, qkv, out, lse, T, J, C, H, qk_scale(D)));
  MP_LAUNCH_CHECK();
  return MP_OK;
}
template <typename TS>
static int temporal_bwd_t(const TS* qkv, const TS* out, const TS* dout, const float* lse, float* delta, TS* dqkv, int B, int T, int J,
                          int C, int H, hipStream_t st) {
  const int D = C / H;
  dim3 grid(B * J * H, cdiv(T, 256));
  // k is pre-multiplied by `scale` in the dK/dV kernel and q in the dQ kernel, so that a = scale * q.k in both
  MP_DISPATCH_D(D, hipLaunchKernelGGL((attn_temporal_bwd_dq_kernel<DD, TS>), grid, dim3(256), 0, st, qkv, out, dout, lse, delta, dqkv,
                                      T, J, C, H, qk_scale(D)));
  MP_LAUNCH_CHECK();
  MP_DISPATCH_D(D, hipLaunchKernelGGL((attn_temporal_bwd_dkv_kernel<DD, TS>), grid, dim3(256), 0, st, qkv, dout, lse, delta, dqkv, T, J,
                                      C, H, qk_scale(D)));
  MP_LAUNCH_CHECK();
  return MP_OK;
}

bool attn_tmfma_supported(int T, int D);
int attn_tmfma_fwd(const bf16* qkv, bf16* out, float* lse, int B, int T, int J, int C, int H, hipStream_t st);
int attn_tmfma_bwd(const bf16* qkv, const bf16* out, const bf16* dout, const float* lse, bf16* dqkv, int B, int T, int J, int C, int H,
                   hipStream_t st);

int attn_temporal_fwd(const void* qkv, void* out, float* lse, int is_bf16, int B, int T, int J, int C, int H, hipStream_t st) {
  MP_CHECK(C % H == 0, MP_ERR_ARG, "attn_temporal_fwd: C %% H");
  if (is_bf16 && attn_tmfma_supported(T, C / H)) return attn_tmfma_fwd((const bf16*)qkv, (bf16*)out, lse, B, T, J, C, H, st);
  return is_bf16 ? temporal_fwd_t<bf16>((const bf16*)qkv, (bf16*)out, lse, B, T, J, C, H, st)
                 : temporal_fwd_t<float>((const float*)qkv, (float*)out, lse, B, T, J, C, H, st);
}
int attn_temporal_bwd(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv, int is_bf16,
                      int B, int T, int J, int C, int H, hipStream_t st) {
  MP_CHECK(C % H == 0, MP_ERR_ARG, "attn_temporal_bwd: C %% H");
  if (is_bf16 && attn_tmfma_supported(T, C / H))
    return attn_tmfma_bwd((const bf16*)qkv, (const bf16*)out, (const bf16*)dout, lse, (bf16*)dqkv, B, T, J, C, H, st);
  return is_bf16 ? temporal_bwd_t<bf16>((const bf16*)qkv, (const bf16*)out, (const bf16*)dout, lse, delta, (bf16*)dqkv, B, T, J, C, H, st)
                 : temporal_bwd_t<float>((const float*)qkv, (const float*)out, (const float*)dout, lse, delta, (float*)dqkv, B, T, J, C,
                                         H, st);
}

}  // namespace mp
