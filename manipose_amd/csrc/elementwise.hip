// HBM-bound row kernels of the MixSTE backbone: LayerNorm fwd/bwd (wave per token row), the two input
// embeddings, positional-embedding gradients, DropPath masks and the fused Adam update.
#include <stdarg.h>
#include "common.h"
#include "kernels.h"

namespace mp {

// ---------------------------------------------------------------------------------------------
// error string
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* last_error() { return g_err; }

// ---------------------------------------------------------------------------------------------
// LayerNorm forward.  mix_ste.py:353-358 (norm1/norm2, eps 1e-6), :143,154,166,170 (shared
// Spatial_norm / Temporal_norm), :149 (x += Temporal_pos_embed fused behind the spatial post-norm).
// One wave per token row, row held in registers (C <= 1024), two-pass mean/variance like ATen.
// ---------------------------------------------------------------------------------------------
constexpr int LN_MAXV = 4;  // float4 per lane -> C <= 1024

template <int V>
__device__ __forceinline__ void row_stats(const float4 (&v)[V], int lane, int C, float eps, float& mean, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < V; ++i)
    if (lane * 4 + 256 * i < C) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < V; ++i)
    if (lane * 4 + 256 * i < C) {
      const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (c * c + d * d);
    }
  rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
}

// V float4 per lane (C <= 256 V); gamma / beta of both stages live in registers for the whole kernel
// B16 (f16f8 output only): also write the plain bf16 copy y2_b16 (operand forms 1 / 2 without the fp16 backward).  A template parameter: without the
// copy's pointer arithmetic the f16f8 form needs 100 instead of 104 VGPRs - 5 instead of 4 waves per SIMD for the default form (3), which has no copy.
template <typename T, int V, bool B16 = false>
__global__ __launch_bounds__(256) void ln_fwd_kernel(LnFwdArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int C = a.C;
  float4 g1[V], b1[V], g2[V], b2[V];
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int c = lane * 4 + 256 * i;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    g1[i] = (a.g1 != nullptr && c < C) ? ld4(a.g1 + c) : z;
    b1[i] = (a.g1 != nullptr && c < C) ? ld4(a.b1 + c) : z;
    g2[i] = (a.g2 != nullptr && c < C) ? ld4(a.g2 + c) : z;
    b2[i] = (a.g2 != nullptr && c < C) ? ld4(a.b2 + c) : z;
  }
  auto load_row = [&](int m, float4 (&v)[V]) {
    const float* xr = a.x + (long)m * C;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int c = lane * 4 + 256 * i;
      if (c < C) v[i] = ld4(xr + c);
    }
  };
  auto finish_row = [&](int m, float4 (&v)[V]) {
    if (a.g1 != nullptr) {
      float mean, rstd;
      row_stats<V>(v, lane, C, a.eps1, mean, rstd);
      const float* pr = (a.pos != nullptr) ? a.pos + (long)((m / a.J) % a.T) * C : nullptr;
#pragma unroll
      for (int i = 0; i < V; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < C) {
          float4 o;
          o.x = (v[i].x - mean) * rstd * g1[i].x + b1[i].x;
          o.y = (v[i].y - mean) * rstd * g1[i].y + b1[i].y;
          o.z = (v[i].z - mean) * rstd * g1[i].z + b1[i].z;
          o.w = (v[i].w - mean) * rstd * g1[i].w + b1[i].w;
          if (pr != nullptr) {
            const float4 p = ld4(pr + c);
            o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
          }
          v[i] = o;
          if (a.x1 != nullptr) st4(a.x1 + (long)m * C + c, o);      // null: the consumer recomputes it from x and stats1
        }
      }
      if (lane == 0) {
        a.stats1[2 * (long)m] = mean;
        a.stats1[2 * (long)m + 1] = rstd;
      }
    }
    if (a.g2 != nullptr) {
      float mean, rstd;
      row_stats<V>(v, lane, C, a.eps2, mean, rstd);
      T* yr = reinterpret_cast<T*>(a.y2) + (long)m * C;
      const long lo_off = (a.y2_lo != nullptr && !__is_same(T, f16f8)) ? reinterpret_cast<T*>(a.y2_lo) - reinterpret_cast<T*>(a.y2) : 0;   // planar output only
#pragma unroll
      for (int i = 0; i < V; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < C) {
          float4 o;
          o.x = (v[i].x - mean) * rstd * g2[i].x + b2[i].x;
          o.y = (v[i].y - mean) * rstd * g2[i].y + b2[i].y;
          o.z = (v[i].z - mean) * rstd * g2[i].z + b2[i].z;
          o.w = (v[i].w - mean) * rstd * g2[i].w + b2[i].w;
          if constexpr (sizeof(T) == 2 && __is_same(T, f16f8)) {
            st4_f16f8(yr + c, reinterpret_cast<char*>(a.y2_lo) + ((long)m * C + c) * 2, o, false);
            if constexpr (B16) st4(reinterpret_cast<bf16*>(a.y2_b16) + (long)m * C + c, o);
          } else st4(yr + c, o, lo_off);
        }
      }
      if (lane == 0) {
        a.stats2[2 * (long)m] = mean;
        a.stats2[2 * (long)m + 1] = rstd;
      }
    }
  };
  // two rows in flight per wave: the second row's loads are outstanding while the first row is reduced and stored
  for (int m = wave; m < a.M; m += 2 * nwaves) {
    float4 va[V], vb[V];
    const int mb = m + nwaves;
    load_row(m, va);
    if (mb < a.M) load_row(mb, vb);
    finish_row(m, va);
    if (mb < a.M) finish_row(mb, vb);
  }
}

// one row per wave, no grid-stride loop: short-lived workgroups stream measurably faster than a persistent grid on this chip
// (tools/probes/hbm_stream.hip: LayerNorm-shaped pass 5.9-6.2 TB/s against 4.8-5.1 TB/s for 2048 workgroups looping over the rows)
static int row_grid(int M) { return max(1, cdiv(M, 4)); }

int ln_fwd(const LnFwdArgs& a_in, int out_mode, hipStream_t st) {
  LnFwdArgs a = a_in;
  MP_CHECK(a.C % 4 == 0 && a.C <= 256 * LN_MAXV, MP_ERR_ARG, "ln_fwd: C=%d must be a multiple of 4 and <= 1024", a.C);
  MP_CHECK(a.g1 != nullptr || a.g2 != nullptr, MP_ERR_ARG, "ln_fwd: no stage requested");
  MP_CHECK(out_mode < 2 || a.g2 == nullptr || a.y2_lo != nullptr, MP_ERR_ARG, "ln_fwd: planar output without its lo plane");
  MP_CHECK(out_mode != 3 || a.g2 == nullptr || a.C % 64 == 0, MP_ERR_ARG, "ln_fwd: f16f8 output needs C %% 64 == 0");
  if (out_mode < 2) a.y2_lo = nullptr;
#define MP_LN_FWD(TT, V) hipLaunchKernelGGL((ln_fwd_kernel<TT, V>), dim3(row_grid(a.M)), dim3(256), 0, st, a)
#define MP_LN_FWD_V(V) do { if (out_mode == 3 && a.y2_b16 != nullptr) hipLaunchKernelGGL((ln_fwd_kernel<f16f8, V, true>), dim3(row_grid(a.M)), dim3(256), 0, st, a); \
    else if (out_mode == 3) MP_LN_FWD(f16f8, V); else if (out_mode == 2) MP_LN_FWD(bf16p, V); else if (out_mode == 1) MP_LN_FWD(bf16, V); else MP_LN_FWD(float, V); } while (0)
  if (a.C <= 256)      MP_LN_FWD_V(1);
  else if (a.C <= 512) MP_LN_FWD_V(2);
  else                 MP_LN_FWD_V(4);
#undef MP_LN_FWD_V
#undef MP_LN_FWD
  MP_LAUNCH_CHECK();
  return MP_OK;
}

// ---------------------------------------------------------------------------------------------
// LayerNorm backward: dx = [dskip +] rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat * xhat)),
// dxhat = dy * gamma.  dgamma/dbeta: per-wave register sums -> per-block LDS sum -> partial rows
// in scratch -> reduce_partials_kernel (deterministic, no atomics).
// ---------------------------------------------------------------------------------------------
// saturation / non-finite counters of the scaled-fp16 gradient stores (common.h sat_f16x4): 4 floats behind the scale (mp_model::gsc)
__device__ __forceinline__ unsigned* gs_cnt(const float* gsc) { return reinterpret_cast<unsigned*>(const_cast<float*>(gsc)) + 4; }
#ifndef LNB_GRID_N
#define LNB_GRID_N 1024
#endif
constexpr int LNB_GRID = LNB_GRID_N;
// (Round 4: walking the rows last-written-first, so that a kernel starts on what its producer left in the 256 MB Infinity Cache, moves nothing:
// LayerNorm class 27.65 / 27.73 -> 27.57 / 27.58 ms with every kernel on one queue, profiles/r04_probes/ab_step3.log.)   // 4 workgroups (16 waves) per CU

// V float4 per lane (C <= 256 V); R rows in flight per wave: every load of the R rows (x, dy, skip gradient, statistics, DropPath
// scale) is issued before the first row is reduced.
template <typename TDY, int V, int R>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const TDY* __restrict__ dy, const float* __restrict__ x,
                                                      const float* __restrict__ stats, const float* __restrict__ gamma,
                                                      const float* dskip, float* dx, bf16* __restrict__ dx_b16,
                                                      const float* __restrict__ mask, int mask_mode, int T, int J,
                                                      float* __restrict__ partial, int M, int C, float rs, const float* __restrict__ dys_p,
                                                      const float* __restrict__ b16_gs_p) {
  __shared__ float red[4 * 2 * 256 * V];  // [wave][dgamma|dbeta][C <= 256 V]
  const float dys = dys_p != nullptr ? *dys_p : 1.0f, b16_gs = b16_gs_p != nullptr ? *b16_gs_p : 0.f;      // this backward's gradient scale (engine: grad_scale_kernel)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  float4 dg[V], db[V], gm[V];
#pragma unroll
  for (int i = 0; i < V; ++i) {
    dg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    db[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int c = lane * 4 + 256 * i;
    gm[i] = (c < C) ? ld4(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int m0 = wave; m0 < M; m0 += R * nwaves) {
    float4 xv[R][V], g[R][V], k[R][V];
    float mean[R], rstd[R], ms[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int m = m0 + r * nwaves;
      if (m < M) {
        mean[r] = lone(stats[2 * (long)m]);
        rstd[r] = lone(stats[2 * (long)m + 1]);
        ms[r] = (dx_b16 != nullptr) ? droppath_scale(mask, mask_mode, __builtin_amdgcn_readfirstlane(m), T, J) : 1.0f;
#pragma unroll
        for (int i = 0; i < V; ++i) {
          const int c = lane * 4 + 256 * i;
          if (c < C) {
            xv[r][i] = ld4(x + (long)m * C + c);
            g[r][i] = ld4(dy + (long)m * C + c);
            k[r][i] = (dskip != nullptr) ? ld4(dskip + (long)m * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int m = m0 + r * nwaves;
      if (m >= M) break;
      float4 xh[V], d[V];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < V; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < C) {
          const float4 gg = make_float4(g[r][i].x * dys, g[r][i].y * dys, g[r][i].z * dys, g[r][i].w * dys);      // dys: 1, or the inverse of the scale dy carries
          xh[i] = make_float4((xv[r][i].x - mean[r]) * rstd[r], (xv[r][i].y - mean[r]) * rstd[r], (xv[r][i].z - mean[r]) * rstd[r],
                              (xv[r][i].w - mean[r]) * rstd[r]);
          dg[i].x += gg.x * xh[i].x; dg[i].y += gg.y * xh[i].y; dg[i].z += gg.z * xh[i].z; dg[i].w += gg.w * xh[i].w;
          db[i].x += gg.x; db[i].y += gg.y; db[i].z += gg.z; db[i].w += gg.w;
          d[i] = make_float4(gg.x * gm[i].x, gg.y * gm[i].y, gg.z * gm[i].z, gg.w * gm[i].w);
          s1 += (d[i].x + d[i].y) + (d[i].z + d[i].w);
          s2 += (d[i].x * xh[i].x + d[i].y * xh[i].y) + (d[i].z * xh[i].z + d[i].w * xh[i].w);
        }
      }
      s1 = wave_sum(s1) / (float)C;
      s2 = wave_sum(s2) / (float)C;
#pragma unroll
      for (int i = 0; i < V; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < C) {
          float4 o;
          o.x = rstd[r] * (d[i].x - s1 - xh[i].x * s2);
          o.y = rstd[r] * (d[i].y - s1 - xh[i].y * s2);
          o.z = rstd[r] * (d[i].z - s1 - xh[i].z * s2);
          o.w = rstd[r] * (d[i].w - s1 - xh[i].w * s2);
          if (dskip != nullptr) { o.x += rs * k[r][i].x; o.y += rs * k[r][i].y; o.z += rs * k[r][i].z; o.w += rs * k[r][i].w; }
          st4(dx + (long)m * C + c, o);
          if (dx_b16 != nullptr) {   // 2-byte copy, pre-scaled by the consumer branch's DropPath mask: A operand of its GEMMs - bf16, or (b16_gs != 0) fp16 of b16_gs x value
            if (b16_gs != 0.f) st4_f16(dx_b16 + (long)m * C + c, o, ms[r] * b16_gs, gs_cnt(b16_gs_p));
            else st4(dx_b16 + (long)m * C + c, make_float4(o.x * ms[r], o.y * ms[r], o.z * ms[r], o.w * ms[r]));
          }
        }
      }
    }
  }
  // block reduction of dgamma/dbeta
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int c = lane * 4 + 256 * i;
    if (c < C) {
      *reinterpret_cast<float4*>(&red[(wv * 2 + 0) * 256 * V + c]) = dg[i];
      *reinterpret_cast<float4*>(&red[(wv * 2 + 1) * 256 * V + c]) = db[i];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    const int which = i / C, c = i - which * C;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) s += red[(w * 2 + which) * 256 * V + c];
    partial[(long)blockIdx.x * 2 * C + i] = s;
  }
}

// out segment j (<=4): dst[j][i] += sum_p partial[p][off_j + i]
struct ReduceDst { float* dst[4]; int off[5]; int stride[4]; };
// block = RP_OUT outputs x RP_GRP row groups: each thread sums P / RP_GRP partial rows (4 independent chains), LDS-combines the groups.
// 16 x 16 (instead of 32 x 8) doubles the workgroups (the partials are L2-resident; the kernel is latency-, not bandwidth-bound).
constexpr int RP_OUT = 16, RP_GRP = 16;
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial, int P, int n, ReduceDst d) {
  __shared__ float red[RP_GRP][RP_OUT + 1];
  const int oi = threadIdx.x % RP_OUT, grp = threadIdx.x / RP_OUT;
  const int i = blockIdx.x * RP_OUT + oi;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < n) {
    int p = grp;
    for (; p + 3 * RP_GRP < P; p += 4 * RP_GRP) {
      s0 += partial[(long)p * n + i];
      s1 += partial[(long)(p + RP_GRP) * n + i];
      s2 += partial[(long)(p + 2 * RP_GRP) * n + i];
      s3 += partial[(long)(p + 3 * RP_GRP) * n + i];
    }
    for (; p < P; p += RP_GRP) s0 += partial[(long)p * n + i];
  }
  red[grp][oi] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (grp == 0 && i < n) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < RP_GRP; ++k) s += red[k][oi];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (i >= d.off[j] && i < d.off[j + 1] && d.dst[j] != nullptr) d.dst[j][(long)(i - d.off[j]) * d.stride[j]] += s;
  }
}

// The reduction of the per-workgroup partial sums into the parameter gradients is needed by nobody downstream on `st`: with
// (st_param, ev) given it runs on st_param behind `ev`, recorded here on `st` after the kernel that wrote the partials (the caller
// owns `scratch` until it has run).  Replaces `st` by the stream to launch on; false on a HIP error.
static bool param_stream(hipStream_t& st, hipStream_t st_param, hipEvent_t ev) {
  if (st_param == nullptr || ev == nullptr) return true;
  if (hipEventRecord(ev, st) != hipSuccess || hipStreamWaitEvent(st_param, ev, 0) != hipSuccess) {
    set_error("ln_bwd: could not order the partial-sum reduction on the parameter stream");
    return false;
  }
  st = st_param;
  return true;
}

int ln_bwd(const void* dy, int dy_bf16, const float* x, const float* stats, const float* gamma, const float* dskip, float* dx, void* dx_b16,
           const float* mask, int mask_mode, int T, int J, float* dgamma, float* dbeta, int M, int C, float* scratch,
           long scratch_floats, hipStream_t st, hipStream_t st_param, hipEvent_t ev, float rs, const float* dy_scale, const float* b16_gs) {
  MP_CHECK(C % 4 == 0 && C <= 1024, MP_ERR_ARG, "ln_bwd: C=%d unsupported", C);
#ifndef LNB_R
#define LNB_R 1      // rows in flight per wave: 499 -> 490 us isolated with one (92 VGPRs, the 1024 workgroups of the scratch rows all resident); profiles/r04_probes/ln_bwd_variants2.log
#endif
  // persistent grid = the workgroups resident at once (occupancy query x CUs, at most LNB_GRID: the partial-sum scratch has LNB_GRID rows)
  static int slots[2] = {0, 0};
  if (C <= 512 && slots[dy_bf16 ? 1 : 0] == 0) {
    int dev = 0, cus = 0, per_cu = 0;
    const void* fn = dy_bf16 ? (const void*)ln_bwd_kernel<bf16, 2, LNB_R> : (const void*)ln_bwd_kernel<float, 2, LNB_R>;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, 0) != hipSuccess || cus <= 0 || per_cu <= 0) {
      (void)hipGetLastError();
      slots[dy_bf16 ? 1 : 0] = LNB_GRID;
    } else slots[dy_bf16 ? 1 : 0] = min(cus * per_cu, LNB_GRID);
  }
  const int grid = max(1, min(cdiv(M, 4), C <= 512 ? slots[dy_bf16 ? 1 : 0] : LNB_GRID));
  MP_CHECK(scratch_floats >= (long)grid * 2 * C, MP_ERR_ARG, "ln_bwd: scratch too small");
#define MP_LN_BWD(TDY, V, R)                                                                                                             \
  hipLaunchKernelGGL((ln_bwd_kernel<TDY, V, R>), dim3(grid), dim3(256), 0, st, (const TDY*)dy, x, stats, gamma, dskip, dx, (bf16*)dx_b16, \
                     mask, mask ? mask_mode : 0, T, J, scratch, M, C, rs, dy_scale, b16_gs)
  if (C <= 512) { if (dy_bf16) MP_LN_BWD(bf16, 2, LNB_R); else MP_LN_BWD(float, 2, LNB_R); }
  else          { if (dy_bf16) MP_LN_BWD(bf16, 4, 1); else MP_LN_BWD(float, 4, 1); }
#undef MP_LN_BWD
  MP_LAUNCH_CHECK();
  ReduceDst d = {{dgamma, dbeta, nullptr, nullptr}, {0, C, 2 * C, 2 * C, 2 * C}, {1, 1, 1, 1}};
  if (!param_stream(st, st_param, ev)) return MP_ERR_HIP;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(cdiv(2 * C, RP_OUT)), dim3(256), 0, st, scratch, grid, 2 * C, d);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

#ifndef LNB2_R
#define LNB2_R 1      // rows in flight per wave.  Round 4 (tools/probes/ln_bwd_probe.hip, profiles/r04_probes/ln_bwd_variants1.log): isolated at the bench's token
#endif                // count 576 us with two rows (168 VGPRs, 3 waves per SIMD), 530 us with one (ln_bwd_kernel: 528 us for the same 16 B per element); in the step the
                      // LayerNorm class 28.17 -> 27.57 ms with every kernel on one queue, 168.0 -> 166.1 ms per step (same box, alternating, two rounds)
// Fused pair of LayerNorm backwards across a block boundary (precision-independent, C <= 512):
//   t  = dskip + LN1'(dy1; x1, stats1, gamma1)        (norm1 of block l+1, plus the residual skip gradient)
//   dx = LN0'(t; x0, stats0, gamma0)                  (shared post-norm behind block l)
// saving one fp32 write + read of the gradient stream per block.  Partials: [dgamma1 | dbeta1 | dgamma0 | dbeta0].
// R rows in flight per wave: every load of the R rows is issued before the first row is reduced.  (Round 3 measured one row per wave slower - on a
// fixed grid of 1024 workgroups; with the grid taken from the occupancy query below the register-lighter one-row form puts more waves on a CU and wins.)
template <typename TDY, int R>
__global__ __launch_bounds__(256) void ln_bwd2_kernel(const TDY* __restrict__ dy1,
                                                       const float* __restrict__ stats1, const float* __restrict__ gamma1,
                                                       const float* dskip, const float* __restrict__ x0,
                                                       const float* __restrict__ stats0, const float* __restrict__ gamma0,
                                                       const float* __restrict__ beta0, float* dx,
                                                       bf16* __restrict__ dx_b16, const float* __restrict__ mask, int mask_mode, int T,
                                                       int J, float* __restrict__ partial, int M, int C, float rs, const float* __restrict__ dys_p,
                                                       const float* __restrict__ b16_gs_p) {
  constexpr int V = 2;
  const float dys = dys_p != nullptr ? *dys_p : 1.0f, b16_gs = b16_gs_p != nullptr ? *b16_gs_p : 0.f;
  __shared__ float red[4 * 4 * 512];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  float4 acc[4][V], g1[V], g0[V], b0[V];
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int c = lane * 4 + 256 * i;
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k][i] = make_float4(0.f, 0.f, 0.f, 0.f);
    g1[i] = (c < C) ? ld4(gamma1 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    g0[i] = (c < C) ? ld4(gamma0 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    b0[i] = (c < C) ? ld4(beta0 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int m0 = wave; m0 < M; m0 += R * nwaves) {
    // every input of the R rows is requested up front
    float mean1[R], rstd1[R], mean0[R], rstd0[R], ms[R];
    float4 gy[R][V], kk[R][V], xv0[R][V];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int m = m0 + r * nwaves;
      if (m < M) {
        mean1[r] = lone(stats1[2 * (long)m]); rstd1[r] = lone(stats1[2 * (long)m + 1]);
        mean0[r] = lone(stats0[2 * (long)m]); rstd0[r] = lone(stats0[2 * (long)m + 1]);
        ms[r] = (dx_b16 != nullptr) ? droppath_scale(mask, mask_mode, __builtin_amdgcn_readfirstlane(m), T, J) : 1.0f;
#pragma unroll
        for (int i = 0; i < V; ++i) {
          const int c = lane * 4 + 256 * i;
          if (c < C) {
            gy[r][i] = ld4(dy1 + (long)m * C + c);
            kk[r][i] = ld4(dskip + (long)m * C + c);
            xv0[r][i] = ld4(x0 + (long)m * C + c);
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int m = m0 + r * nwaves;
      if (m >= M) break;
      float4 xh[V], d[V], t[V];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < V; ++i) {
        const int c = lane * 4 + 256 * i;
        xh[i] = d[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < C) {
          // x1 is the post-norm output of x0 (no positional table behind the blocks this kernel serves): recomputed with the forward's
          // expression (ln_fwd stage 1) instead of read back - one fp32 activation read less per row
          float4 xv;
          xv.x = (xv0[r][i].x - mean0[r]) * rstd0[r] * g0[i].x + b0[i].x;
          xv.y = (xv0[r][i].y - mean0[r]) * rstd0[r] * g0[i].y + b0[i].y;
          xv.z = (xv0[r][i].z - mean0[r]) * rstd0[r] * g0[i].z + b0[i].z;
          xv.w = (xv0[r][i].w - mean0[r]) * rstd0[r] * g0[i].w + b0[i].w;
          const float4 g = make_float4(gy[r][i].x * dys, gy[r][i].y * dys, gy[r][i].z * dys, gy[r][i].w * dys);      // dys: 1, or the inverse of the scale dy1 carries
          xh[i] = make_float4((xv.x - mean1[r]) * rstd1[r], (xv.y - mean1[r]) * rstd1[r], (xv.z - mean1[r]) * rstd1[r], (xv.w - mean1[r]) * rstd1[r]);
          acc[0][i].x += g.x * xh[i].x; acc[0][i].y += g.y * xh[i].y; acc[0][i].z += g.z * xh[i].z; acc[0][i].w += g.w * xh[i].w;
          acc[1][i].x += g.x; acc[1][i].y += g.y; acc[1][i].z += g.z; acc[1][i].w += g.w;
          d[i] = make_float4(g.x * g1[i].x, g.y * g1[i].y, g.z * g1[i].z, g.w * g1[i].w);
          s1 += (d[i].x + d[i].y) + (d[i].z + d[i].w);
          s2 += (d[i].x * xh[i].x + d[i].y * xh[i].y) + (d[i].z * xh[i].z + d[i].w * xh[i].w);
        }
      }
      s1 = wave_sum(s1) / (float)C;
      s2 = wave_sum(s2) / (float)C;
      float u1 = 0.f, u2 = 0.f;
#pragma unroll
      for (int i = 0; i < V; ++i) {
        const int c = lane * 4 + 256 * i;
        t[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < C) {
          const float4 k = kk[r][i], xv = xv0[r][i];
          t[i].x = rstd1[r] * (d[i].x - s1 - xh[i].x * s2) + rs * k.x;
          t[i].y = rstd1[r] * (d[i].y - s1 - xh[i].y * s2) + rs * k.y;
          t[i].z = rstd1[r] * (d[i].z - s1 - xh[i].z * s2) + rs * k.z;
          t[i].w = rstd1[r] * (d[i].w - s1 - xh[i].w * s2) + rs * k.w;
          xh[i] = make_float4((xv.x - mean0[r]) * rstd0[r], (xv.y - mean0[r]) * rstd0[r], (xv.z - mean0[r]) * rstd0[r], (xv.w - mean0[r]) * rstd0[r]);
          acc[2][i].x += t[i].x * xh[i].x; acc[2][i].y += t[i].y * xh[i].y; acc[2][i].z += t[i].z * xh[i].z; acc[2][i].w += t[i].w * xh[i].w;
          acc[3][i].x += t[i].x; acc[3][i].y += t[i].y; acc[3][i].z += t[i].z; acc[3][i].w += t[i].w;
          d[i] = make_float4(t[i].x * g0[i].x, t[i].y * g0[i].y, t[i].z * g0[i].z, t[i].w * g0[i].w);
          u1 += (d[i].x + d[i].y) + (d[i].z + d[i].w);
          u2 += (d[i].x * xh[i].x + d[i].y * xh[i].y) + (d[i].z * xh[i].z + d[i].w * xh[i].w);
        }
      }
      u1 = wave_sum(u1) / (float)C;
      u2 = wave_sum(u2) / (float)C;
#pragma unroll
      for (int i = 0; i < V; ++i) {
        const int c = lane * 4 + 256 * i;
        if (c < C) {
          const float4 o = make_float4(rstd0[r] * (d[i].x - u1 - xh[i].x * u2), rstd0[r] * (d[i].y - u1 - xh[i].y * u2),
                                       rstd0[r] * (d[i].z - u1 - xh[i].z * u2), rstd0[r] * (d[i].w - u1 - xh[i].w * u2));
          st4(dx + (long)m * C + c, o);
          if (dx_b16 != nullptr) {
            if (b16_gs != 0.f) st4_f16(dx_b16 + (long)m * C + c, o, ms[r] * b16_gs, gs_cnt(b16_gs_p));
            else st4(dx_b16 + (long)m * C + c, make_float4(o.x * ms[r], o.y * ms[r], o.z * ms[r], o.w * ms[r]));
          }
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int c = lane * 4 + 256 * i;
    if (c < C) {
#pragma unroll
      for (int k = 0; k < 4; ++k) *reinterpret_cast<float4*>(&red[(wv * 4 + k) * 512 + c]) = acc[k][i];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 4 * C; i += 256) {
    const int which = i / C, c = i - which * C;
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) sum += red[(w * 4 + which) * 512 + c];
    partial[(long)blockIdx.x * 4 * C + i] = sum;
  }
}

int ln_bwd2(const void* dy1, int dy_bf16, const float* x1, const float* stats1, const float* gamma1, const float* dskip,
            const float* x0, const float* stats0, const float* gamma0, const float* beta0, float* dx, void* dx_b16, const float* mask,
            int mask_mode, int T,
            int J, float* dgamma1, float* dbeta1, float* dgamma0, float* dbeta0, int M, int C, float* scratch, long scratch_floats,
            hipStream_t st, hipStream_t st_param, hipEvent_t ev, float rs, const float* dy_scale, const float* b16_gs) {
  MP_CHECK(C % 4 == 0 && C <= 512, MP_ERR_ARG, "ln_bwd2: C=%d unsupported", C);
  MP_CHECK(beta0 != nullptr, MP_ERR_ARG, "ln_bwd2: x1 is recomputed from x0, beta0 is required");
  (void)x1;      // (kept in the signature: the stored block input of the blocks that have one; the kernel recomputes it from x0)
  // persistent grid = the workgroups that are resident at once (168 VGPRs: 3 waves per SIMD, 3 workgroups per CU).  With LNB_GRID = 1024
  // workgroups on 768 slots the launch ran one full round and a second one at a third of the occupancy: 720 us where ln_bwd_kernel (128
  // VGPRs, 1024 slots) moves the same bytes in 470 us.
  static int slots[2] = {0, 0};
  if (slots[dy_bf16 ? 1 : 0] == 0) {
    int dev = 0, cus = 0, per_cu = 0;
    const void* fn = dy_bf16 ? (const void*)ln_bwd2_kernel<bf16, LNB2_R> : (const void*)ln_bwd2_kernel<float, LNB2_R>;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, 0) != hipSuccess || cus <= 0 || per_cu <= 0) {
      (void)hipGetLastError();
      slots[dy_bf16 ? 1 : 0] = LNB_GRID;
    } else slots[dy_bf16 ? 1 : 0] = min(cus * per_cu, LNB_GRID);
  }
#ifdef LNB2_GRID
  const int grid = max(1, min(cdiv(M, 4), LNB2_GRID));      // probe builds (tools/probes/ln_bwd_probe.hip)
#else
  const int grid = max(1, min(cdiv(M, 4), slots[dy_bf16 ? 1 : 0]));
#endif
  MP_CHECK(scratch_floats >= (long)grid * 4 * C, MP_ERR_ARG, "ln_bwd2: scratch too small");
  if (dy_bf16)
    hipLaunchKernelGGL((ln_bwd2_kernel<bf16, LNB2_R>), dim3(grid), dim3(256), 0, st, (const bf16*)dy1, stats1, gamma1, dskip, x0, stats0, gamma0,
                       beta0, dx, (bf16*)dx_b16, mask, mask ? mask_mode : 0, T, J, scratch, M, C, rs, dy_scale, b16_gs);
  else
    hipLaunchKernelGGL((ln_bwd2_kernel<float, LNB2_R>), dim3(grid), dim3(256), 0, st, (const float*)dy1, stats1, gamma1, dskip, x0, stats0, gamma0,
                       beta0, dx, (bf16*)dx_b16, mask, mask ? mask_mode : 0, T, J, scratch, M, C, rs, dy_scale, b16_gs);
  MP_LAUNCH_CHECK();
  ReduceDst d = {{dgamma1, dbeta1, dgamma0, dbeta0}, {0, C, 2 * C, 3 * C, 4 * C}, {1, 1, 1, 1}};
  if (!param_stream(st, st_param, ev)) return MP_ERR_HIP;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(cdiv(4 * C, RP_OUT)), dim3(256), 0, st, scratch, grid, 4 * C, d);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

// ---------------------------------------------------------------------------------------------
// Gradient scale of one backward (the layers whose backward GEMMs run on fp16 operands): S = the power of two that brings the largest
// element of a reference gradient - the engine passes the residual gradient at the top of the backbone - into [2^11, 2^12); every gradient
// operand of those layers is carried as fp16 of S x value, so the loss may be normalised any way (a summed loss gives gradients ~1e7 times
// those of a mean over a 79-window batch; a fixed scale overflowed fp16 on the former).  Where the window sits was MEASURED (round 4,
// tests/test_gpu_parity.py test_fp16_backward_gradient_window_at_full_width_and_after_training: dz / dqkv / the residual-gradient copy at
// the far end of the chain, against the same operands of a bf16 backward): with S from max |d_poses| at 1 the largest interior element was
// 0.7 but 2.6 % of the non-zero dz elements were fp16-subnormal at random init, and after 200 optimisation steps 38 % - the WTA loss
// gradient is a unit vector per joint whatever the error, its maximum says nothing about the interior.  Hence the reference tensor, and
// 11 binades of lift: with the reference maximum in [2^11, 2^12) dz / dqkv of the last block peak at 0.06-0.25 of it at random init and
// at 0.002 of it after those 200 steps (the gradient thins out down the chain once the net has fitted its batch), where 8 binades of lift
// still left 2.9 % of dz's true values subnormal.  2^4 of headroom remains above the reference maximum (fp16 ends at 2^16); stores
// saturate there and are counted, they never write inf (common.h sat_f16x4).  gsc[0] = S, gsc[1] = 1 / S, gsc[2] = scratch (bits of the maximum), gsc[3] = 1, gsc[4], gsc[5]
// = counters (unsigned) of this backward's fp16 gradient stores that hit the +-65504 clamp / met a non-finite value (mp_model_grad_health).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void grad_amax_kernel(const float* __restrict__ a, long na, const float* __restrict__ b, long nb, unsigned* __restrict__ out) {
  float mx = 0.f;
  const long stride = (long)gridDim.x * blockDim.x, i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (long i = i0; i < na; i += stride) mx = fmaxf(mx, fabsf(a[i]));
  for (long i = i0; i < nb; i += stride) mx = fmaxf(mx, fabsf(b[i]));
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(mx));      // non-negative floats order like their bit patterns
}
constexpr int GS_LIFT = 11;     // binades between 1 and where the largest incoming gradient is placed
__global__ void grad_scale_kernel(float* gsc) {
  const unsigned bits = reinterpret_cast<const unsigned*>(gsc)[2];
  int be = (int)((bits >> 23) & 0xffu);                 // biased exponent of the maximum; 0 (all gradients zero or denormal): no scaling
  be = bits == 0u ? 127 : min(max(be, 127 - 60), 127 + 60);
  gsc[0] = __uint_as_float((unsigned)(254 - be + GS_LIFT) << 23);
  gsc[1] = __uint_as_float((unsigned)(be - GS_LIFT) << 23);
  gsc[3] = 1.0f;
  gsc[4] = 0.f; gsc[5] = 0.f;      // this backward's saturation / non-finite counters (unsigned, common.h sat_f16x4)
}
int grad_scale(const float* d_poses, long n_poses, const float* d_scores, long n_scores, float* gsc, hipStream_t st) {
  if (hipMemsetAsync(gsc + 2, 0, sizeof(float), st) != hipSuccess) { set_error("grad_scale: memset failed"); return MP_ERR_HIP; }
  hipLaunchKernelGGL(grad_amax_kernel, dim3(256), dim3(256), 0, st, d_poses, n_poses, d_scores, d_scores ? n_scores : 0L, reinterpret_cast<unsigned*>(gsc + 2));
  MP_LAUNCH_CHECK();
  hipLaunchKernelGGL(grad_scale_kernel, dim3(1), dim3(1), 0, st, gsc);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

// mp_model_grad_health_async: {S, 1 / S, scratch, 1, clamped (u32), non-finite (u32)} -> {S, clamped, non-finite, 1 / S} as floats, on the device
__global__ void grad_health_pack_kernel(const float* __restrict__ gsc, float* __restrict__ out) {
  const unsigned* c = reinterpret_cast<const unsigned*>(gsc + 4);
  out[0] = gsc[0]; out[1] = (float)c[0]; out[2] = (float)c[1]; out[3] = gsc[1];
}
int grad_health_pack(const float* gsc, float* out4, hipStream_t st) {
  hipLaunchKernelGGL(grad_health_pack_kernel, dim3(1), dim3(1), 0, st, gsc, out4);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

// ---------------------------------------------------------------------------------------------
// rotation backbone input embedding: Linear(2, C) + Spatial_pos_embed (mix_ste.py:134-138)
// ---------------------------------------------------------------------------------------------
__global__ void embed_fwd_kernel(const float* __restrict__ xin, const float* __restrict__ W, const float* __restrict__ b,
                                 const float* __restrict__ spos, float* __restrict__ out, int M, int C, int J) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // one float4 of channels
  const int c4 = C / 4;
  if (i >= (long)M * c4) return;
  const int m = (int)(i / c4), c = (int)(i % c4) * 4;
  const float x0 = lone(xin[2 * (long)m]), x1 = lone(xin[2 * (long)m + 1]);
  const float4 w01 = ld4(W + 2 * c), w23 = ld4(W + 2 * c + 4);   // W[c][0..1] interleaved
  const float4 bb = ld4(b + c), p = ld4(spos + (long)(m % J) * C + c);
  float4 o;
  o.x = w01.x * x0 + w01.y * x1 + bb.x + p.x;
  o.y = w01.z * x0 + w01.w * x1 + bb.y + p.y;
  o.z = w23.x * x0 + w23.y * x1 + bb.z + p.z;
  o.w = w23.z * x0 + w23.w * x1 + bb.w + p.w;
  st4(out + (long)m * C + c, o);
}

int embed_fwd(const float* xin, const float* W, const float* b, const float* spos, float* out, int M, int C, int J,
              hipStream_t st) {
  MP_CHECK(C % 4 == 0, MP_ERR_ARG, "embed_fwd: C %% 4");
  const long n = (long)M * (C / 4);
  hipLaunchKernelGGL(embed_fwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, xin, W, b, spos, out, M, C, J);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

// dW[c][i] = sum_m g[m][c] x[m][i]; db[c] = sum_m g[m][c]; dspos[j][c] = sum_{m % J == j} g[m][c]
// grid (cdiv(C,256), chunks): thread = channel, rows of a chunk walked frame by frame.
constexpr int EMB_CHUNKS = 128;
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ g, const float* __restrict__ xin,
                                                         float* __restrict__ partial, int M, int C, int J) {
  extern __shared__ float sp[];  // [J][256] spatial-pos partial sums
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int frames = M / J;
  const int per = cdiv(frames, gridDim.y);
  const int f0 = blockIdx.y * per, f1 = min(frames, f0 + per);
  for (int j = 0; j < J; ++j) sp[j * 256 + threadIdx.x] = 0.f;
  float w0 = 0.f, w1 = 0.f, bs = 0.f;
  if (c < C) {
    for (int f = f0; f < f1; ++f) {
      for (int j = 0; j < J; ++j) {
        const long m = (long)f * J + j;
        const float v = g[m * C + c];
        w0 += v * xin[2 * m];
        w1 += v * xin[2 * m + 1];
        bs += v;
        sp[j * 256 + threadIdx.x] += v;
      }
    }
    const long n = (long)(3 + J) * C;
    float* pr = partial + (long)blockIdx.y * n;
    pr[2 * c] = w0;
    pr[2 * c + 1] = w1;
    pr[2 * C + c] = bs;
    for (int j = 0; j < J; ++j) pr[3 * C + (long)j * C + c] = sp[j * 256 + threadIdx.x];
  }
}

// The same sums with a thread = four channels and every load of a frame (JT joint rows) in flight before the first is used: the block is
// 128 float4 columns x 2 frame groups, the spatial-position sums stay in registers, the two groups are added through LDS.  (The
// thread-per-channel kernel above walks one dependent 4-byte load at a time: 402 us for the 668 MB of the bench size.)
constexpr int EMB4_CHUNKS = 256;
template <int JT>
__global__ __launch_bounds__(256) void embed_bwd4_kernel(const float* __restrict__ g, const float* __restrict__ xin, float* __restrict__ partial, int M, int C) {
  extern __shared__ float4 sh4[];    // [3 + JT][128]
  const int c4 = threadIdx.x & 127, grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 7);
  const int c = (blockIdx.x * 128 + c4) * 4;
  const bool ok = c < C;
  const int frames = M / JT;
  const int per = cdiv(frames, gridDim.y);
  const int f0 = blockIdx.y * per, f1 = min(frames, f0 + per);
  float4 w0 = make_float4(0.f, 0.f, 0.f, 0.f), w1 = w0, bs = w0, sp[JT];
#pragma unroll
  for (int j = 0; j < JT; ++j) sp[j] = w0;
  for (int f = f0 + grp; f < f1; f += 2) {
    float4 v[JT];
    const float* gr = g + (long)f * JT * C + (ok ? c : 0);
#pragma unroll
    for (int j = 0; j < JT; ++j) v[j] = ld4(gr + (long)j * C);
#pragma unroll
    for (int j = 0; j < JT; ++j) v[j] = lone4(v[j]);  // each component is splat over (w0, w1) below: common.h, lone()
    const float* xr = xin + (long)f * JT * 2;         // wave-uniform
#pragma unroll
    for (int j = 0; j < JT; ++j) {
      const float x0 = lone(xr[2 * j]), x1 = lone(xr[2 * j + 1]);
      w0.x += v[j].x * x0; w0.y += v[j].y * x0; w0.z += v[j].z * x0; w0.w += v[j].w * x0;
      w1.x += v[j].x * x1; w1.y += v[j].y * x1; w1.z += v[j].z * x1; w1.w += v[j].w * x1;
      bs.x += v[j].x; bs.y += v[j].y; bs.z += v[j].z; bs.w += v[j].w;
      sp[j].x += v[j].x; sp[j].y += v[j].y; sp[j].z += v[j].z; sp[j].w += v[j].w;
    }
  }
  if (grp == 1) {
    sh4[0 * 128 + c4] = w0; sh4[1 * 128 + c4] = w1; sh4[2 * 128 + c4] = bs;
#pragma unroll
    for (int j = 0; j < JT; ++j) sh4[(3 + j) * 128 + c4] = sp[j];
  }
  __syncthreads();
  if (grp == 0 && ok) {
    auto add = [](float4 a, const float4& b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); };
    w0 = add(w0, sh4[0 * 128 + c4]); w1 = add(w1, sh4[1 * 128 + c4]); bs = add(bs, sh4[2 * 128 + c4]);
    const long n = (long)(3 + JT) * C;
    float* pr = partial + (long)blockIdx.y * n;
    st4(pr + 2 * c, make_float4(w0.x, w1.x, w0.y, w1.y));            // dW is [C][2]
    st4(pr + 2 * c + 4, make_float4(w0.z, w1.z, w0.w, w1.w));
    st4(pr + 2 * C + c, bs);
#pragma unroll
    for (int j = 0; j < JT; ++j) st4(pr + 3 * C + (long)j * C + c, add(sp[j], sh4[(3 + j) * 128 + c4]));
  }
}

long embed_bwd_scratch_floats(int C, int J) { return (long)EMB4_CHUNKS * (3 + J) * C; }

int embed_bwd(const float* g, const float* xin, float* dW, float* db, float* dspos, int M, int C, int J, float* scratch,
              long scratch_floats, hipStream_t st) {
  if (J == 17 && C % 4 == 0 && M % J == 0) {
    const int chunks = max(1, min(EMB4_CHUNKS, M / J / 2));
    const int n = (3 + J) * C;
    MP_CHECK(scratch_floats >= (long)chunks * n, MP_ERR_ARG, "embed_bwd: scratch too small");
    hipLaunchKernelGGL(embed_bwd4_kernel<17>, dim3(cdiv(C / 4, 128), chunks), dim3(256), (3 + 17) * 128 * sizeof(float4), st, g, xin, scratch, M, C);
    MP_LAUNCH_CHECK();
    ReduceDst d = {{dW, db, dspos, nullptr}, {0, 2 * C, 3 * C, n, n}, {1, 1, 1, 1}};
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(cdiv(n, RP_OUT)), dim3(256), 0, st, scratch, chunks, n, d);
    MP_LAUNCH_CHECK();
    return MP_OK;
  }
  const int chunks = max(1, min(EMB_CHUNKS, M / J));
  const int n = (3 + J) * C;
  MP_CHECK(scratch_floats >= (long)chunks * n, MP_ERR_ARG, "embed_bwd: scratch too small");
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(cdiv(C, 256), chunks), dim3(256), J * 256 * sizeof(float), st, g, xin, scratch,
                     M, C, J);
  MP_LAUNCH_CHECK();
  ReduceDst d = {{dW, db, dspos, nullptr}, {0, 2 * C, 3 * C, n, n}, {1, 1, 1, 1}};
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(cdiv(n, RP_OUT)), dim3(256), 0, st, scratch, chunks, n, d);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

// ---------------------------------------------------------------------------------------------
// bones net input: Linear(J*2 = IN, S*Cs = O) + Spatial_pos_embed flattened (manifold_mix_ste.py:133-150)
// ---------------------------------------------------------------------------------------------
constexpr int BE_FRAMES = 32, BE_IN = 34;      // frames per workgroup: every weight element is loaded once per 32 frames
__global__ __launch_bounds__(256) void bones_embed_fwd_kernel(const float* __restrict__ xin, const float* __restrict__ W,
                                                               const float* __restrict__ b, const float* __restrict__ spos,
                                                               float* __restrict__ out, int BT, int O) {
  __shared__ float xs[BE_FRAMES * BE_IN];
  const int f0 = blockIdx.y * BE_FRAMES;
  for (int i = threadIdx.x; i < BE_FRAMES * BE_IN; i += 256) {
    const int f = f0 + i / BE_IN;
    xs[i] = (f < BT) ? xin[(long)f0 * BE_IN + i] : 0.f;
  }
  __syncthreads();
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= O) return;
  float acc[BE_FRAMES];
  const float base = b[o] + spos[o];
#pragma unroll
  for (int f = 0; f < BE_FRAMES; ++f) acc[f] = base;
  for (int i = 0; i < BE_IN; ++i) {
    const float w = lone(W[(long)o * BE_IN + i]);
#pragma unroll
    for (int f = 0; f < BE_FRAMES; ++f) acc[f] += w * xs[f * BE_IN + i];
  }
#pragma unroll
  for (int f = 0; f < BE_FRAMES; ++f)
    if (f0 + f < BT) out[(long)(f0 + f) * O + o] = acc[f];
}

int bones_embed_fwd(const float* xin, const float* W, const float* b, const float* spos, float* out, int BT, int IN, int O,
                    hipStream_t st) {
  MP_CHECK(IN == BE_IN, MP_ERR_ARG, "bones_embed_fwd: in_features %d != 34 (17 joints x 2)", IN);
  hipLaunchKernelGGL(bones_embed_fwd_kernel, dim3(cdiv(O, 256), cdiv(BT, BE_FRAMES)), dim3(256), 0, st, xin, W, b, spos, out,
                     BT, O);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

constexpr int BEB_CHUNKS = 32;
__global__ __launch_bounds__(256) void bones_embed_bwd_kernel(const float* __restrict__ g, const float* __restrict__ xin,
                                                               float* __restrict__ partial, int BT, int O) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  const int per = cdiv(BT, gridDim.y);
  const int f0 = blockIdx.y * per, f1 = min(BT, f0 + per);
  if (o >= O) return;
  float acc[BE_IN + 1];
#pragma unroll
  for (int i = 0; i <= BE_IN; ++i) acc[i] = 0.f;
  for (int f = f0; f < f1; ++f) {
    const float v = g[(long)f * O + o];
    const float* xr = xin + (long)f * BE_IN;   // wave-uniform address
#pragma unroll
    for (int i = 0; i < BE_IN; ++i) acc[i] += v * xr[i];
    acc[BE_IN] += v;
  }
  float* pr = partial + (long)blockIdx.y * O * (BE_IN + 1);
#pragma unroll
  for (int i = 0; i < BE_IN; ++i) pr[(long)o * BE_IN + i] = acc[i];
  pr[(long)O * BE_IN + o] = acc[BE_IN];
}

__global__ void add_vec_kernel(const float* __restrict__ src, float* __restrict__ dst, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] += src[i];
}

int bones_embed_bwd(const float* g, const float* xin, float* dW, float* db, float* dspos, int BT, int IN, int O,
                    float* scratch, long scratch_floats, hipStream_t st) {
  MP_CHECK(IN == BE_IN, MP_ERR_ARG, "bones_embed_bwd: in_features %d != 34", IN);
  const int chunks = max(1, min(BEB_CHUNKS, BT));
  const int n = O * (BE_IN + 1);
  MP_CHECK(scratch_floats >= (long)(chunks + 1) * n, MP_ERR_ARG, "bones_embed_bwd: scratch too small");
  hipLaunchKernelGGL(bones_embed_bwd_kernel, dim3(cdiv(O, 256), chunks), dim3(256), 0, st, g, xin, scratch, BT, O);
  MP_LAUNCH_CHECK();
  // the positional table is added exactly like the bias (index o = s*Cs + c), so dspos == db contribution
  ReduceDst d = {{dW, db, nullptr, nullptr}, {0, O * BE_IN, n, n, n}, {1, 1, 1, 1}};
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(cdiv(n, RP_OUT)), dim3(256), 0, st, scratch, chunks, n, d);
  MP_LAUNCH_CHECK();
  ReduceDst d2 = {{nullptr, dspos, nullptr, nullptr}, {0, O * BE_IN, n, n, n}, {1, 1, 1, 1}};
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(cdiv(n, RP_OUT)), dim3(256), 0, st, scratch, chunks, n, d2);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

// ---------------------------------------------------------------------------------------------
// dTemporal_pos_embed[t][c] += sum_{b,j} g[(b,t,j)][c]   (mix_ste.py:149)
// ---------------------------------------------------------------------------------------------
// block = (frame t, 256 channels): 64 float4 columns x 4 row groups; every thread sums its share of the B*J rows of that frame with
// four independent accumulators (the loads of a serial chain were the whole cost), the groups are combined through LDS in a fixed order.
__global__ __launch_bounds__(256) void tpos_grad_kernel(const float* __restrict__ g, float* __restrict__ dtpos, int B, int T, int J, int C) {
  __shared__ float4 red[4][64];
  const int t = blockIdx.x, c4 = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int c = blockIdx.y * 256 + c4 * 4;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
  if (c < C) {
    // rows (b, j) of this frame in steps of 4 per group, walked with a running (b, j) (J >= 4): no integer division
    int b = 0, j = grp;
    auto step = [&](float4& acc, bool& more) {              // token of the running (b, j), then advance by 4 rows
      more = b < B;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (more) v = ld4(g + ((long)(b * T + t) * J + j) * C + c);
      j += 4;
      if (j >= J) { j -= J; ++b; }
      return v;
    };
    bool more = true;
    while (more) {                                           // four loads in flight, four independent sums (fixed order)
      bool m0, m1, m2, m3;
      const float4 v0 = step(a0, m0), v1 = step(a1, m1), v2 = step(a2, m2), v3 = step(a3, m3);
      a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
      a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
      a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
      a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
      more = m3;
    }
  }
  red[grp][c4] = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z),
                             (a0.w + a1.w) + (a2.w + a3.w));
  __syncthreads();
  if (grp == 0 && c < C) {
    float4 o = ld4(dtpos + (long)t * C + c);
#pragma unroll
    for (int k = 0; k < 4; ++k) { o.x += red[k][c4].x; o.y += red[k][c4].y; o.z += red[k][c4].z; o.w += red[k][c4].w; }
    st4(dtpos + (long)t * C + c, o);
  }
}

int tpos_grad(const float* g, float* dtpos, int B, int T, int J, int C, hipStream_t st) {
  MP_CHECK(C % 4 == 0, MP_ERR_ARG, "tpos_grad: C %% 4");
  hipLaunchKernelGGL(tpos_grad_kernel, dim3(T, cdiv(C, 256)), dim3(256), 0, st, g, dtpos, B, T, J, C);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

__global__ void scale_copy_kernel(float* __restrict__ dst, const float* __restrict__ src, float s, long n, int accumulate) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = accumulate ? dst[i] + s * src[i] : s * src[i];
}
int scale_copy(float* dst, const float* src, float s, long n, hipStream_t st) {
  hipLaunchKernelGGL(scale_copy_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, dst, src, s, n, 0);
  MP_LAUNCH_CHECK();
  return MP_OK;
}
int axpy_scaled(float* dst, const float* src, float s, long n, hipStream_t st) {
  hipLaunchKernelGGL(scale_copy_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, dst, src, s, n, 1);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

// out[m][:] = mask(m) * g[m][:]  (DropPath backward on a branch gradient)
template <typename TO>
__global__ void scale_rows_kernel(const float* __restrict__ g, const float* __restrict__ mask, int mode, TO* __restrict__ out,
                                  int M, int C, int T, int J) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c4 = C / 4;
  if (i >= (long)M * c4) return;
  const int m = (int)(i / c4);
  const float s = droppath_scale(mask, mode, m, T, J);
  float4 v = ld4(g + i * 4);
  v.x *= s; v.y *= s; v.z *= s; v.w *= s;
  st4(out + i * 4, v);
}

int scale_rows(const float* g, const float* mask, int mask_mode, void* out, int out_bf16, int M, int C, int T, int J, hipStream_t st) {
  const long n = (long)M * (C / 4);
  if (out_bf16)
    hipLaunchKernelGGL(scale_rows_kernel<bf16>, dim3(cdiv(n, 256)), dim3(256), 0, st, g, mask, mask_mode, (bf16*)out, M, C, T, J);
  else
    hipLaunchKernelGGL(scale_rows_kernel<float>, dim3(cdiv(n, 256)), dim3(256), 0, st, g, mask, mask_mode, (float*)out, M, C, T, J);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

// ---------------------------------------------------------------------------------------------
// Adam with L2 weight decay == torch.optim.Adam(lr, weight_decay) (main_h36m_lifting.py:234-238)
// ---------------------------------------------------------------------------------------------
// MULT: per-element learning-rate and weight-decay multipliers (mup.optim.MuAdam: matrix-like parameters train with lr / width_mult
// and weight_decay * width_mult, main_h36m_lifting.py:227-232)
template <bool MULT>
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            long n, float lr_bc1, float inv_sqrt_bc2, float beta1, float beta2, float eps, float wd,
                            float gscale, const float* __restrict__ lr_mult, const float* __restrict__ wd_mult) {
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    float4 pp = ld4(p + i), gg = ld4(g + i), mm = ld4(m + i), vv = ld4(v + i);
    float4 lm = make_float4(1.f, 1.f, 1.f, 1.f), wm = lm;
    if (MULT) { lm = ld4(lr_mult + i); wm = ld4(wd_mult + i); }
    float* P = &pp.x; float* G = &gg.x; float* Mo = &mm.x; float* V = &vv.x; float* LM = &lm.x; float* WM = &wm.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gr = G[k] * gscale + (MULT ? wd * WM[k] : wd) * P[k];
      Mo[k] = beta1 * Mo[k] + (1.f - beta1) * gr;
      V[k] = beta2 * V[k] + (1.f - beta2) * gr * gr;
      P[k] -= (MULT ? lr_bc1 * LM[k] : lr_bc1) * (Mo[k] / (sqrtf(V[k]) * inv_sqrt_bc2 + eps));
    }
    st4(p + i, pp); st4(m + i, mm); st4(v + i, vv);
  } else {
    for (; i < n; ++i) {
      const float gr = g[i] * gscale + (MULT ? wd * wd_mult[i] : wd) * p[i];
      m[i] = beta1 * m[i] + (1.f - beta1) * gr;
      v[i] = beta2 * v[i] + (1.f - beta2) * gr * gr;
      p[i] -= (MULT ? lr_bc1 * lr_mult[i] : lr_bc1) * (m[i] / (sqrtf(v[i]) * inv_sqrt_bc2 + eps));
    }
  }
}

int adam_step(float* p, const float* g, float* m, float* v, long n, int step, float lr, float beta1, float beta2, float eps,
              float weight_decay, float grad_scale, hipStream_t st, const float* lr_mult, const float* wd_mult) {
  MP_CHECK(step >= 1, MP_ERR_ARG, "adam_step: step must be >= 1");
  MP_CHECK((lr_mult == nullptr) == (wd_mult == nullptr), MP_ERR_ARG, "adam_step: lr and weight-decay multipliers come together");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  if (lr_mult != nullptr)
    hipLaunchKernelGGL(adam_kernel<true>, dim3(cdiv(cdiv(n, 4), 256)), dim3(256), 0, st, p, g, m, v, n, (float)(lr / bc1),
                       (float)(1.0 / sqrt(bc2)), beta1, beta2, eps, weight_decay, grad_scale, lr_mult, wd_mult);
  else
  hipLaunchKernelGGL(adam_kernel<false>, dim3(cdiv(cdiv(n, 4), 256)), dim3(256), 0, st, p, g, m, v, n, (float)(lr / bc1),
                     (float)(1.0 / sqrt(bc2)), beta1, beta2, eps, weight_decay, grad_scale, lr_mult, wd_mult);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

// ---------------------------------------------------------------------------------------------
// DropPath masks (timm DropPath semantics: Bernoulli(keep) / keep per sample of dim 0), counter-based
// hash RNG so that the masks are a pure function of (seed, step, branch, sample).
// ---------------------------------------------------------------------------------------------
struct MaskDescs { MaskDesc d[48]; int n, base; };
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ void droppath_masks_kernel(float* __restrict__ masks, MaskDescs ds, unsigned long long seed,
                                      unsigned long long step) {
  if ((int)blockIdx.y >= ds.n) return;
  const MaskDesc d = ds.d[blockIdx.y];
  const int k = ds.base + blockIdx.y;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < d.count; i += gridDim.x * blockDim.x) {
    const unsigned long long r = splitmix64(splitmix64(seed ^ (step * 0xD1342543DE82EF95ull)) ^ ((unsigned long long)k << 40) ^ i);
    const float u = (float)(r >> 40) * (1.0f / 16777216.0f);
    masks[d.offset + i] = (d.keep >= 1.0f) ? 1.0f : ((u < d.keep) ? 1.0f / d.keep : 0.0f);
  }
}

int droppath_masks(float* masks, const MaskDesc* descs, int ndesc, unsigned long long seed, unsigned long long step,
                   hipStream_t st) {
  // any number of branches (4 per block of both nets: 4 (layers + layers_seg)), 48 descriptors per launch; the branch index that
  // seeds the hash is the global one
  for (int base = 0; base < ndesc; base += 48) {
    MaskDescs ds;
    ds.n = min(48, ndesc - base);
    ds.base = base;
    for (int i = 0; i < ds.n; ++i) ds.d[i] = descs[base + i];
    hipLaunchKernelGGL(droppath_masks_kernel, dim3(8, ds.n), dim3(256), 0, st, masks, ds, seed, step);
    MP_LAUNCH_CHECK();
  }
  return MP_OK;
}

}  // namespace mp
