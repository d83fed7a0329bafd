// Fused Winner-Take-All multi-hypothesis training loss, forward AND gradient in one pass over the poses:
//   wloss     = mean_{b,t} min_k mean_j w_j |p_kj - y_j|                       losses.py:104-138 (+ :14-43)
//   score_reg = beta * BCE(scores, onehot(argmin_k))                            losses.py:141-170
//   vloss     = vel_w * mean_{b,k,t<T-1,j} |d_t p - d_t y|   (ALL hypotheses)   losses.py:75-101, axis = 2
//   sreg      = smooth_w * mean_{b,k,t<T-1,j,c} w_j (d_t p)^2                   regularizations.py:160-174
// train.sq_loss (`squared`): wloss = mean_{b,t} min_k mean_{j,c} w_j (p - y)^2 (losses.py:46-72,110-116) and
// vloss = vel_w * mean_{b,k,t,j,c} (d_t p - d_t y)^2 (losses.py:96-97); the other two terms do not change.
// assembled as main_h36m_lifting.py:101-209 does (the reference evaluates the WTA part twice per step and
// syncs the host 5 times; here it is one kernel + a 1-block finalize and no host sync).
// A lane per joint, three frames per wave (see wta_loss_kernel).  K = 1 with scores == nullptr is the single-hypothesis loss of ManifoldMixSTE.
#include "common.h"
#include "kernels.h"

namespace mp {

constexpr int LJ = 17;
static const float H36M_W[LJ] = {1, 1, 2.5f, 2.5f, 1, 2.5f, 2.5f, 1, 1, 1, 1.5f, 1.5f, 4, 4, 1.5f, 4, 4};   // STANDARD_H36M_WEIGHTS, losses.py:6-8
struct JointW { float w[LJ]; };        // per-joint weights of the WTA / smoothness terms, passed by value (kernel argument)

struct LossScales { float wta, bce, vel, smooth; };

__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wv] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// Wave mapping (round 6; the north star's "wavefront shuffles for the K-way argmin"): a LANE PER JOINT, three consecutive frames per wave (51 of 64
// lanes, like the decoder kernel), so that a wave's loads are the whole 204-byte rows of its frames - consecutive frames of a window are consecutive rows:
// 612 contiguous bytes per hypothesis - instead of one thread walking a row 12 bytes at a time (the thread-per-frame form of rounds 1-5 reached 6 % of the
// HBM rate on 150 workgroups).  Per hypothesis the 17 per-joint distances of a frame are summed by a shuffle tree inside the frame's lane group; the
// group's first lane takes the minimum over K and the winner goes back to the 17 lanes by a shuffle.  The K scoring terms of a frame are taken by the
// group's first K lanes.  A workgroup owns `fpb` consecutive frames (a quarter per wave, three at a time) and writes ONE partial row: the sums keep a fixed
// order (lane tree, wave order, workgroup order), two runs give the same bits.
__device__ __forceinline__ float group17_sum(float v, int j) {      // sum over the 17 lanes of a frame's group (valid in the group's lane 0)
#pragma unroll
  for (int d = 16; d >= 1; d >>= 1) {
    const float o = __shfl_down(v, d, 64);
    if (j < d && j + d < LJ) v += o;
  }
  return v;
}

__global__ __launch_bounds__(256) void wta_loss_kernel(const float* __restrict__ poses, const float* __restrict__ scores,
                                                        const float* __restrict__ y, JointW jw, int squared, LossScales sc,
                                                        float* __restrict__ partial, int* __restrict__ argmin,
                                                        float* __restrict__ dposes, float* __restrict__ dscores, int B, int K,
                                                        int T, int fpb) {
  __shared__ float red[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = lane / LJ, j = lane - slot * LJ;              // lanes 51-63: slot 3, idle
  const bool lane_on = slot < 3;
  const int per_wave = fpb >> 2;                                 // frames per wave (fpb % 4 == 0)
  const int f_begin = blockIdx.x * fpb + wave * per_wave, f_end = min(min(f_begin + per_wave, (blockIdx.x + 1) * fpb), B * T);
  const float wj = lane_on ? jw.w[j] : 0.f;
  float lw = 0.f, lb = 0.f, lv = 0.f, ls = 0.f;
  for (int f0 = f_begin; f0 < f_end; f0 += 3) {                  // (wave-uniform trip count: the shuffles below see every lane)
    const int f = f0 + slot;
    const bool valid = lane_on && f < f_end;
    const int fc = valid ? f : f_begin;                          // idle lanes read a valid frame and contribute nothing
    const int b = fc / T, t = fc - b * T;
    const float* yr = y + (long)fc * LJ * 3 + 3 * j * (lane_on ? 1 : 0);
    const float y0 = yr[0], y1 = yr[1], y2 = yr[2];
    // ---- pass 1: per-hypothesis weighted distance of this joint, summed over the frame's lane group; the winner ----
    float best = INFINITY;
    int kb = 0;
    for (int k = 0; k < K; ++k) {
      const float* pr = poses + (((long)b * K + k) * T + t) * LJ * 3 + 3 * (lane_on ? j : 0);
      const float dx = pr[0] - y0, dy = pr[1] - y1, dz = pr[2] - y2;
      const float d2 = dx * dx + dy * dy + dz * dz;
      float e = group17_sum(wj * (squared ? d2 : sqrtf(d2)), j);
      e /= squared ? (float)(LJ * 3) : (float)LJ;
      if (e < best) { best = e; kb = k; }                        // (meaningful in the group's lane 0)
    }
    const int src = min(slot, 2) * LJ;                           // the group's first lane
    kb = __shfl(kb, src, 64);
    best = __shfl(best, src, 64);
    if (valid && j == 0) {
      lw += best * sc.wta;
      if (argmin != nullptr) argmin[f] = kb;
    }
    // ---- scoring BCE (torch clamps log at -100; backward divides by max(s(1-s), 1e-12)): hypothesis k on the group's lane k ----
    if (scores != nullptr && valid && j < K) {
      const float s = scores[((long)b * K + j) * T + t];
      const float gt = (j == kb) ? 1.0f : 0.0f;
      const float l1 = fmaxf(logf(s), -100.0f), l0 = fmaxf(logf(1.0f - s), -100.0f);
      lb -= (gt * l1 + (1.0f - gt) * l0) * sc.bce;
      if (dscores != nullptr) dscores[((long)b * K + j) * T + t] = sc.bce * (s - gt) / fmaxf(s * (1.0f - s), 1e-12f);
    }
    // ---- pass 2: velocity / smoothness terms and the pose gradient of this joint, all hypotheses ----
    if (valid) {
      const bool has_prev = t > 0, has_next = t < T - 1;
      const float yn0 = has_next ? yr[LJ * 3] : 0.f, yn1 = has_next ? yr[LJ * 3 + 1] : 0.f, yn2 = has_next ? yr[LJ * 3 + 2] : 0.f;
      const float yp0 = has_prev ? yr[-LJ * 3] : 0.f, yp1 = has_prev ? yr[-LJ * 3 + 1] : 0.f, yp2 = has_prev ? yr[-LJ * 3 + 2] : 0.f;
      for (int k = 0; k < K; ++k) {
        const long fo = (((long)b * K + k) * T + t) * LJ * 3 + 3 * j;
        const float* pr = poses + fo;
        const float p0 = pr[0], p1 = pr[1], p2 = pr[2];
        float g0 = 0.f, g1 = 0.f, g2 = 0.f;
        if (k == kb) {
          const float dx = p0 - y0, dy = p1 - y1, dz = p2 - y2;
          const float n = sqrtf(dx * dx + dy * dy + dz * dz);
          if (squared) {
            const float c = sc.wta * wj * (2.0f / (float)(LJ * 3));
            g0 += c * dx; g1 += c * dy; g2 += c * dz;
          } else if (n > 0.f) {
            const float c = sc.wta * wj / ((float)LJ * n);
            g0 += c * dx; g1 += c * dy; g2 += c * dz;
          }
        }
        if (has_next) {
          const float s0 = pr[LJ * 3] - p0, s1 = pr[LJ * 3 + 1] - p1, s2 = pr[LJ * 3 + 2] - p2;
          const float u0 = s0 - (yn0 - y0), u1 = s1 - (yn1 - y1), u2 = s2 - (yn2 - y2);
          const float n2 = u0 * u0 + u1 * u1 + u2 * u2, n = sqrtf(n2);
          lv += (squared ? n2 : n) * sc.vel;
          ls += wj * (s0 * s0 + s1 * s1 + s2 * s2) * sc.smooth;
          if (squared || n > 0.f) {
            const float c = squared ? 2.0f * sc.vel : sc.vel / n;
            g0 -= c * u0; g1 -= c * u1; g2 -= c * u2;
          }
          const float c2 = 2.0f * sc.smooth * wj;
          g0 -= c2 * s0; g1 -= c2 * s1; g2 -= c2 * s2;
        }
        if (has_prev) {
          const float s0 = p0 - pr[-LJ * 3], s1 = p1 - pr[-LJ * 3 + 1], s2 = p2 - pr[-LJ * 3 + 2];
          const float u0 = s0 - (y0 - yp0), u1 = s1 - (y1 - yp1), u2 = s2 - (y2 - yp2);
          const float n = sqrtf(u0 * u0 + u1 * u1 + u2 * u2);
          if (squared || n > 0.f) {
            const float c = squared ? 2.0f * sc.vel : sc.vel / n;
            g0 += c * u0; g1 += c * u1; g2 += c * u2;
          }
          const float c2 = 2.0f * sc.smooth * wj;
          g0 += c2 * s0; g1 += c2 * s1; g2 += c2 * s2;
        }
        if (dposes != nullptr) { dposes[fo] = g0; dposes[fo + 1] = g1; dposes[fo + 2] = g2; }
      }
    }
  }
  const float a = block_sum_256(lw, red), bb = block_sum_256(lb, red), c = block_sum_256(lv, red), d = block_sum_256(ls, red);
  if (threadIdx.x == 0) {
    partial[4 * blockIdx.x] = a; partial[4 * blockIdx.x + 1] = bb; partial[4 * blockIdx.x + 2] = c; partial[4 * blockIdx.x + 3] = d;
  }
}

// one block: terms[i] = sum_p partial[p][i] for i < 4
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ partial, int P, float* __restrict__ terms,
                                                             int nterms_out, int skip_bce) {
  __shared__ float red[4];
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  for (int p = threadIdx.x; p < P; p += 256)
    for (int i = 0; i < 4; ++i) s[i] += partial[4 * p + i];
  float tot[4];
  for (int i = 0; i < 4; ++i) tot[i] = block_sum_256(s[i], red);
  if (threadIdx.x == 0) {
    if (skip_bce) { terms[0] = tot[0]; terms[1] = tot[2]; terms[2] = tot[3]; }
    else { terms[0] = tot[0]; terms[1] = tot[1]; terms[2] = tot[2]; terms[3] = tot[3]; }
  }
  (void)nterms_out;
}

static int loss_impl(const float* poses, const float* scores, const float* y, const LossCfg& cfg, float* terms, int* argmin,
                     float* dposes, float* dscores, int B, int K, int T, float* scratch, long scratch_floats, int skip_bce,
                     hipStream_t st) {
  MP_CHECK(B > 0 && K >= 1 && K <= 8 && T >= 2, MP_ERR_ARG, "wta_loss: B=%d K=%d T=%d unsupported (T >= 2, K <= 8)", B, K, T);
  // frames per workgroup: 48 (four waves x 12 frames: short-lived workgroups, 800 of them at the benchmark's 38 394 frames) when the caller's scratch holds
  // that many partial rows, else the 256 of the ABI's minimum scratch size (4 * ceil(B T / 256) floats)
  const int fpb = scratch_floats >= 4L * cdiv(B * T, 48) ? 48 : 256;
  const int grid = cdiv(B * T, fpb);
  MP_CHECK(scratch_floats >= 4L * grid, MP_ERR_ARG, "wta_loss: scratch too small");
  LossScales sc;
  sc.wta = 1.0f / ((float)B * T);
  sc.bce = (scores != nullptr) ? cfg.beta / ((float)B * K * T) : 0.f;
  sc.vel = cfg.vel_w / ((float)B * K * (T - 1) * LJ * (cfg.squared ? 3 : 1));
  sc.smooth = cfg.smooth_w / ((float)B * K * (T - 1) * LJ * 3);
  MP_CHECK(cfg.use_joint_weights >= 0 && cfg.use_joint_weights <= 2, MP_ERR_ARG, "wta_loss: w_loss %d (0 none, 1 H36M weights, 2 custom)", cfg.use_joint_weights);
  JointW jw;
  for (int j = 0; j < LJ; ++j) jw.w[j] = cfg.use_joint_weights == 1 ? H36M_W[j] : (cfg.use_joint_weights == 2 ? cfg.joint_weights[j] : 1.0f);
  hipLaunchKernelGGL(wta_loss_kernel, dim3(grid), dim3(256), 0, st, poses, scores, y, jw, cfg.squared, sc, scratch, argmin,
                     dposes, dscores, B, K, T, fpb);
  MP_LAUNCH_CHECK();
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, scratch, grid, terms, 4, skip_bce);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

int wta_loss(const float* poses, const float* scores, const float* y, const LossCfg& cfg, float* terms, int* argmin, float* dposes,
             float* dscores, int B, int K, int T, float* scratch, long scratch_floats, hipStream_t st) {
  MP_CHECK(scores != nullptr, MP_ERR_ARG, "wta_loss: scores required");
  return loss_impl(poses, scores, y, cfg, terms, argmin, dposes, dscores, B, K, T, scratch, scratch_floats, 0, st);
}

int single_loss(const float* poses, const float* y, const LossCfg& cfg, float* terms, float* dposes, int B, int T, float* scratch,
                long scratch_floats, hipStream_t st) {
  return loss_impl(poses, nullptr, y, cfg, terms, nullptr, dposes, nullptr, B, 1, T, scratch, scratch_floats, 1, st);
}

// ---------------------------------------------------------------------------------------------
// eval-time aggregation (RMCLManifoldMixSTE.aggregate, rmcl_manifold_mix_ste.py:141-185) and MPJPE
// (mean_joint_errors.py:31-36).  mode 0 weighted_ave, 1 best_score, 2 oracle (unweighted argmin vs y)
// ---------------------------------------------------------------------------------------------
__global__ void aggregate_kernel(const float* __restrict__ poses, const float* __restrict__ scores, const float* __restrict__ y,
                                 int mode, float* __restrict__ out, int B, int K, int T) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= B * T) return;
  const int b = f / T, t = f % T;
  float* o = out + (long)f * LJ * 3;
  if (mode == 0) {
    for (int i = 0; i < LJ * 3; ++i) {
      float a = 0.f;
      for (int k = 0; k < K; ++k) a += poses[(((long)b * K + k) * T + t) * LJ * 3 + i] * scores[((long)b * K + k) * T + t];
      o[i] = a;
    }
    return;
  }
  int kb = 0;
  if (mode == 1) {
    float best = -INFINITY;
    for (int k = 0; k < K; ++k) {
      const float s = scores[((long)b * K + k) * T + t];
      if (s > best) { best = s; kb = k; }
    }
  } else {
    float best = INFINITY;
    const float* yr = y + (long)f * LJ * 3;
    for (int k = 0; k < K; ++k) {
      const float* pr = poses + (((long)b * K + k) * T + t) * LJ * 3;
      float e = 0.f;
      for (int j = 0; j < LJ; ++j) {
        const float dx = pr[3 * j] - yr[3 * j], dy = pr[3 * j + 1] - yr[3 * j + 1], dz = pr[3 * j + 2] - yr[3 * j + 2];
        e += sqrtf(dx * dx + dy * dy + dz * dz);
      }
      if (e < best) { best = e; kb = k; }
    }
  }
  const float* pr = poses + (((long)b * K + kb) * T + t) * LJ * 3;
  for (int i = 0; i < LJ * 3; ++i) o[i] = pr[i];
}

int aggregate_poses(const float* poses, const float* scores, const float* y, int mode, float* out, int B, int K, int T,
                    hipStream_t st) {
  MP_CHECK(mode >= 0 && mode <= 2, MP_ERR_ARG, "aggregate_poses: mode %d (0 weighted_ave, 1 best_score, 2 oracle)", mode);
  MP_CHECK(mode == 2 ? y != nullptr : scores != nullptr, MP_ERR_ARG, "aggregate_poses: missing scores / ground truth");
  hipLaunchKernelGGL(aggregate_kernel, dim3(cdiv(B * T, 128)), dim3(128), 0, st, poses, scores, y, mode, out, B, K, T);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

__global__ __launch_bounds__(256) void mpjpe_partial_kernel(const float* __restrict__ pred, const float* __restrict__ gt, long n,
                                                             float* __restrict__ partial) {
  __shared__ float red[4];
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float dx = pred[3 * i] - gt[3 * i], dy = pred[3 * i + 1] - gt[3 * i + 1], dz = pred[3 * i + 2] - gt[3 * i + 2];
    s += sqrtf(dx * dx + dy * dy + dz * dz);
  }
  const float tot = block_sum_256(s, red);
  if (threadIdx.x == 0) {
    partial[4 * blockIdx.x] = tot;
    partial[4 * blockIdx.x + 1] = 0.f; partial[4 * blockIdx.x + 2] = 0.f; partial[4 * blockIdx.x + 3] = 0.f;
  }
}

int mpjpe_sum(const float* pred, const float* gt, long njoints, float* out_sum, float* scratch, long scratch_floats,
              hipStream_t st) {
  const int grid = (int)max(1L, min((njoints + 255) / 256, 1024L));
  MP_CHECK(scratch_floats >= 4L * grid + 4, MP_ERR_ARG, "mpjpe_sum: scratch too small");
  hipLaunchKernelGGL(mpjpe_partial_kernel, dim3(grid), dim3(256), 0, st, pred, gt, njoints, scratch);
  MP_LAUNCH_CHECK();
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, scratch, grid, scratch + 4L * grid, 4, 0);
  MP_LAUNCH_CHECK();
  MP_HIP(hipMemcpyAsync(out_sum, scratch + 4L * grid, sizeof(float), hipMemcpyDeviceToDevice, st));
  return MP_OK;
}

// ---------------------------------------------------------------------------------------------
// rigid_seg_reg term of make_loss (main_h36m_lifting.py:170-177): weight * segments_time_consistency(pred, mode="sum")
// (metrics/regularizations.py:8-45, metrics/utils.py:4-20) = weight * sum_{window, bone} Var_t(bone length), unbiased variance over the
// T frames of a window, for (B, T, 17, 3) predictions of the single-hypothesis models; forward and gradient in one kernel:
// one workgroup per window, a thread per frame; d_poses is ACCUMULATED into (the other loss terms are already there).
// ---------------------------------------------------------------------------------------------
__constant__ int c_rs_parent[LJ] = {-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 9, 8, 11, 12, 8, 14, 15};   // data/skeleton.py, bones = (j, parent_j)

__global__ __launch_bounds__(256) void rigid_segments_kernel(const float* __restrict__ poses, float weight, float* __restrict__ partial,
                                                              float* __restrict__ dposes, int T) {
  __shared__ float s_sum[4][LJ - 1], s_sq[4][LJ - 1], s_mean[LJ - 1];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float* base = poses + (long)b * T * LJ * 3;
  // frame-0 lengths as the common shift of the variance sums
  float ref[LJ - 1];
  for (int j = 1; j < LJ; ++j) {
    const int p = c_rs_parent[j];
    const float dx = base[3 * j] - base[3 * p], dy = base[3 * j + 1] - base[3 * p + 1], dz = base[3 * j + 2] - base[3 * p + 2];
    ref[j - 1] = sqrtf(dx * dx + dy * dy + dz * dz);
  }
  float a1[LJ - 1], a2[LJ - 1];
  for (int k = 0; k < LJ - 1; ++k) { a1[k] = 0.f; a2[k] = 0.f; }
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    const float* fr = base + (long)t * LJ * 3;
    for (int j = 1; j < LJ; ++j) {
      const int p = c_rs_parent[j];
      const float dx = fr[3 * j] - fr[3 * p], dy = fr[3 * j + 1] - fr[3 * p + 1], dz = fr[3 * j + 2] - fr[3 * p + 2];
      const float d = sqrtf(dx * dx + dy * dy + dz * dz) - ref[j - 1];
      a1[j - 1] += d; a2[j - 1] += d * d;
    }
  }
  for (int k = 0; k < LJ - 1; ++k) {
    const float u = wave_sum(a1[k]), v = wave_sum(a2[k]);
    if (lane == 0) { s_sum[wv][k] = u; s_sq[wv][k] = v; }
  }
  __syncthreads();
  float var_tot = 0.f;
  if (threadIdx.x < LJ - 1) {
    const int k = threadIdx.x;
    const float u = s_sum[0][k] + s_sum[1][k] + s_sum[2][k] + s_sum[3][k], v = s_sq[0][k] + s_sq[1][k] + s_sq[2][k] + s_sq[3][k];
    s_mean[k] = ref[k] + u / (float)T;
    var_tot = fmaxf(v - u * u / (float)T, 0.f) / (float)(T - 1);
  }
  var_tot = wave_sum(var_tot);               // bones 0..15 live in wave 0
  if (threadIdx.x == 0) partial[b] = weight * var_tot;
  __syncthreads();
  if (dposes == nullptr) return;
  const float c = weight * 2.0f / (float)(T - 1);
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    const float* fr = base + (long)t * LJ * 3;
    float g[LJ][3];
    for (int j = 0; j < LJ; ++j) { g[j][0] = 0.f; g[j][1] = 0.f; g[j][2] = 0.f; }
    for (int j = 1; j < LJ; ++j) {
      const int p = c_rs_parent[j];
      const float dx = fr[3 * j] - fr[3 * p], dy = fr[3 * j + 1] - fr[3 * p + 1], dz = fr[3 * j + 2] - fr[3 * p + 2];
      const float len = sqrtf(dx * dx + dy * dy + dz * dz);
      if (len > 0.f) {
        const float s = c * (len - s_mean[j - 1]) / len;
        g[j][0] += s * dx; g[j][1] += s * dy; g[j][2] += s * dz;
        g[p][0] -= s * dx; g[p][1] -= s * dy; g[p][2] -= s * dz;
      }
    }
    float* o = dposes + ((long)b * T + t) * LJ * 3;
    for (int j = 0; j < LJ; ++j) { o[3 * j] += g[j][0]; o[3 * j + 1] += g[j][1]; o[3 * j + 2] += g[j][2]; }
  }
}

__global__ __launch_bounds__(256) void rigid_finalize_kernel(const float* __restrict__ partial, int B, float* __restrict__ term) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) s += partial[i];
  const float tot = block_sum_256(s, red);
  if (threadIdx.x == 0) term[0] = tot;
}

int rigid_segments_loss(const float* poses, float weight, float* term, float* dposes, int B, int T, float* scratch, long scratch_floats,
                        hipStream_t st) {
  MP_CHECK(poses && term && scratch && B > 0 && T >= 2 && scratch_floats >= B, MP_ERR_ARG, "rigid_segments_loss: bad argument (B=%d T=%d)", B, T);
  hipLaunchKernelGGL(rigid_segments_kernel, dim3(B), dim3(256), 0, st, poses, weight, scratch, dposes, T);
  MP_LAUNCH_CHECK();
  hipLaunchKernelGGL(rigid_finalize_kernel, dim3(1), dim3(256), 0, st, scratch, B, term);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

}  // namespace mp
