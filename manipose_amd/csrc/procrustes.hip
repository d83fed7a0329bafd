// Procrustes-aligned errors (the alignment variants): P-MPJPE ("Protocol #2",
// hpe/mh_so3_hpe/metrics/mean_joint_errors.py:148-189) and the 'procrustes' alignment of keypoint_3d_pck / keypoint_3d_auc
// (metrics/pck.py:5-60,127-131).  Both solve, per frame, for the similarity transform (scale a, proper rotation R, translation t)
// taking the predicted joints onto the target ones in the least-squares sense; the reference does it with a batched numpy SVD on
// the host.  Here one thread per frame uses Horn's closed form: with H = sum_j y0_j x0_j^T (centred prediction / target) the
// optimal PROPER rotation is the unit quaternion that maximises q^T N(H) q, N the symmetric 4x4 matrix of Horn (1987), and the
// maximum eigenvalue equals trace(S) of the SVD with the reflection fix (the sign flip of the last singular value the reference
// applies) - so scale and rotation come from the dominant eigenpair of N, found by cyclic Jacobi sweeps in registers.
// Outputs per frame: the aligned per-joint errors reduced like pose_metrics (sum of errors, PCK / AUC counts).
#include "common.h"
#include "kernels.h"

namespace mp {

constexpr int PR_J = 17;

// dominant eigenpair of a symmetric 4x4 matrix (cyclic Jacobi, 8 sweeps: converged to fp32 round-off for these matrices)
__device__ __forceinline__ void eig4_max(float (&a)[4][4], float (&q)[4], float& lam) {
  float v[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
#pragma unroll 1
  for (int sweep = 0; sweep < 8; ++sweep) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int r = p + 1; r < 4; ++r) {
        const float apq = a[p][r];
        if (fabsf(apq) > 1e-30f) {
          const float theta = (a[r][r] - a[p][p]) / (2.0f * apq);
          const float t = copysignf(1.0f, theta) / (fabsf(theta) + sqrtf(theta * theta + 1.0f));
          const float c = 1.0f / sqrtf(t * t + 1.0f), s = t * c;
#pragma unroll
          for (int k = 0; k < 4; ++k) {          // A <- A J (columns p, r)
            const float akp = a[k][p], akr = a[k][r];
            a[k][p] = c * akp - s * akr;
            a[k][r] = s * akp + c * akr;
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {          // A <- J^T A (rows p, r)
            const float apk = a[p][k], ark = a[r][k];
            a[p][k] = c * apk - s * ark;
            a[r][k] = s * apk + c * ark;
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {          // V <- V J
            const float vkp = v[k][p], vkr = v[k][r];
            v[k][p] = c * vkp - s * vkr;
            v[k][r] = s * vkp + c * vkr;
          }
        }
      }
  }
  int best = 0;
  lam = a[0][0];
#pragma unroll
  for (int k = 1; k < 4; ++k)
    if (a[k][k] > lam) { lam = a[k][k]; best = k; }
#pragma unroll
  for (int k = 0; k < 4; ++k) q[k] = best == 0 ? v[k][0] : best == 1 ? v[k][1] : best == 2 ? v[k][2] : v[k][3];
}

struct PrArgs {
  const float* pred; const float* gt; const unsigned char* mask;     // (N, 17, 3) contiguous; mask (N, 17) or null
  long N;
  float pred_scale, gt_scale, pck_thr, auc_step;
  int auc_n, scaled;                                                  // scaled = 1: similarity transform (scale fitted); 0: rigid
};

// out partial row per block: [sum ||e||, #(e < thr), sum AUC counts, #visible, frames]
__global__ __launch_bounds__(256) void procrustes_kernel(PrArgs a, float* __restrict__ partial) {
  __shared__ float red[4][5];
  const long f = (long)blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const bool live = f < a.N;
  float s_e = 0.f, s_pck = 0.f, s_auc = 0.f, s_vis = 0.f;
  if (live) {
    float x[PR_J][3], y[PR_J][3];            // x: target, y: prediction (the reference's X / Y)
    float mx[3] = {0, 0, 0}, my[3] = {0, 0, 0};
#pragma unroll
    for (int j = 0; j < PR_J; ++j)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        x[j][c] = a.gt_scale * a.gt[(f * PR_J + j) * 3 + c];
        y[j][c] = a.pred_scale * a.pred[(f * PR_J + j) * 3 + c];
        mx[c] += x[j][c]; my[c] += y[j][c];
      }
#pragma unroll
    for (int c = 0; c < 3; ++c) { mx[c] *= 1.0f / PR_J; my[c] *= 1.0f / PR_J; }
    float h[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};      // h[r][c] = sum_j y0_j[r] x0_j[c]  (prediction x target)
    float nx = 0.f, ny = 0.f;
#pragma unroll
    for (int j = 0; j < PR_J; ++j) {
      float x0[3], y0[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) { x0[c] = x[j][c] - mx[c]; y0[c] = y[j][c] - my[c]; nx += x0[c] * x0[c]; ny += y0[c] * y0[c]; }
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) h[r][c] += y0[r] * x0[c];
    }
    // Horn's N for the rotation taking y0 onto x0 (quaternion w, x, y, z)
    float n[4][4];
    n[0][0] = h[0][0] + h[1][1] + h[2][2];
    n[0][1] = n[1][0] = h[1][2] - h[2][1];
    n[0][2] = n[2][0] = h[2][0] - h[0][2];
    n[0][3] = n[3][0] = h[0][1] - h[1][0];
    n[1][1] = h[0][0] - h[1][1] - h[2][2];
    n[1][2] = n[2][1] = h[0][1] + h[1][0];
    n[1][3] = n[3][1] = h[2][0] + h[0][2];
    n[2][2] = -h[0][0] + h[1][1] - h[2][2];
    n[2][3] = n[3][2] = h[1][2] + h[2][1];
    n[3][3] = -h[0][0] - h[1][1] + h[2][2];
    float q[4], lam;
    eig4_max(n, q, lam);
    const float qn = 1.0f / sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const float w = q[0] * qn, qx = q[1] * qn, qy = q[2] * qn, qz = q[3] * qn;
    float R[3][3];                            // x0 ~ R y0
    R[0][0] = 1 - 2 * (qy * qy + qz * qz); R[0][1] = 2 * (qx * qy - w * qz);     R[0][2] = 2 * (qx * qz + w * qy);
    R[1][0] = 2 * (qx * qy + w * qz);     R[1][1] = 1 - 2 * (qx * qx + qz * qz); R[1][2] = 2 * (qy * qz - w * qx);
    R[2][0] = 2 * (qx * qz - w * qy);     R[2][1] = 2 * (qy * qz + w * qx);     R[2][2] = 1 - 2 * (qx * qx + qy * qy);
    // scale: trace(S) normX / normY with the normalised H of the reference == lam / sum |y0|^2 for the unnormalised one
    const float sc = a.scaled ? lam / ny : 1.0f;
    (void)nx;
#pragma unroll
    for (int j = 0; j < PR_J; ++j) {
      float e2 = 0.f;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const float y0r = sc * (R[r][0] * (y[j][0] - my[0]) + R[r][1] * (y[j][1] - my[1]) + R[r][2] * (y[j][2] - my[2])) + mx[r];
        const float d = y0r - x[j][r];
        e2 += d * d;
      }
      const float e = sqrtf(e2);
      s_e += e;
      const bool vis = a.mask == nullptr || a.mask[f * PR_J + j] != 0;
      if (vis) {
        s_vis += 1.f;
        s_pck += (e < a.pck_thr) ? 1.f : 0.f;
        s_auc += fmaxf(0.f, (float)(a.auc_n - 1) - floorf(e / a.auc_step));
      }
    }
  }
  const float vals[5] = {s_e, s_pck, s_auc, s_vis, live ? 1.f : 0.f};
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const float s = wave_sum(vals[k]);
    if (lane == 0) red[wv][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < 5) partial[(long)blockIdx.x * 5 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ void procrustes_finalize_kernel(const float* __restrict__ partial, int blocks, float* __restrict__ out) {
  if (threadIdx.x < 5) {
    double s = 0.0;
    for (int b = 0; b < blocks; ++b) s += (double)partial[(long)b * 5 + threadIdx.x];
    out[threadIdx.x] = (float)s;
  }
}

int procrustes_errors(const float* pred, const float* gt, const unsigned char* mask, long N, int J, float pred_scale, float gt_scale,
                      float pck_thr, float auc_max, int auc_n, int scaled, float* out, float* scratch, long scratch_floats, hipStream_t st) {
  MP_CHECK(pred && gt && out && scratch, MP_ERR_ARG, "procrustes_errors: null pointer");
  MP_CHECK(J == PR_J && N > 0, MP_ERR_ARG, "procrustes_errors: J=%d N=%ld (17 joints)", J, N);
  MP_CHECK(auc_n >= 2 && auc_max > 0.f, MP_ERR_ARG, "procrustes_errors: bad AUC grid");
  const int blocks = (int)cdiv(N, 256L);
  MP_CHECK(scratch_floats >= 5L * blocks, MP_ERR_ARG, "procrustes_errors: scratch too small");
  PrArgs a = {pred, gt, mask, N, pred_scale, gt_scale, pck_thr, auc_max / (float)(auc_n - 1), auc_n, scaled};
  hipLaunchKernelGGL(procrustes_kernel, dim3(blocks), dim3(256), 0, st, a, scratch);
  MP_LAUNCH_CHECK();
  hipLaunchKernelGGL(procrustes_finalize_kernel, dim3(1), dim3(64), 0, st, scratch, blocks, out);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

}  // namespace mp
