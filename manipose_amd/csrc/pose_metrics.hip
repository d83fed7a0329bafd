// Evaluation analytics of predicted pose sequences in ONE pass over the frames: everything the
// reference computes with a dozen separate torch/numpy passes after `evaluate()` -
//   mpjpe_error / mse_error / jointwise_error        hpe/mh_so3_hpe/metrics/mean_joint_errors.py:31-80
//   segments_len_err                                  mean_joint_errors.py:83-130   (bone lengths: metrics/utils.py:4-20)
//   sagittal_symmetry(_per_bone)                      metrics/regularizations.py:96-157
//   segments_time_consistency(_per_bone)              metrics/regularizations.py:8-64  (variance over time of every bone length)
//   mean_velocity_error (eval form)                   metrics/losses.py:75-101
//   keypoint_3d_pck / keypoint_3d_auc                 metrics/pck.py:92-199 (alignment 'none' / 'scale')
// - is a function of per-frame bone lengths and per-joint errors.  One thread per frame reads the 17 x 3 predicted (and
// target) coordinates through caller-supplied element strides (the reference passes (B,3,J,L) permutations; no copy is
// made), forms the 16 bone lengths and the per-joint errors in registers and the block reduces ~120 running sums
// (DPP wave reductions, LDS across the 4 waves); a second kernel adds the per-block partial rows in a fixed order
// (deterministic, no atomics).  The time variance of a bone length is accumulated as sums of (len - len at frame 0) and its
// square: the manifold models predict constant lengths, where the plain E[x^2] - E[x]^2 form cancels catastrophically in fp32.
#include "common.h"
#include "kernels.h"

namespace mp {

// 17-joint H36M / MPI-INF-3DHP tree (data/skeleton.py:101-120 with the parents of dataset_3dhp.py:132-138), compiled in like the
// FK decoder's: bone k = (joint k+1, its parent); left/right bone pairs from joints_left = (4,5,6,11,12,13), joints_right = (1,2,3,14,15,16)
constexpr int PM_J = 17, PM_NB = 16, PM_NP = 6;
__device__ constexpr int PM_PARENT[PM_J] = {-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 9, 8, 11, 12, 8, 14, 15};
__device__ constexpr int PM_PAIR_L[PM_NP] = {3, 4, 5, 10, 11, 12}, PM_PAIR_R[PM_NP] = {0, 1, 2, 13, 14, 15};
// per-batch-item output row: [PM_SCALARS scalars | NB x 4 per bone | NP x 2 per pair | J x 2 per joint]
//   scalars: 0 sum_j ||e||   1 sum_j ||e||^2   2 sum_pairs |l - r|   3 sum_pairs (l - r)^2   4 sum_bones |gt - pred|
//            5 sum_bones (gt - pred)   6 #(||e|| < pck threshold)   7 sum_j #(AUC thresholds above ||e||)   8 #visible joints
//            9 sum_j ||d_t pred - d_t gt|| (t >= 1)   10 sum_j ||d_t pred - d_t gt||^2   11 frames counted
//   per bone: sum (len - len0), sum (len - len0)^2, sum |gt - pred|, sum (gt - pred)
//   per pair: sum |l - r|, sum (l - r)^2;  per joint: sum ||e||, sum ||e||^2
constexpr int PM_SCALARS = 12;

struct PmArgs {
  const float* pred; long ps[4];         // element strides of (b, t, j, c)
  const float* gt; long gs[4];           // nullable: prediction-only metrics
  const unsigned char* mask;             // (B, L, J) visibility or null
  int B, L, NV;
  float pred_scale, gt_scale, pck_thr, auc_step;
  int auc_n, scale_align;                // scale_align: pred *= <pred,gt>/<pred,pred> per frame (pck.py 'scale' alignment)
};

__device__ __forceinline__ void pm_load(const float* p, const long (&s)[4], int b, int t, float sc, float (&x)[PM_J][3]) {
  const float* base = p + b * s[0] + t * s[1];
#pragma unroll
  for (int j = 0; j < PM_J; ++j) {
    x[j][0] = sc * base[j * s[2]];
    x[j][1] = sc * base[j * s[2] + s[3]];
    x[j][2] = sc * base[j * s[2] + 2 * s[3]];
  }
}

// frame from an LDS image of consecutive (J x 3)-float frames (row pitch 51 dwords: odd, so a wave's 64 rows hit 64 banks)
__device__ __forceinline__ void pm_load_lds(const float* img, int row, float sc, float (&x)[PM_J][3]) {
#pragma unroll
  for (int j = 0; j < PM_J; ++j) {
    x[j][0] = sc * img[row * (3 * PM_J) + 3 * j];
    x[j][1] = sc * img[row * (3 * PM_J) + 3 * j + 1];
    x[j][2] = sc * img[row * (3 * PM_J) + 3 * j + 2];
  }
}

// FR frames (= threads) per block.  STAGED: both tensors are plain contiguous (B, L, J, 3): the block's FR frames plus the one
// before them (for the velocity terms) are copied into LDS with coalesced loads - every byte of the inputs is read from HBM
// exactly once - and each thread then takes its frame from there; otherwise every thread gathers its frame through the strides.
template <bool STAGED, int FR>
__global__ __launch_bounds__(FR) void pose_metrics_kernel(PmArgs a, float* __restrict__ partial, float* __restrict__ len0_out) {
  extern __shared__ float red[];           // [FR/64 waves][NV] + len0[NB] (+ the two frame images when STAGED)
  constexpr int NWV = FR / 64, ROW = 3 * PM_J;
  float* len0 = red + NWV * a.NV;
  float* img_p = len0 + PM_NB;
  float* img_g = img_p + (FR + 1) * ROW;
  constexpr int J = PM_J, NB = PM_NB, NP = PM_NP;
  const int b = blockIdx.y, t0 = blockIdx.x * FR, t = t0 + threadIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const bool live = t < a.L;
  if (STAGED) {
    const int first = max(t0 - 1, 0), last = min(t0 + FR, a.L);          // frames [first, last) -> image rows from (first - (t0 - 1))
    const long src = ((long)b * a.L + first) * ROW;
    const int dst = (first - (t0 - 1)) * ROW, count = (last - first) * ROW;
    for (int i = threadIdx.x; i < count; i += FR) {
      img_p[dst + i] = a.pred[src + i];
      img_g[dst + i] = a.gt[src + i];
    }
  }
  if (threadIdx.x < NB) {                  // reference length of every bone of this batch item: frame 0
    const float* base = a.pred + b * a.ps[0];
    const int j = threadIdx.x + 1, p = PM_PARENT[j];
    float s = 0.f;
    for (int c = 0; c < 3; ++c) {
      const float d = a.pred_scale * (base[j * a.ps[2] + c * a.ps[3]] - base[p * a.ps[2] + c * a.ps[3]]);
      s += d * d;
    }
    len0[threadIdx.x] = sqrtf(s);
    if (blockIdx.x == 0) len0_out[b * NB + threadIdx.x] = len0[threadIdx.x];
  }
  __syncthreads();                         // len0 (and the staged frames) visible

  float x[J][3], y[J][3];
  float plen[NB], glen[NB];
  const bool has_gt = a.gt != nullptr;
  if (live) {
    if (STAGED) {
      pm_load_lds(img_p, threadIdx.x + 1, a.pred_scale, x);
      pm_load_lds(img_g, threadIdx.x + 1, a.gt_scale, y);
    } else {
      pm_load(a.pred, a.ps, b, t, a.pred_scale, x);
      if (has_gt) pm_load(a.gt, a.gs, b, t, a.gt_scale, y);
    }
    if (has_gt && a.scale_align) {
      float pp = 0.f, pg = 0.f;
#pragma unroll
      for (int j = 0; j < J; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) { pp += x[j][c] * x[j][c]; pg += x[j][c] * y[j][c]; }
      const float f = pg / pp;
#pragma unroll
      for (int j = 0; j < J; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) x[j][c] *= f;
    }
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const int j = k + 1, p = PM_PARENT[k + 1];
      const float dx = x[j][0] - x[p][0], dy = x[j][1] - x[p][1], dz = x[j][2] - x[p][2];
      plen[k] = sqrtf(dx * dx + dy * dy + dz * dz);
      if (has_gt) {
        const float ex = y[j][0] - y[p][0], ey = y[j][1] - y[p][1], ez = y[j][2] - y[p][2];
        glen[k] = sqrtf(ex * ex + ey * ey + ez * ez);
      }
    }
  }
  // every running sum is reduced as soon as it is formed: value index v -> wave_sum -> red[wave][v]
  auto emit = [&](int v, float val) {
    const float s = wave_sum(live ? val : 0.f);
    if (lane == 0) red[wv * a.NV + v] = s;
  };
  constexpr int oB = PM_SCALARS, oP = oB + 4 * NB, oJ = oP + 2 * NP;
  float s_e = 0.f, s_e2 = 0.f, s_pck = 0.f, s_auc = 0.f, s_vis = 0.f;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    float e = 0.f, e2 = 0.f;
    if (live && has_gt) {
      const float dx = x[j][0] - y[j][0], dy = x[j][1] - y[j][1], dz = x[j][2] - y[j][2];
      e2 = dx * dx + dy * dy + dz * dz;
      e = sqrtf(e2);
      const bool vis = a.mask == nullptr || a.mask[((long)b * a.L + t) * J + j] != 0;
      if (vis) {
        s_vis += 1.f;
        s_pck += (e < a.pck_thr) ? 1.f : 0.f;
        // number of thresholds i * step (i = 0 .. auc_n-1) strictly above e
        const float q = floorf(e / a.auc_step);
        s_auc += fmaxf(0.f, (float)(a.auc_n - 1) - q);
      }
    }
    s_e += e; s_e2 += e2;
    emit(oJ + 2 * j, e);
    emit(oJ + 2 * j + 1, e2);
  }
  float s_sym = 0.f, s_sym2 = 0.f;
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const float d = live ? fabsf(plen[PM_PAIR_L[k]] - plen[PM_PAIR_R[k]]) : 0.f;
    s_sym += d; s_sym2 += d * d;
    emit(oP + 2 * k, d);
    emit(oP + 2 * k + 1, d * d);
  }
  float s_la = 0.f, s_ls = 0.f;
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    const float d0 = live ? plen[k] - len0[k] : 0.f;
    const float dl = (live && has_gt) ? glen[k] - plen[k] : 0.f;
    s_la += fabsf(dl); s_ls += dl;
    emit(oB + 4 * k, d0);
    emit(oB + 4 * k + 1, d0 * d0);
    emit(oB + 4 * k + 2, fabsf(dl));
    emit(oB + 4 * k + 3, dl);
  }
  float s_v = 0.f, s_v2 = 0.f;
  if (live && has_gt && t >= 1) {          // velocity error against frame t-1 (mean_velocity_error, axis = time)
    float xp[J][3], yp[J][3];
    if (STAGED) {
      pm_load_lds(img_p, threadIdx.x, a.pred_scale, xp);
      pm_load_lds(img_g, threadIdx.x, a.gt_scale, yp);
    } else {
      pm_load(a.pred, a.ps, b, t - 1, a.pred_scale, xp);
      pm_load(a.gt, a.gs, b, t - 1, a.gt_scale, yp);
    }
    if (a.scale_align) {                   // the aligned prediction of the previous frame
      float pp = 0.f, pg = 0.f;
#pragma unroll
      for (int j = 0; j < J; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) { pp += xp[j][c] * xp[j][c]; pg += xp[j][c] * yp[j][c]; }
      const float f = pg / pp;
#pragma unroll
      for (int j = 0; j < J; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) xp[j][c] *= f;
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const float dx = (x[j][0] - xp[j][0]) - (y[j][0] - yp[j][0]), dy = (x[j][1] - xp[j][1]) - (y[j][1] - yp[j][1]),
                  dz = (x[j][2] - xp[j][2]) - (y[j][2] - yp[j][2]);
      const float v2 = dx * dx + dy * dy + dz * dz;
      s_v += sqrtf(v2); s_v2 += v2;
    }
  }
  emit(0, s_e); emit(1, s_e2); emit(2, s_sym); emit(3, s_sym2); emit(4, s_la); emit(5, s_ls);
  emit(6, s_pck); emit(7, s_auc); emit(8, s_vis); emit(9, s_v); emit(10, s_v2); emit(11, 1.f);
  __syncthreads();
  float* prow = partial + ((long)b * gridDim.x + blockIdx.x) * a.NV;
  for (int v = threadIdx.x; v < a.NV; v += FR) {
    float sacc = red[v];
#pragma unroll
    for (int w = 1; w < NWV; ++w) sacc += red[w * a.NV + v];
    prow[v] = sacc;
  }
}

__global__ void pose_metrics_finalize_kernel(const float* __restrict__ partial, int chunks, int NV, float* __restrict__ out) {
  const int b = blockIdx.x;
  for (int v = threadIdx.x; v < NV; v += blockDim.x) {
    double s = 0.0;
    for (int c = 0; c < chunks; ++c) s += (double)partial[((long)b * chunks + c) * NV + v];
    out[(long)b * NV + v] = (float)s;
  }
}

int pose_metrics_row_floats() { return PM_SCALARS + 4 * PM_NB + 2 * PM_NP + 2 * PM_J; }

int pose_metrics(const float* pred, const long* ps, const float* gt, const long* gs, const unsigned char* mask, int B, int L, int J,
                 float pred_scale, float gt_scale, float pck_thr, float auc_max, int auc_n, int scale_align, float* out, float* len0,
                 float* scratch, long scratch_floats, hipStream_t st) {
  MP_CHECK(pred && ps && out && len0 && scratch, MP_ERR_ARG, "pose_metrics: null pointer");
  MP_CHECK(J == PM_J, MP_ERR_ARG, "pose_metrics: %d joints; the 17-joint H36M / 3DHP tree is compiled in", J);
  MP_CHECK(B > 0 && L > 0, MP_ERR_ARG, "pose_metrics: B=%d L=%d", B, L);
  MP_CHECK(gt == nullptr || gs != nullptr, MP_ERR_ARG, "pose_metrics: target without strides");
  MP_CHECK(auc_n >= 2 && auc_max > 0.f, MP_ERR_ARG, "pose_metrics: bad AUC grid");
  PmArgs a = {};
  a.pred = pred; a.gt = gt; a.mask = mask;
  for (int i = 0; i < 4; ++i) { a.ps[i] = ps[i]; a.gs[i] = gt ? gs[i] : 0; }
  a.B = B; a.L = L;
  a.NV = pose_metrics_row_floats();
  a.pred_scale = pred_scale; a.gt_scale = gt_scale; a.pck_thr = pck_thr; a.auc_step = auc_max / (float)(auc_n - 1); a.auc_n = auc_n;
  a.scale_align = scale_align;
  const long row = 3 * PM_J;
  const bool staged = gt != nullptr && ps[0] == (long)L * row && ps[1] == row && ps[2] == 3 && ps[3] == 1 && gs[0] == ps[0] &&
                      gs[1] == ps[1] && gs[2] == ps[2] && gs[3] == ps[3];
  const int fr = staged ? 128 : 256, chunks = cdiv(L, fr);
  MP_CHECK(scratch_floats >= (long)B * chunks * a.NV, MP_ERR_ARG, "pose_metrics: scratch too small (%ld < %ld)", scratch_floats,
           (long)B * chunks * a.NV);
  if (staged) {
    const size_t lds = sizeof(float) * (2 * a.NV + PM_NB + 2 * (128 + 1) * row);
    hipLaunchKernelGGL((pose_metrics_kernel<true, 128>), dim3(chunks, B), dim3(128), lds, st, a, scratch, len0);
  } else {
    const size_t lds = sizeof(float) * (4 * a.NV + PM_NB);
    hipLaunchKernelGGL((pose_metrics_kernel<false, 256>), dim3(chunks, B), dim3(256), lds, st, a, scratch, len0);
  }
  MP_LAUNCH_CHECK();
  hipLaunchKernelGGL(pose_metrics_finalize_kernel, dim3(B), dim3(128), 0, st, scratch, chunks, a.NV, out);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

// segments_len_err(mode="no_agg") (mean_joint_errors.py:83-130): the per-frame table gt - predicted bone length, (B*L, 16);
// thread per frame, inputs through element strides like pose_metrics (the reference passes (B,3,J,L) views)
__global__ void bone_length_table_kernel(const float* __restrict__ pred, const float* __restrict__ gt, long p0, long p1, long p2, long p3,
                                         long g0, long g1, long g2, long g3, int B, int L, int signed_, float* __restrict__ out) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= B * L) return;
  const int b = f / L, t = f - b * L;
  const float* pb = pred + b * p0 + t * p1;
  const float* gb = gt + b * g0 + t * g1;
#pragma unroll
  for (int k = 0; k < PM_NB; ++k) {
    const int j = k + 1, q = PM_PARENT[j];
    float lp = 0.f, lg = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float dp = pb[j * p2 + c * p3] - pb[q * p2 + c * p3], dg = gb[j * g2 + c * g3] - gb[q * g2 + c * g3];
      lp += dp * dp; lg += dg * dg;
    }
    const float d = sqrtf(lg) - sqrtf(lp);
    out[(long)f * PM_NB + k] = signed_ ? d : fabsf(d);
  }
}

int bone_length_table(const float* pred, const long* ps, const float* gt, const long* gs, int B, int L, int signed_, float* out, hipStream_t st) {
  MP_CHECK(B > 0 && L > 0, MP_ERR_ARG, "bone_length_table: B=%d L=%d", B, L);
  hipLaunchKernelGGL(bone_length_table_kernel, dim3(cdiv((long)B * L, 256)), dim3(256), 0, st, pred, gt, ps[0], ps[1], ps[2], ps[3], gs[0], gs[1],
                     gs[2], gs[3], B, L, signed_, out);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

}  // namespace mp
