// Dataset ingest: raw pose arrays of the on-disk formats -> the resident (frames, 17, 3) / (frames, 17, 2) sequences the window
// kernel (windows.hip) cuts batches from.  One pass per sequence, once per run; everything stays in HBM afterwards.
//   3-D, Human3.6M (reference: data/h36m_lifting.py:620-660 joint selection, data/utils.py:29-58 read_3d_data,
//        data/camera.py:24-28 world_to_camera, data/quaternion.py:6-31): x = qrot(qinverse(R), X[map[j]] - t), then minus the
//        transformed root joint;
//   3-D, MPI-INF-3DHP (data/dataset_3dhp.py:153-176,185-203): (X[map[j]] - X[root]) / 1000, optional valid-frame selection;
//   2-D (data/camera.py:9-14 via data/utils.py:9-26 / dataset_3dhp.py:170-175,212-224): X / w * 2 - [1, h / w]; the reference
//        subtracts a float64 list from a float32 array, i.e. the subtraction runs in double and is rounded to float32 when stored.
#include "common.h"
#include "kernels.h"

namespace mp {

struct IngestMap { int j[32]; };

#pragma clang fp contract(off)
__device__ __forceinline__ void cross3(const float a[3], const float b[3], float o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

// v + 2 (w (q x v) + q x (q x v))  -  quaternion.py:6-21
__device__ __forceinline__ void qrot3(float w, const float q[3], const float v[3], float o[3]) {
  float uv[3], uuv[3];
  cross3(q, v, uv);
  cross3(q, uv, uuv);
  for (int c = 0; c < 3; ++c) o[c] = v[c] + 2.0f * (w * uv[c] + uuv[c]);
}

__global__ __launch_bounds__(256) void ingest_pose3d_kernel(const float* __restrict__ raw, int Jraw, const int* __restrict__ frames,
                                                             long N, IngestMap map, int J, int has_cam, float qw, float qx, float qy,
                                                             float qz, float tx, float ty, float tz, int root_raw, int root_out,
                                                             float div, float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * J) return;
  const long n = i / J;
  const int j = (int)(i - n * J);
  const float* fr = raw + (long)(frames != nullptr ? frames[n] : n) * Jraw * 3;
  float x[3], r[3] = {0.f, 0.f, 0.f};
  for (int c = 0; c < 3; ++c) x[c] = fr[map.j[j] * 3 + c];
  if (root_raw >= 0)
    for (int c = 0; c < 3; ++c) x[c] = x[c] - fr[root_raw * 3 + c];
  if (has_cam) {
    const float q[3] = {qx, qy, qz}, t[3] = {tx, ty, tz};
    float v[3];
    for (int c = 0; c < 3; ++c) v[c] = x[c] - t[c];
    qrot3(qw, q, v, x);
    if (root_out >= 0) {
      for (int c = 0; c < 3; ++c) v[c] = fr[map.j[root_out] * 3 + c] - t[c];
      qrot3(qw, q, v, r);
    }
  } else if (root_out >= 0) {
    for (int c = 0; c < 3; ++c) r[c] = fr[map.j[root_out] * 3 + c];
  }
  for (int c = 0; c < 3; ++c) out[i * 3 + c] = (x[c] - r[c]) / div;
}

__global__ __launch_bounds__(256) void ingest_pose2d_kernel(const float* __restrict__ raw, int Jraw, int Craw,
                                                             const int* __restrict__ frames, long N, IngestMap map, int J, float w,
                                                             double h_over_w, float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * J) return;
  const long n = i / J;
  const int j = (int)(i - n * J);
  const float* p = raw + ((long)(frames != nullptr ? frames[n] : n) * Jraw + map.j[j]) * Craw;
  out[i * 2] = (float)((double)(p[0] / w * 2.0f) - 1.0);
  out[i * 2 + 1] = (float)((double)(p[1] / w * 2.0f) - h_over_w);
}

static int fill_map(const int* joint_map, int J, int Jraw, IngestMap& m) {
  MP_CHECK(J >= 1 && J <= 32 && Jraw >= 1, MP_ERR_ARG, "ingest: J=%d (1..32), Jraw=%d", J, Jraw);
  for (int j = 0; j < J; ++j) {
    m.j[j] = joint_map != nullptr ? joint_map[j] : j;
    MP_CHECK(m.j[j] >= 0 && m.j[j] < Jraw, MP_ERR_ARG, "ingest: joint_map[%d] = %d outside the %d raw joints", j, m.j[j], Jraw);
  }
  return MP_OK;
}

int ingest_pose3d(const float* raw, int Jraw, const int* frames, long N, const int* joint_map, int J, const float* quat,
                  const float* trans, int root_raw, int root_out, float div, float* out, hipStream_t st) {
  IngestMap m;
  int rc = fill_map(joint_map, J, Jraw, m);
  if (rc != MP_OK) return rc;
  MP_CHECK(N >= 0 && div != 0.f && (N == 0 || (raw != nullptr && out != nullptr)), MP_ERR_ARG,
           "ingest_pose3d: null buffer, N < 0 or div == 0");
  MP_CHECK((quat == nullptr) == (trans == nullptr), MP_ERR_ARG, "ingest_pose3d: orientation and translation come together");
  MP_CHECK(root_raw >= -1 && root_raw < Jraw && root_out >= -1 && root_out < J, MP_ERR_ARG, "ingest_pose3d: root index out of range");
  if (N == 0) return MP_OK;
  // qinverse (quaternion.py:24-31): conjugate of a unit quaternion
  const float qw = quat ? quat[0] : 1.f, qx = quat ? -quat[1] : 0.f, qy = quat ? -quat[2] : 0.f, qz = quat ? -quat[3] : 0.f;
  hipLaunchKernelGGL(ingest_pose3d_kernel, dim3((unsigned)cdiv(N * J, 256L)), dim3(256), 0, st, raw, Jraw, frames, N, m, J,
                     quat != nullptr, qw, qx, qy, qz, trans ? trans[0] : 0.f, trans ? trans[1] : 0.f, trans ? trans[2] : 0.f, root_raw,
                     root_out, div, out);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

int ingest_pose2d(const float* raw, int Jraw, int Craw, const int* frames, long N, const int* joint_map, int J, float w, float h,
                  float* out, hipStream_t st) {
  IngestMap m;
  int rc = fill_map(joint_map, J, Jraw, m);
  if (rc != MP_OK) return rc;
  MP_CHECK(N >= 0 && (N == 0 || (raw != nullptr && out != nullptr)) && Craw >= 2 && w > 0.f && h > 0.f, MP_ERR_ARG,
           "ingest_pose2d: null buffer, N < 0, fewer than 2 channels or an empty image");
  if (N == 0) return MP_OK;
  hipLaunchKernelGGL(ingest_pose2d_kernel, dim3((unsigned)cdiv(N * J, 256L)), dim3(256), 0, st, raw, Jraw, Craw, frames, N, m, J, w,
                     (double)h / (double)w, out);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

}  // namespace mp
