// Fused manifold decoder: 6-D -> SO(3) Gram-Schmidt (rotation_tools.py:35-57) or the 4-D "two unit 2-vectors" representation
// (rotation_tools.py:60-116, model.rot_dim=4: R = R_theta R_phi, two planar rotations), T-pose offsets from segment
// lengths (pose_decoder.py:98-120, closed form offset_j = op_j * len_{j-1}) and the forward-kinematics chain
// over the 17-joint H36M tree (forward_kinematics.py:6-48), forward and backward, in one kernel each.
//
// Mapping: one lane per joint, 3 poses per 64-lane wave (lanes 0..50).  The kinematic chain is walked level
// by level (tree depth <= 5); a joint fetches its parent's world rotation / position with wavefront shuffles,
// so nothing but the 6-D inputs and the joint positions ever touches HBM:
//   forward : 17*6*4 = 408 B read + 17*3*4 = 204 B written per decoded pose (+ 64 B of lengths per window)
//   backward: 408 + 204 B read, 408 + 64 B written per decoded pose.
// The reference materialises (B*K*T, 17, 3, 3) rotation matrices, a 1215x repeated length tensor and ~100
// python-loop launches for the same work.
#include "common.h"
#include "kernels.h"

namespace mp {

constexpr int NJ = 17;
__constant__ int c_parent[NJ] = {-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 9, 8, 11, 12, 8, 14, 15};   // skeleton.py / dataset_3dhp.py:132-138
__constant__ int c_depth[NJ] = {0, 1, 2, 3, 1, 2, 3, 1, 2, 3, 4, 3, 4, 5, 3, 4, 5};
__constant__ int c_child[NJ][3] = {{1, 4, 7},   {2, -1, -1},  {3, -1, -1},  {-1, -1, -1}, {5, -1, -1},  {6, -1, -1},
                                   {-1, -1, -1}, {8, -1, -1},  {9, 11, 14},  {10, -1, -1}, {-1, -1, -1}, {12, -1, -1},
                                   {13, -1, -1}, {-1, -1, -1}, {15, -1, -1}, {16, -1, -1}, {-1, -1, -1}};
// T_POSE_OPERATORS, h36m_lifting.py:40-57 (index = joint; joint 0 unused)
__constant__ float c_op[NJ][3] = {{0, 0, 0},  {1, 0, 0},  {0, -1, 0}, {0, -1, 0}, {-1, 0, 0}, {0, -1, 0}, {0, -1, 0}, {0, 1, 0}, {0, 1, 0},
                                  {0, 1, 0},  {0, 1, 0},  {-1, 0, 0}, {-1, 0, 0}, {-1, 0, 0}, {1, 0, 0},  {1, 0, 0},  {1, 0, 0}};
constexpr int MAX_DEPTH = 5;
constexpr float GS_EPS = 1e-8f;   // rotation_tools.py:10-13

struct M3 { float m[9]; };   // row-major 3x3

__device__ __forceinline__ M3 matmul3(const M3& a, const M3& b) {
  M3 c;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) c.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
  return c;
}
__device__ __forceinline__ M3 matmul3_nt(const M3& a, const M3& b) {   // a * b^T
  M3 c;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) c.m[3 * i + j] = a.m[3 * i] * b.m[3 * j] + a.m[3 * i + 1] * b.m[3 * j + 1] + a.m[3 * i + 2] * b.m[3 * j + 2];
  return c;
}
__device__ __forceinline__ M3 matmul3_tn(const M3& a, const M3& b) {   // a^T * b
  M3 c;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) c.m[3 * i + j] = a.m[i] * b.m[j] + a.m[3 + i] * b.m[3 + j] + a.m[6 + i] * b.m[6 + j];
  return c;
}
__device__ __forceinline__ M3 shfl_m3(const M3& a, int src) {
  M3 c;
#pragma unroll
  for (int i = 0; i < 9; ++i) c.m[i] = __shfl(a.m[i], src, 64);
  return c;
}
__device__ __forceinline__ void cross3(const float* u, const float* v, float* o) {
  o[0] = u[1] * v[2] - u[2] * v[1];
  o[1] = u[2] * v[0] - u[0] * v[2];
  o[2] = u[0] * v[1] - u[1] * v[0];
}

// Gram-Schmidt: returns R with columns x | y | z; also the intermediates needed by the backward
struct GS { float x[3], y[3], z[3], na, nz; };
// The Gram-Schmidt step is evaluated with the reference's operation order and one rounding per operation
// (no FMA contraction, true divisions): for (near-)degenerate inputs (colinear 6-D halves) the result is
// ill-conditioned and only the same rounding sequence reproduces the reference's joints.
__device__ __forceinline__ void cross3_exact(const float* u, const float* v, float* o) {
  o[0] = __fsub_rn(__fmul_rn(u[1], v[2]), __fmul_rn(u[2], v[1]));
  o[1] = __fsub_rn(__fmul_rn(u[2], v[0]), __fmul_rn(u[0], v[2]));
  o[2] = __fsub_rn(__fmul_rn(u[0], v[1]), __fmul_rn(u[1], v[0]));
}
__device__ __forceinline__ float norm3_exact(const float* v) {
  return __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(v[0], v[0]), __fmul_rn(v[1], v[1])), __fmul_rn(v[2], v[2])));
}
__device__ __forceinline__ GS gram_schmidt(const float* r6) {
  GS g;
  const float na = norm3_exact(r6);
  g.na = na;
  const float da = fmaxf(na, GS_EPS);
  g.x[0] = __fdiv_rn(r6[0], da); g.x[1] = __fdiv_rn(r6[1], da); g.x[2] = __fdiv_rn(r6[2], da);
  float zc[3];
  cross3_exact(g.x, r6 + 3, zc);
  const float nz = norm3_exact(zc);
  g.nz = nz;
  const float dz = fmaxf(nz, GS_EPS);
  g.z[0] = __fdiv_rn(zc[0], dz); g.z[1] = __fdiv_rn(zc[1], dz); g.z[2] = __fdiv_rn(zc[2], dz);
  cross3_exact(g.z, g.x, g.y);
  return g;
}
__device__ __forceinline__ M3 gs_matrix(const GS& g) {
  M3 R;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    R.m[3 * r] = g.x[r];
    R.m[3 * r + 1] = g.y[r];
    R.m[3 * r + 2] = g.z[r];
  }
  return R;
}

// ---- 4-D representation (rotation_tools.py:60-116): (c1, s1) = normalize(r[0:2]), (c2, s2) = normalize(r[2:4]);
//   R_theta = [theta_x | theta_y | e_z], theta_y = (c1, s1, 0), theta_x = theta_y x e_z = (s1, -c1, 0)
//   R_phi   = [e_x | phi_y | phi_z],     phi_y = (0, c2, s2),   phi_z = e_x x phi_y = (0, -s2, c2)
//   R = R_theta R_phi = [[s1, c1 c2, -c1 s2], [-c1, s1 c2, -s1 s2], [0, s2, c2]]
struct R4 { float c1, s1, c2, s2, na, nb; };
__device__ __forceinline__ R4 rot4_angles(const float* r4) {
  R4 g;
  g.na = __fsqrt_rn(__fadd_rn(__fmul_rn(r4[0], r4[0]), __fmul_rn(r4[1], r4[1])));
  g.nb = __fsqrt_rn(__fadd_rn(__fmul_rn(r4[2], r4[2]), __fmul_rn(r4[3], r4[3])));
  const float da = fmaxf(g.na, GS_EPS), db = fmaxf(g.nb, GS_EPS);
  g.c1 = __fdiv_rn(r4[0], da); g.s1 = __fdiv_rn(r4[1], da);
  g.c2 = __fdiv_rn(r4[2], db); g.s2 = __fdiv_rn(r4[3], db);
  return g;
}
__device__ __forceinline__ M3 rot4_matrix(const R4& g) {
  M3 R;
  R.m[0] = g.s1;  R.m[1] = g.c1 * g.c2; R.m[2] = -(g.c1 * g.s2);
  R.m[3] = -g.c1; R.m[4] = g.s1 * g.c2; R.m[5] = -(g.s1 * g.s2);
  R.m[6] = 0.f;   R.m[7] = g.s2;        R.m[8] = g.c2;
  return R;
}
// d(unit 2-vector) -> d(raw 2-vector), the normalize_vector backward (eps clamp like the 6-D path)
__device__ __forceinline__ void unit2_bwd(float c, float s, float n, float dc, float ds, float* o) {
  if (n > GS_EPS) {
    const float dot = c * dc + s * ds, inv = 1.0f / n;
    o[0] = (dc - c * dot) * inv; o[1] = (ds - s * dot) * inv;
  } else {
    o[0] = dc * (1.0f / GS_EPS); o[1] = ds * (1.0f / GS_EPS);
  }
}

struct FkGeom { int n, j, slot, b, k, t; bool valid; long row; };
__device__ __forceinline__ FkGeom fk_geom(int B, int K, int T) {
  FkGeom g;
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  g.slot = lane / NJ;
  g.j = lane - g.slot * NJ;
  g.n = wave * 3 + g.slot;                        // pose index in (b, k, t) order
  g.valid = (g.slot < 3) && (g.n < B * K * T);
  const int nn = g.valid ? g.n : 0;
  g.t = nn % T;
  g.k = (nn / T) % K;
  g.b = nn / (T * K);
  if (g.slot >= 3) g.j = 0;
  g.row = ((long)g.k * B * T + (long)g.b * T + g.t) * NJ + g.j;   // row of the (K, M, stride) head output
  return g;
}

template <int RD>
__global__ __launch_bounds__(256) void fk_fwd_kernel(const float* __restrict__ rot, int rs, const float* __restrict__ lengths,
                                                      float* __restrict__ poses, int B, int K, int T) {
  const FkGeom g = fk_geom(B, K, T);
  float r6[6] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f};
  float off[3] = {0.f, 0.f, 0.f};
  if (g.valid) {
#pragma unroll
    for (int c = 0; c < RD; ++c) r6[c] = rot[g.row * rs + c];
    if (g.j > 0) {
      const float len = lengths[g.b * (NJ - 1) + g.j - 1];
      off[0] = c_op[g.j][0] * len; off[1] = c_op[g.j][1] * len; off[2] = c_op[g.j][2] * len;
    }
  }
  M3 R;
  if (RD == 6) R = gs_matrix(gram_schmidt(r6));
  else R = rot4_matrix(rot4_angles(r6));
  M3 Rw = R;
  float p[3] = {0.f, 0.f, 0.f};
  const int plane = g.slot * NJ + max(c_parent[g.j], 0);
  const int depth = c_depth[g.j];
#pragma unroll
  for (int d = 1; d <= MAX_DEPTH; ++d) {
    const M3 Rp = shfl_m3(Rw, plane);
    const float p0 = __shfl(p[0], plane, 64), p1 = __shfl(p[1], plane, 64), p2 = __shfl(p[2], plane, 64);
    if (depth == d) {
      Rw = matmul3(Rp, R);
      p[0] = Rw.m[0] * off[0] + Rw.m[1] * off[1] + Rw.m[2] * off[2] + p0;
      p[1] = Rw.m[3] * off[0] + Rw.m[4] * off[1] + Rw.m[5] * off[2] + p1;
      p[2] = Rw.m[6] * off[0] + Rw.m[7] * off[1] + Rw.m[8] * off[2] + p2;
    }
  }
  if (g.valid) {
    float* o = poses + ((long)g.n * NJ + g.j) * 3;
    o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
  }
}

template <int RD>
__global__ __launch_bounds__(256) void fk_bwd_kernel(const float* __restrict__ rot, int rs, const float* __restrict__ lengths,
                                                      const float* __restrict__ dposes, float* __restrict__ drot,
                                                      float* __restrict__ dlen_pose, int B, int K, int T) {
  const FkGeom g = fk_geom(B, K, T);
  float r6[6] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f};
  float off[3] = {0.f, 0.f, 0.f};
  float G[3] = {0.f, 0.f, 0.f};
  if (g.valid) {
#pragma unroll
    for (int c = 0; c < RD; ++c) r6[c] = rot[g.row * rs + c];
    if (g.j > 0) {
      const float len = lengths[g.b * (NJ - 1) + g.j - 1];
      off[0] = c_op[g.j][0] * len; off[1] = c_op[g.j][1] * len; off[2] = c_op[g.j][2] * len;
      const float* gp = dposes + ((long)g.n * NJ + g.j) * 3;   // the root position is the constant 0: no gradient
      G[0] = gp[0]; G[1] = gp[1]; G[2] = gp[2];
    }
  }
  GS gs;
  R4 g4;
  M3 R;
  if (RD == 6) { gs = gram_schmidt(r6); R = gs_matrix(gs); }
  else { g4 = rot4_angles(r6); R = rot4_matrix(g4); }
  const int base = g.slot * NJ;
  const int plane = base + max(c_parent[g.j], 0);
  const int depth = c_depth[g.j];
  // forward recompute of the world rotations
  M3 Rw = R;
#pragma unroll
  for (int d = 1; d <= MAX_DEPTH; ++d) {
    const M3 Rp = shfl_m3(Rw, plane);
    if (depth == d) Rw = matmul3(Rp, R);
  }
  const M3 Rparent = shfl_m3(Rw, plane);
  // subtree sums of the position gradients (p_j = Rw_j off_j + p_parent)
  int cl[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) cl[c] = c_child[g.j][c];
#pragma unroll
  for (int d = MAX_DEPTH; d >= 1; --d) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int src = base + max(cl[c], 0);
      const float g0 = __shfl(G[0], src, 64), g1 = __shfl(G[1], src, 64), g2 = __shfl(G[2], src, 64);
      if (depth == d - 1 && cl[c] >= 0) { G[0] += g0; G[1] += g1; G[2] += g2; }
    }
  }
  // world-rotation gradients: own outer product + children's dRw_c * R_c^T
  M3 dRw;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int c = 0; c < 3; ++c) dRw.m[3 * i + c] = G[i] * off[c];
  M3 msg;
#pragma unroll
  for (int i = 0; i < 9; ++i) msg.m[i] = 0.f;
#pragma unroll
  for (int d = MAX_DEPTH; d >= 1; --d) {
    if (depth == d) msg = matmul3_nt(dRw, R);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const M3 mc = shfl_m3(msg, base + max(cl[c], 0));
      if (depth == d - 1 && cl[c] >= 0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) dRw.m[i] += mc.m[i];
      }
    }
  }
  // local rotation gradient and segment-length gradient
  M3 dR = dRw;
  if (g.j > 0) dR = matmul3_tn(Rparent, dRw);
  float doff[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) doff[c] = Rw.m[c] * G[0] + Rw.m[3 + c] * G[1] + Rw.m[6 + c] * G[2];
  const float dlen = c_op[g.j][0] * doff[0] + c_op[g.j][1] * doff[1] + c_op[g.j][2] * doff[2];
  if (RD == 4) {      // R = [[s1, c1 c2, -c1 s2], [-c1, s1 c2, -s1 s2], [0, s2, c2]]
    const float ds1 = dR.m[0] + dR.m[4] * g4.c2 - dR.m[5] * g4.s2;
    const float dc1 = dR.m[1] * g4.c2 - dR.m[2] * g4.s2 - dR.m[3];
    const float dc2 = dR.m[1] * g4.c1 + dR.m[4] * g4.s1 + dR.m[8];
    const float ds2 = dR.m[7] - dR.m[2] * g4.c1 - dR.m[5] * g4.s1;
    float da2[2], db2[2];
    unit2_bwd(g4.c1, g4.s1, g4.na, dc1, ds1, da2);
    unit2_bwd(g4.c2, g4.s2, g4.nb, dc2, ds2, db2);
    if (g.valid) {
      float* o = drot + g.row * rs;
      o[0] = da2[0]; o[1] = da2[1]; o[2] = db2[0]; o[3] = db2[1];
      if (g.j > 0) dlen_pose[(long)g.n * (NJ - 1) + g.j - 1] = dlen;
    }
    return;
  }
  // Gram-Schmidt backward
  float dx[3] = {dR.m[0], dR.m[3], dR.m[6]}, dy[3] = {dR.m[1], dR.m[4], dR.m[7]}, dz[3] = {dR.m[2], dR.m[5], dR.m[8]};
  float t3[3];
  cross3(gs.x, dy, t3);                       // y = z x x  ->  dz += x x dy ; dx += dy x z
  dz[0] += t3[0]; dz[1] += t3[1]; dz[2] += t3[2];
  cross3(dy, gs.z, t3);
  dx[0] += t3[0]; dx[1] += t3[1]; dx[2] += t3[2];
  float dzc[3];
  if (gs.nz > GS_EPS) {
    const float dot = gs.z[0] * dz[0] + gs.z[1] * dz[1] + gs.z[2] * dz[2];
    const float inv = 1.0f / gs.nz;
#pragma unroll
    for (int c = 0; c < 3; ++c) dzc[c] = (dz[c] - gs.z[c] * dot) * inv;
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) dzc[c] = dz[c] * (1.0f / GS_EPS);
  }
  float db[3];
  cross3(r6 + 3, dzc, t3);                    // zc = x x b  ->  dx += b x dzc ; db = dzc x x
  dx[0] += t3[0]; dx[1] += t3[1]; dx[2] += t3[2];
  cross3(dzc, gs.x, db);
  float da[3];
  if (gs.na > GS_EPS) {
    const float dot = gs.x[0] * dx[0] + gs.x[1] * dx[1] + gs.x[2] * dx[2];
    const float inv = 1.0f / gs.na;
#pragma unroll
    for (int c = 0; c < 3; ++c) da[c] = (dx[c] - gs.x[c] * dot) * inv;
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) da[c] = dx[c] * (1.0f / GS_EPS);
  }
  if (g.valid) {
    float* o = drot + g.row * rs;
    o[0] = da[0]; o[1] = da[1]; o[2] = da[2]; o[3] = db[0]; o[4] = db[1]; o[5] = db[2];
    if (g.j > 0) dlen_pose[(long)g.n * (NJ - 1) + g.j - 1] = dlen;
  }
}

static int fk_grid(int N) { return cdiv(cdiv(N, 3), 4); }

int fk_decode_fwd(const float* rot, int rot_stride, int rot_dim, const float* lengths, float* poses, int B, int K, int T,
                  hipStream_t st) {
  MP_CHECK((rot_dim == 6 || rot_dim == 4), MP_ERR_ARG, "fk_decode: rotation representation of dimension %d (4 or 6)", rot_dim);
  MP_CHECK(B > 0 && K > 0 && T > 0 && rot_stride >= rot_dim, MP_ERR_ARG, "fk_decode_fwd: bad dims B=%d K=%d T=%d stride=%d", B, K, T,
           rot_stride);
  if (rot_dim == 6) hipLaunchKernelGGL(fk_fwd_kernel<6>, dim3(fk_grid(B * K * T)), dim3(256), 0, st, rot, rot_stride, lengths, poses, B, K, T);
  else hipLaunchKernelGGL(fk_fwd_kernel<4>, dim3(fk_grid(B * K * T)), dim3(256), 0, st, rot, rot_stride, lengths, poses, B, K, T);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

int fk_decode_bwd(const float* rot, int rot_stride, int rot_dim, const float* lengths, const float* dposes, float* drot,
                  float* dlen_pose, int B, int K, int T, hipStream_t st) {
  MP_CHECK((rot_dim == 6 || rot_dim == 4), MP_ERR_ARG, "fk_decode: rotation representation of dimension %d (4 or 6)", rot_dim);
  MP_CHECK(B > 0 && K > 0 && T > 0 && rot_stride >= rot_dim, MP_ERR_ARG, "fk_decode_bwd: bad dims");
  if (rot_dim == 6)
    hipLaunchKernelGGL(fk_bwd_kernel<6>, dim3(fk_grid(B * K * T)), dim3(256), 0, st, rot, rot_stride, lengths, dposes, drot, dlen_pose,
                       B, K, T);
  else
    hipLaunchKernelGGL(fk_bwd_kernel<4>, dim3(fk_grid(B * K * T)), dim3(256), 0, st, rot, rot_stride, lengths, dposes, drot, dlen_pose,
                       B, K, T);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

}  // namespace mp
