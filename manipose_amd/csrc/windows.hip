// GPU-resident PoseSequenceGenerator: training / evaluation windows are cut out of the pose sequences
// where they lie in HBM, so eight GPUs at ~130 k poses/s each are not fed through eight CPU workers.
//   reference: hpe/mh_so3_hpe/data/generators.py:44-219 (window tables, random or strided start, replicate padding of a short
//   last window, miss_type "no_miss") + PoseFlip, hpe/mh_so3_hpe/augmentations/transforms.py:7-28 / functional.py:7-31
//   (u / x negated, left and right joints swapped).
// All sequences are stored back to back: poses_2d (Ntot, J, 2), poses_3d (Ntot, J, 3), seq_offset (S+1) first frame of each
// sequence.  One thread per (window, frame, joint) copies 2 + 3 floats: source frame = min(start + t, length - 1) of the
// window's sequence, source joint = mirror[j] for flipped windows; optional per-(frame, joint) multipliers / additive noise on the
// 2-D input (the generator's occlusion patterns, drawn on the host from numpy's RNG like the reference).  Pure data movement:
// 20 B read + 20 B written per joint.
#include "common.h"
#include "kernels.h"

namespace mp {

constexpr int WIN_MAXJ = 32;
struct WinArgs {
  const float* p2; const float* p3; const long* seq_offset;
  const int* win_seq; const int* win_start; const unsigned char* win_flip;
  const float* mask2d;                   // (B, T, J) multipliers of the 2-D input (occlusion patterns) or null
  const float* noise2d;                  // (B, T, J, 2) additive noise of the 2-D input (miss_type 'noisy') or null
  float* X; float* y;
  int B, T, J, S;
  unsigned char mirror[WIN_MAXJ];
};

__global__ __launch_bounds__(256) void gather_windows_kernel(WinArgs a) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;       // (w, t, j)
  const long n = (long)a.B * a.T * a.J;
  if (i >= n) return;
  const int j = (int)(i % a.J);
  const long wt = i / a.J;
  const int t = (int)(wt % a.T), w = (int)(wt / a.T);
  const int s = a.win_seq[w];
  const long f0 = a.seq_offset[s], len = a.seq_offset[s + 1] - f0;
  const long f = f0 + min((long)a.win_start[w] + t, len - 1);        // replicate the last frame past the end of the sequence
  const bool flip = a.win_flip != nullptr && a.win_flip[w] != 0;
  const int js = flip ? a.mirror[j] : j;
  const float sx = flip ? -1.0f : 1.0f;
  const float2 u = *reinterpret_cast<const float2*>(a.p2 + (f * a.J + js) * 2);
  const float* q = a.p3 + (f * a.J + js) * 3;
  float2 o2 = make_float2(sx * u.x, u.y);
  if (a.noise2d != nullptr) { const float2 nz = *reinterpret_cast<const float2*>(a.noise2d + i * 2); o2.x += nz.x; o2.y += nz.y; }
  if (a.mask2d != nullptr) { const float m = a.mask2d[i]; o2.x *= m; o2.y *= m; }
  *reinterpret_cast<float2*>(a.X + i * 2) = o2;
  float* o = a.y + i * 3;
  o[0] = sx * q[0]; o[1] = q[1]; o[2] = q[2];
}

int gather_windows(const float* p2, const float* p3, const long* seq_offset, int S, const int* win_seq, const int* win_start,
                   const unsigned char* win_flip, const int* mirror, const float* mask2d, const float* noise2d, int B, int T, int J,
                   float* X, float* y, hipStream_t st) {
  MP_CHECK(p2 && p3 && seq_offset && win_seq && win_start && X && y, MP_ERR_ARG, "gather_windows: null pointer");
  MP_CHECK(B > 0 && T > 0 && S > 0 && J > 0 && J <= WIN_MAXJ, MP_ERR_ARG, "gather_windows: B=%d T=%d S=%d J=%d out of range", B, T, S, J);
  MP_CHECK(win_flip == nullptr || mirror != nullptr, MP_ERR_ARG, "gather_windows: flip flags without a joint mirror table");
  WinArgs a = {};
  a.p2 = p2; a.p3 = p3; a.seq_offset = seq_offset; a.win_seq = win_seq; a.win_start = win_start; a.win_flip = win_flip;
  a.mask2d = mask2d; a.noise2d = noise2d;
  a.X = X; a.y = y; a.B = B; a.T = T; a.J = J; a.S = S;
  for (int j = 0; j < J; ++j) {
    const int m = mirror ? mirror[j] : j;
    MP_CHECK(m >= 0 && m < J, MP_ERR_ARG, "gather_windows: mirror[%d] = %d out of range", j, m);
    a.mirror[j] = (unsigned char)m;
  }
  const long n = (long)B * T * J;
  hipLaunchKernelGGL(gather_windows_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, a);
  MP_LAUNCH_CHECK();
  return MP_OK;
}

}  // namespace mp
