// What a row-streaming kernel can reach on this box: a float4 copy (read 4 B + write 4 B per element) and a LayerNorm-shaped pass
// (read an fp32 row of 512, wave-reduce it, write two bf16 planes: 4 B in, 4 B out per element), over grids / rows in flight / cache policies.
// Buffers of 668 MB (the bench's token count x 512 channels), far beyond the 256 MB Infinity Cache.
// hipcc --offload-arch=gfx950 -O3 tools/probes/hbm_stream.hip -o /tmp/hbm_stream && /tmp/hbm_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int R, int NT>
__global__ __launch_bounds__(256) void copy_rows(const float4* __restrict__ src, float4* __restrict__ dst, int M) {
  // one wave per row of 512 floats (2 float4 per lane), R rows in flight per wave
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int m0 = wave; m0 < M; m0 += R * nwaves) {
    float4 v[R][2];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int m = m0 + r * nwaves;
      if (m < M) {
        const float4* p = src + (long)m * 128 + lane;
        if (NT) {
          const f4 t0 = __builtin_nontemporal_load((const f4*)p), t1 = __builtin_nontemporal_load((const f4*)(p + 64));
          v[r][0] = make_float4(t0.x, t0.y, t0.z, t0.w); v[r][1] = make_float4(t1.x, t1.y, t1.z, t1.w);
        }
        else { v[r][0] = p[0]; v[r][1] = p[64]; }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int m = m0 + r * nwaves;
      if (m < M) {
        float4* q = dst + (long)m * 128 + lane;
        if (NT) {
          __builtin_nontemporal_store(f4{v[r][0].x, v[r][0].y, v[r][0].z, v[r][0].w}, (f4*)q);
          __builtin_nontemporal_store(f4{v[r][1].x, v[r][1].y, v[r][1].z, v[r][1].w}, (f4*)(q + 64));
        }
        else { q[0] = v[r][0]; q[1] = v[r][1]; }
      }
    }
  }
}

// contiguous chunk per wave instead of rows strided by the wave count
template <int R>
__global__ __launch_bounds__(256) void copy_chunks(const float4* __restrict__ src, float4* __restrict__ dst, int M) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const int per = (M + nwaves - 1) / nwaves;
  const int lo = wave * per, hi = min(M, lo + per);
  for (int m0 = lo; m0 < hi; m0 += R) {
    float4 v[R][2];
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (m0 + r < hi) { const float4* p = src + (long)(m0 + r) * 128 + lane; v[r][0] = p[0]; v[r][1] = p[64]; }
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (m0 + r < hi) { float4* q = dst + (long)(m0 + r) * 128 + lane; q[0] = v[r][0]; q[1] = v[r][1]; }
  }
}

__device__ __forceinline__ float wsum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int R>
__global__ __launch_bounds__(256) void ln_rows(const float4* __restrict__ src, uint2* __restrict__ hi, uint2* __restrict__ lo, int M) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int m0 = wave; m0 < M; m0 += R * nwaves) {
    float4 v[R][2];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int m = m0 + r * nwaves;
      if (m < M) { const float4* p = src + (long)m * 128 + lane; v[r][0] = p[0]; v[r][1] = p[64]; }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int m = m0 + r * nwaves;
      if (m < M) {
        float s = 0.f;
        for (int i = 0; i < 2; ++i) s += v[r][i].x + v[r][i].y + v[r][i].z + v[r][i].w;
        const float mean = wsum(s) * (1.f / 512);
        float q = 0.f;
        for (int i = 0; i < 2; ++i) {
          const float a = v[r][i].x - mean, b = v[r][i].y - mean, c = v[r][i].z - mean, d = v[r][i].w - mean;
          q += a * a + b * b + c * c + d * d;
        }
        const float rstd = rsqrtf(wsum(q) * (1.f / 512) + 1e-6f);
        for (int i = 0; i < 2; ++i) {
          const float o[4] = {(v[r][i].x - mean) * rstd, (v[r][i].y - mean) * rstd, (v[r][i].z - mean) * rstd, (v[r][i].w - mean) * rstd};
          unsigned h[4], l[4];
          for (int k = 0; k < 4; ++k) {
            const unsigned u = __float_as_uint(o[k]);
            const unsigned hb = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
            h[k] = hb >> 16;
            const unsigned w = __float_as_uint(o[k] - __uint_as_float(hb));
            l[k] = (w + 0x7FFFu + ((w >> 16) & 1u)) >> 16;
          }
          hi[(long)m * 128 + lane + 64 * i] = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
          lo[(long)m * 128 + lane + 64 * i] = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
        }
      }
    }
  }
}


// LN-shaped, every wave owns K consecutive rows (R in flight), grid = M / (4 K) workgroups: the structure a kernel needs when it keeps per-wave
// state across rows (gamma / beta in registers, dgamma / dbeta sums)
template <int K, int R>
__global__ __launch_bounds__(256) void ln_chunk(const float4* __restrict__ src, uint2* __restrict__ hi, uint2* __restrict__ lo, int M) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lo_r = wave * K, hi_r = min(M, lo_r + K);
  for (int m0 = lo_r; m0 < hi_r; m0 += R) {
    float4 v[R][2];
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (m0 + r < hi_r) { const float4* p = src + (long)(m0 + r) * 128 + lane; v[r][0] = p[0]; v[r][1] = p[64]; }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int m = m0 + r;
      if (m < hi_r) {
        float s = 0.f;
        for (int i = 0; i < 2; ++i) s += v[r][i].x + v[r][i].y + v[r][i].z + v[r][i].w;
        const float mean = wsum(s) * (1.f / 512);
        float q = 0.f;
        for (int i = 0; i < 2; ++i) {
          const float a = v[r][i].x - mean, b = v[r][i].y - mean, c = v[r][i].z - mean, d = v[r][i].w - mean;
          q += a * a + b * b + c * c + d * d;
        }
        const float rstd = rsqrtf(wsum(q) * (1.f / 512) + 1e-6f);
        for (int i = 0; i < 2; ++i) {
          const float o[4] = {(v[r][i].x - mean) * rstd, (v[r][i].y - mean) * rstd, (v[r][i].z - mean) * rstd, (v[r][i].w - mean) * rstd};
          unsigned h[4], l[4];
          for (int k = 0; k < 4; ++k) {
            const unsigned u = __float_as_uint(o[k]);
            const unsigned hb = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
            h[k] = hb >> 16;
            const unsigned w = __float_as_uint(o[k] - __uint_as_float(hb));
            l[k] = (w + 0x7FFFu + ((w >> 16) & 1u)) >> 16;
          }
          hi[(long)m * 128 + lane + 64 * i] = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
          lo[(long)m * 128 + lane + 64 * i] = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
        }
      }
    }
  }
}

// the same with the rows of a workgroup interleaved over its 4 waves (row = base + 4 i + wave): a workgroup sweeps 4 K consecutive rows front to back
template <int K, int R>
__global__ __launch_bounds__(256) void copy_wgsweep(const float4* __restrict__ src, float4* __restrict__ dst, int M) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int base = blockIdx.x * 4 * K;
  for (int i0 = 0; i0 < K; i0 += R) {
    float4 v[R][2];
#pragma unroll
    for (int r = 0; r < R; ++r) { const int m = base + 4 * (i0 + r) + wv; if (m < M) { const float4* p = src + (long)m * 128 + lane; v[r][0] = p[0]; v[r][1] = p[64]; } }
#pragma unroll
    for (int r = 0; r < R; ++r) { const int m = base + 4 * (i0 + r) + wv; if (m < M) { float4* q = dst + (long)m * 128 + lane; q[0] = v[r][0]; q[1] = v[r][1]; } }
  }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <typename F>
static float time_us(F launch, int n = 10) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) launch();
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < n; ++i) launch();
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / n;
}

int main() {
  const int M = 326349;
  const size_t bytes = (size_t)M * 512 * 4;
  float4 *a, *b;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
  const double gb = 2.0 * bytes / 1e9;
  for (int grid : {1024, 2048, 4096, 8192, 16384, 81588}) {
    printf("grid %5d x 256:", grid);
    printf("  rows R=1 %6.0f", gb / time_us([&] { hipLaunchKernelGGL((copy_rows<1, 0>), dim3(grid), dim3(256), 0, 0, a, b, M); }) * 1e6);
    printf("  R=2 %6.0f", gb / time_us([&] { hipLaunchKernelGGL((copy_rows<2, 0>), dim3(grid), dim3(256), 0, 0, a, b, M); }) * 1e6);
    printf("  R=4 %6.0f", gb / time_us([&] { hipLaunchKernelGGL((copy_rows<4, 0>), dim3(grid), dim3(256), 0, 0, a, b, M); }) * 1e6);
    printf("  R=8 %6.0f", gb / time_us([&] { hipLaunchKernelGGL((copy_rows<8, 0>), dim3(grid), dim3(256), 0, 0, a, b, M); }) * 1e6);
    printf("  R=4 nt %6.0f", gb / time_us([&] { hipLaunchKernelGGL((copy_rows<4, 1>), dim3(grid), dim3(256), 0, 0, a, b, M); }) * 1e6);
    printf("  chunks R=4 %6.0f", gb / time_us([&] { hipLaunchKernelGGL((copy_chunks<4>), dim3(grid), dim3(256), 0, 0, a, b, M); }) * 1e6);
    printf("  | LN-shaped R=2 %6.0f", gb / time_us([&] { hipLaunchKernelGGL((ln_rows<2>), dim3(grid), dim3(256), 0, 0, a, (uint2*)b, (uint2*)b + (size_t)M * 128, M); }) * 1e6);
    printf("  R=4 %6.0f GB/s\n", gb / time_us([&] { hipLaunchKernelGGL((ln_rows<4>), dim3(grid), dim3(256), 0, 0, a, (uint2*)b, (uint2*)b + (size_t)M * 128, M); }) * 1e6);
  }
  auto g_of = [&](int k) { return (M + 4 * k - 1) / (4 * k); };
#define LNC(K, R) printf("  K=%d R=%d %6.0f", K, R, gb / time_us([&] { hipLaunchKernelGGL((ln_chunk<K, R>), dim3(g_of(K)), dim3(256), 0, 0, a, (uint2*)b, (uint2*)b + (size_t)M * 128, M); }) * 1e6)
  printf("LN-shaped, K consecutive rows per wave:");
  LNC(1, 1); LNC(2, 1); LNC(4, 1); LNC(4, 2); LNC(8, 1); LNC(8, 2); LNC(16, 1); LNC(16, 2); LNC(32, 1); LNC(32, 2); LNC(64, 2);
  printf(" GB/s\n");
#define CWS(K, R) printf("  K=%d R=%d %6.0f", K, R, gb / time_us([&] { hipLaunchKernelGGL((copy_wgsweep<K, R>), dim3(g_of(K)), dim3(256), 0, 0, a, b, M); }) * 1e6)
  printf("copy, workgroup sweeps 4K consecutive rows:");
  CWS(1, 1); CWS(4, 1); CWS(4, 2); CWS(8, 1); CWS(8, 2); CWS(16, 1); CWS(16, 2); CWS(32, 1); CWS(32, 2); CWS(64, 2);
  printf(" GB/s\n");
  printf("hipMemcpyDtoD: %6.0f GB/s\n", gb / time_us([&] { (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }) * 1e6);
  return 0;
}
