// Sustained matrix-core rate of the chip with nothing else in the way: every wave issues independent v_mfma_f32_16x16x32_bf16 from
// registers (no LDS, no memory), 8 waves per CU on all CUs, for a few hundred ms.  Prints TFLOP/s for 1 and 2 waves per SIMD and the
// per-MFMA issue interval implied at the nominal 2.4 GHz.   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(512) void mfma_loop(int iters, float* out) {
  // four operand sets of pseudo-random values, rotated every MFMA: data that toggles like a real GEMM's (constant operands draw less power)
  bf16x8_t av[4], bv[4];
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int s = 0; s < 4; ++s)
    for (int i = 0; i < 8; ++i) {
      h = h * 1664525u + 1013904223u; av[s][i] = (__bf16)(((int)(h >> 9) % 2001 - 1000) * 1e-3f);
      h = h * 1664525u + 1013904223u; bv[s][i] = (__bf16)(((int)(h >> 9) % 2001 - 1000) * 1e-3f);
    }
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[i & 3], bv[(i >> 2) & 3], acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[0] = s;
}

int main() {
  int cus = 256;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  float* out;
  hipMalloc(&out, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int threads : {256, 512}) {
    const int iters = 1000000;
    constexpr int NACC = 16;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(mfma_loop<NACC>, dim3(cus), dim3(threads), 0, 0, iters, out);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0.f;
      hipEventElapsedTime(&ms, e0, e1);
      const double flops = (double)cus * (threads / 64) * (double)iters * NACC * 16 * 16 * 32 * 2;
      const double per_simd_mfma = (double)iters * NACC * (threads / 256);      // MFMAs issued per SIMD
      printf("%d waves/SIMD: %.1f ms, %.0f TFLOP/s, %.2f cycles per MFMA per SIMD at 2.4 GHz (16 = peak)\n", threads / 256, ms, flops / ms / 1e9,
             ms * 1e-3 * 2.4e9 / per_simd_mfma);
    }
  }
  return 0;
}
