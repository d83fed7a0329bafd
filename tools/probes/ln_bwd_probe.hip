// Isolated timing of the two LayerNorm backward kernels of the training step on the bench's token count (M = 79 x 243 x 17 rows of C = 512):
// ln_bwd (norm2 backward + skip, bf16 dy, fp32 + 2-byte gradient out) and ln_bwd2 (norm1 backward + skip fused with the shared post-norm backward
// of the previous block).  Both move 16 bytes per element.  The kernels are compiled INTO this probe (variants: -DLNB2_R=..., -DLNB2_...):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Xclang -target-feature -Xclang -packed-fp32-ops -Iinclude tools/probes/ln_bwd_probe.hip -o /tmp/ln_bwd_probe
#include "../../manipose_amd/csrc/elementwise.hip"
#include <cstdio>
#include <cstdlib>
using namespace mp;
__global__ void fill_kernel(float* p, long n, float scale, unsigned seed) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = scale * ((float)(h & 0xffff) / 32768.0f - 1.0f);
  }
}
__global__ void fill_bf16_kernel(bf16* p, long n, float scale, unsigned seed) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = __float2bfloat16(scale * ((float)(h & 0xffff) / 32768.0f - 1.0f));
  }
}
__global__ void stats_kernel(float* st, long M) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (long)gridDim.x * blockDim.x) { st[2 * i] = 0.01f; st[2 * i + 1] = 1.7f; }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 79 * 243 * 17, C = 512, reps = 10;
  const long n = (long)M * C;
  float *x, *x0, *g, *st1, *st0, *gam, *bet, *dgam, *dbet, *dgam0, *dbet0, *scratch;
  bf16 *dy, *gb;
  CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&x0, n * 4)); CK(hipMalloc(&g, n * 4)); CK(hipMalloc(&dy, n * 2)); CK(hipMalloc(&gb, n * 2));
  CK(hipMalloc(&st1, (long)M * 8)); CK(hipMalloc(&st0, (long)M * 8));
  CK(hipMalloc(&gam, C * 4)); CK(hipMalloc(&bet, C * 4)); CK(hipMalloc(&dgam, C * 4)); CK(hipMalloc(&dbet, C * 4)); CK(hipMalloc(&dgam0, C * 4)); CK(hipMalloc(&dbet0, C * 4));
  const long sf = 32768L * 4 * C;
  CK(hipMalloc(&scratch, sf * 4));
  fill_kernel<<<1024, 256>>>(x, n, 1.0f, 1u); fill_kernel<<<1024, 256>>>(x0, n, 1.0f, 2u); fill_kernel<<<1024, 256>>>(g, n, 1e-3f, 3u);
  fill_bf16_kernel<<<1024, 256>>>(dy, n, 1e-3f, 4u);
  stats_kernel<<<256, 256>>>(st1, M); stats_kernel<<<256, 256>>>(st0, M);
  fill_kernel<<<1, 256>>>(gam, C, 1.0f, 5u); fill_kernel<<<1, 256>>>(bet, C, 0.1f, 6u);
  CK(hipMemset(dgam, 0, C * 4)); CK(hipMemset(dbet, 0, C * 4)); CK(hipMemset(dgam0, 0, C * 4)); CK(hipMemset(dbet0, 0, C * 4));
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int which = 0; which < 2; ++which) {
    float best = 1e30f, sum = 0.f;
    for (int r = 0; r < reps + 2; ++r) {
      CK(hipEventRecord(e0, 0));
      int rc;
      if (which == 0) rc = ln_bwd(dy, 1, x, st1, gam, g, g, gb, nullptr, 0, 243, 17, dgam, dbet, M, C, scratch, sf, 0, nullptr, nullptr, 1.0f, nullptr, nullptr);
      else rc = ln_bwd2(dy, 1, x, st1, gam, g, x0, st0, gam, bet, g, gb, nullptr, 0, 243, 17, dgam, dbet, dgam0, dbet0, M, C, scratch, sf, 0, nullptr, nullptr, 1.0f, nullptr, nullptr);
      if (rc) { printf("launch failed: %s\n", mp::last_error()); return 1; }
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 2) { sum += ms; if (ms < best) best = ms; }
    }
    printf("%s M=%d C=%d: mean %.1f us, best %.1f us (incl. the partial-sum reduction launch) = %.2f TB/s of 16 B per element\n", which == 0 ? "ln_bwd " : "ln_bwd2", M, C,
           sum / reps * 1e3, best * 1e3, 16.0 * n / (sum / reps * 1e-3) / 1e12);
  }
  return 0;
}
