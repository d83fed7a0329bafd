// Matrix-core rate UNDER THE POWER CAP by instruction shape and operand type (round 5: the GEMMs run at the board's 1400 W cap, so what
// counts is flops per joule, not flops per cycle): every wave issues independent MFMAs from registers on toggling pseudo-random operands
// (no LDS, no memory), 2 waves per SIMD on all CUs, ~1.5 s per variant, rocm-smi (socket power, shader clock) sampled while it runs.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_power.hip -o /tmp/mfma_power && /tmp/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unistd.h>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(8 * sizeof(_Float16)))) _Float16 f16x8_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

// MODE 0: v_mfma_f32_16x16x32_bf16 (the GEMMs' instruction)   1: v_mfma_f32_32x32x16_bf16   2: v_mfma_f32_16x16x32_f16
//      3: v_mfma_scale_f32_16x16x128_f8f6f4 on fp8 (e4m3) operands (the f16f8 form's second step)   4: mode 0 on CONSTANT operands
template <int MODE>
__global__ __launch_bounds__(512) void mfma_loop(int iters, float* out) {
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  auto rnd = [&]() { h = h * 1664525u + 1013904223u; return ((int)(h >> 9) % 2001 - 1000) * 1e-3f; };
  float s = 0.f;
  if constexpr (MODE == 0 || MODE == 4) {
    bf16x8_t av[4], bv[4];
    for (int q = 0; q < 4; ++q) for (int i = 0; i < 8; ++i) { av[q][i] = (__bf16)(MODE == 4 ? 0.5f : rnd()); bv[q][i] = (__bf16)(MODE == 4 ? 0.25f : rnd()); }
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[i & 3], bv[(i >> 2) & 3], acc[i], 0, 0, 0);
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
  } else if constexpr (MODE == 1) {
    bf16x8_t av[4], bv[4];
    for (int q = 0; q < 4; ++q) for (int i = 0; i < 8; ++i) { av[q][i] = (__bf16)rnd(); bv[q][i] = (__bf16)rnd(); }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i & 3], bv[(i >> 1) & 3], acc[i & 3], 0, 0, 0);
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
  } else if constexpr (MODE == 2) {
    f16x8_t av[4], bv[4];
    for (int q = 0; q < 4; ++q) for (int i = 0; i < 8; ++i) { av[q][i] = (_Float16)rnd(); bv[q][i] = (_Float16)rnd(); }
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[i & 3], bv[(i >> 2) & 3], acc[i], 0, 0, 0);
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
  } else {
    i32x8 av[4], bv[4];
    for (int q = 0; q < 4; ++q) for (int i = 0; i < 8; ++i) { h = h * 1664525u + 1013904223u; av[q][i] = (int)(h & 0x77777777u); h = h * 1664525u + 1013904223u; bv[q][i] = (int)(h & 0x77777777u); }
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av[i & 3], bv[(i >> 2) & 3], acc[i], 0, 0, 0, 0x7f, 0, 0x7f);
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
  }
  if (s == 12345.678f) out[0] = s;
}

static void smi(char* buf, size_t n) {
  buf[0] = 0;
  FILE* f = popen("/opt/rocm/bin/rocm-smi --showclocks --showpower --csv 2>/dev/null | grep '^card0'", "r");
  if (f) { if (!fgets(buf, (int)n, f)) buf[0] = 0; pclose(f); }
}

template <int MODE>
static void run(const char* what, double flops_per_mfma, int mfma_per_iter, int iters, int cus, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(mfma_loop<MODE>, dim3(cus), dim3(512), 0, 0, iters / 50, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(mfma_loop<MODE>, dim3(cus), dim3(512), 0, 0, iters, out);
  hipEventRecord(e1);
  double pw = 0, ck = 0; int ns = 0;
  for (int k = 0; k < 4; ++k) {
    usleep(250000);
    if (hipEventQuery(e1) == hipSuccess) break;
    char b[512]; smi(b, sizeof(b));
    // card0,(fclk),lvl,(mclk),lvl,(sclk),lvl,(socclk),lvl,power
    int field = 0; char* p = b; double v[10] = {0}; 
    for (char* tok = strtok(p, ","); tok && field < 10; tok = strtok(nullptr, ","), ++field) { const char* q = tok; while (*q && (*q < '0' || *q > '9')) ++q; v[field] = atof(q); }
    if (k >= 1 && v[9] > 0) { pw += v[9]; ck += v[5]; ++ns; }
  }
  hipEventSynchronize(e1);
  float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
  const double tf = (double)cus * 8 * (double)iters * mfma_per_iter * flops_per_mfma / ms / 1e9;
  printf("%-52s %7.0f ms  %7.0f TFLOP/s  %6.0f W  %5.0f MHz  -> %.2f TFLOP/s per W  (%d samples)\n", what, ms, tf, ns ? pw / ns : 0.0, ns ? ck / ns : 0.0, ns ? tf / (pw / ns) : 0.0, ns);
  fflush(stdout);
}

int main() {
  int cus = 256;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  float* out; hipMalloc(&out, 4);
  const int it = 6000000;
  run<0>("v_mfma_f32_16x16x32_bf16, random operands", 16.0 * 16 * 32 * 2, 16, it, cus, out);
  run<1>("v_mfma_f32_32x32x16_bf16, random operands", 32.0 * 32 * 16 * 2, 8, it, cus, out);
  run<2>("v_mfma_f32_16x16x32_f16, random operands", 16.0 * 16 * 32 * 2, 16, it, cus, out);
  run<3>("v_mfma_scale_f32_16x16x128_f8f6f4 (fp8), random bits", 16.0 * 16 * 128 * 2, 16, it / 2, cus, out);
  run<4>("v_mfma_f32_16x16x32_bf16, constant operands", 16.0 * 16 * 32 * 2, 16, it, cus, out);
  run<0>("v_mfma_f32_16x16x32_bf16, random operands (again)", 16.0 * 16 * 32 * 2, 16, it, cus, out);
  return 0;
}
