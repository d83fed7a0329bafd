// Probe (round 6): what does gfx950's v_cvt_scalef32_pk_fp8_f32 compute?  fp8(x / scale) or fp8(x * scale), and does it saturate beyond +-448?
// Prints, for a few inputs and scales, its e4m3 byte next to the bytes of the unscaled v_cvt_pk_fp8_f32 of x * s, x / s and their clamped forms.
//   hipcc --offload-arch=gfx950 -O2 -o cvt_scale_probe tools/probes/cvt_scale.hip && ./cvt_scale_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef short v2i16 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, const float* scales, int ns, unsigned* out) {
  const float x = in[threadIdx.x];
  for (int i = 0; i < ns; ++i) {
    const float s = scales[i];
    v2i16 r = {0, 0};
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, x, x, s, false);
    const unsigned a = (unsigned)__builtin_bit_cast(int, r) & 0xff;
    const unsigned m = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(x * s, x * s, 0, false) & 0xff;
    const unsigned d = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(x / s, x / s, 0, false) & 0xff;
    const float cm = __builtin_amdgcn_fmed3f(x * s, -448.f, 448.f), cd = __builtin_amdgcn_fmed3f(x / s, -448.f, 448.f);
    const unsigned mc = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(cm, cm, 0, false) & 0xff;
    const unsigned dc = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(cd, cd, 0, false) & 0xff;
    out[(threadIdx.x * ns + i) * 5 + 0] = a; out[(threadIdx.x * ns + i) * 5 + 1] = m; out[(threadIdx.x * ns + i) * 5 + 2] = d;
    out[(threadIdx.x * ns + i) * 5 + 3] = mc; out[(threadIdx.x * ns + i) * 5 + 4] = dc;
  }
}
int main() {
  const float xs[] = {0.f, 1e-4f, 0.0019f, 0.3f, 1.0f, 3.3f, 100.f, 447.f, 448.f, 449.f, 1000.f, 1e6f, -0.3f, -1000.f, 3e-7f, 65504.f};
  const float ss[] = {1.0f, 2048.0f, 1.0f / 2048.0f, 0.5f, 3.0f};
  const int nx = sizeof(xs) / 4, ns = sizeof(ss) / 4;
  float *dx, *dsc; unsigned* dout;
  hipMalloc(&dx, sizeof(xs)); hipMalloc(&dsc, sizeof(ss)); hipMalloc(&dout, nx * ns * 5 * 4);
  hipMemcpy(dx, xs, sizeof(xs), hipMemcpyHostToDevice); hipMemcpy(dsc, ss, sizeof(ss), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(nx), 0, 0, dx, dsc, ns, dout);
  unsigned out[16 * 5 * 5];
  hipMemcpy(out, dout, nx * ns * 5 * 4, hipMemcpyDeviceToHost);
  printf("x, scale: scalef32 | cvt(x*s) cvt(x/s) | cvt(clamp(x*s)) cvt(clamp(x/s))\n");
  for (int i = 0; i < nx; ++i)
    for (int j = 0; j < ns; ++j) {
      const unsigned* o = out + (i * ns + j) * 5;
      printf("%12g %10g: %02x | %02x %02x | %02x %02x  %s\n", xs[i], ss[j], o[0], o[1], o[2], o[3], o[4], o[0] == o[4] ? "= clamp(x/s)" : (o[0] == o[3] ? "= clamp(x*s)" : (o[0] == o[2] ? "= x/s" : (o[0] == o[1] ? "= x*s" : "?"))));
    }
  return 0;
}
