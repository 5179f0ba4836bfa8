// Probe (tools/, not product code): does MODE.FP16_OVFL (hwreg MODE bit 23) make the gfx950 fp32 -> fp16 conversions saturate at
// +-65504 while keeping true infinities and NaN?  (v_cvt_f16_f32, v_cvt_pk_f16_f32, v_cvt_pkrtz_f16_f32)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fp16_ovfl_probe tools/fp16_ovfl_probe.hip && /tmp/fp16_ovfl_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void probe(const float* in, unsigned short* out, int n, int ovfl) {
  if (ovfl) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);   // hwreg(HW_REG_MODE, 23, 1) = 1
  const int i = threadIdx.x;
  if (i < n) {
    const float v = in[i];
    out[i] = __builtin_bit_cast(unsigned short, (_Float16)v);                                    // v_cvt_f16_f32
    const h2 p = __builtin_convertvector(f2{v, -v}, h2);                                         // v_cvt_pk_f16_f32
    out[n + i] = __builtin_bit_cast(unsigned short, p[0]);
    out[2 * n + i] = __builtin_bit_cast(unsigned short, p[1]);
    const _Float16 s = (_Float16)v * (_Float16)4.0f;                                             // an fp16 VALU result that overflows
    out[3 * n + i] = __builtin_bit_cast(unsigned short, s);
  }
}
int main() {
  const float h[] = {1.0f, 65504.f, 65519.9f, 65520.f, 70000.f, 1e6f, 3e38f, INFINITY, -INFINITY, NAN, -1e6f, 6e-8f, 2e-8f, 30000.f};
  const int n = sizeof(h) / sizeof(h[0]);
  float* d; unsigned short* o;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, 4 * n * 2);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  for (int ovfl = 0; ovfl < 2; ++ovfl) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o, n, ovfl);
    unsigned short r[4 * 32];
    hipMemcpy(r, o, 4 * n * 2, hipMemcpyDeviceToHost);
    printf("FP16_OVFL=%d\n", ovfl);
    for (int i = 0; i < n; ++i) printf("  %14g -> cvt %04x  cvt_pk(v) %04x  cvt_pk(-v) %04x  (half)v*4 %04x\n", h[i], r[i], r[n + i], r[2 * n + i], r[3 * n + i]);
  }
  return 0;
}
