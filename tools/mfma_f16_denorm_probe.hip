// Probe (tools/, not product code): do the gfx950 16-bit MFMAs keep fp16 SUBNORMAL operands, or flush them to zero?
// An fp16 hi + lo split of a weight of magnitude 2^-5 has its lo part at <= 2^-17: below the smallest fp16 normal (2^-14), i.e. an fp16
// subnormal with 2^-24 spacing.  If the matrix pipe flushed subnormal inputs, an "fp16x3" form of the conforming mode (5 x more accurate
// than bf16x3 in oracle/correction_terms_study.py) would lose exactly those terms; bf16 has the fp32 exponent range and no such question.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f16_denorm_probe tools/mfma_f16_denorm_probe.hip && /tmp/mfma_f16_denorm_probe
// Each case: A[16 x 32] = a everywhere, B[32 x 16] = b everywhere  ->  every C element = 32 a b (exact in fp32), also with the MODE
// register's fp16 denormal bits cleared (FP_DENORM for 64/16-bit = MODE[7:6]) to see whether the matrix pipe looks at them at all.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// (the operands arrive as BIT PATTERNS converted on the host: an in-kernel v_cvt_f16_f32 would itself obey the MODE bits)
__global__ void probe(const unsigned short* ab16, const unsigned short* abb, float* out, int n, int clear_denorm_mode) {
  if (clear_denorm_mode) __builtin_amdgcn_s_setreg(1 | (6 << 6) | (1 << 11), 0);   // hwreg(HW_REG_MODE, offset 6, size 2) = 0: flush 64/16-bit denormals on the VALU
  for (int c = 0; c < n; ++c) {
    const _Float16 a = __builtin_bit_cast(_Float16, ab16[2 * c]), b = __builtin_bit_cast(_Float16, ab16[2 * c + 1]);
    f16x8 va, vb;
    for (int e = 0; e < 8; ++e) { va[e] = a; vb[e] = b; }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(va, vb, acc, 0, 0, 0);
    const __bf16 a2 = __builtin_bit_cast(__bf16, abb[2 * c]), b2 = __builtin_bit_cast(__bf16, abb[2 * c + 1]);
    bf16x8 wa, wb;
    for (int e = 0; e < 8; ++e) { wa[e] = a2; wb[e] = b2; }
    f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, wb, acc2, 0, 0, 0);
    if (threadIdx.x == 0) { out[2 * c] = acc[0]; out[2 * c + 1] = acc2[0]; }
  }
}

int main() {
  // (a, b): a subnormal in fp16 for the first rows (2^-15 .. 2^-24), normal for the last two
  const float h[] = {3.0517578125e-05f, 1024.f,        // 2^-15 (subnormal, 512 ulps of 2^-24)
                     9.5367431640625e-07f, 1024.f,     // 2^-20
                     5.9604644775390625e-08f, 1024.f,  // 2^-24: the smallest subnormal
                     7.62939453125e-06f, 0.03125f,     // 2^-17 (a weight's lo part) x 2^-5
                     1024.f, 9.5367431640625e-07f,     // the subnormal on the B side
                     6.103515625e-05f, 1024.f,         // 2^-14: the smallest normal
                     0.5f, 0.25f};
  const int n = sizeof(h) / sizeof(h[0]) / 2;
  unsigned short h16[64], hbf[64];
  for (int i = 0; i < 2 * n; ++i) {
    const _Float16 v = (_Float16)h[i];                      // host conversion: IEEE, subnormals kept (every value here is exact in fp16)
    h16[i] = __builtin_bit_cast(unsigned short, v);
    hbf[i] = (unsigned short)(__builtin_bit_cast(unsigned, h[i]) >> 16);   // exact in bf16 as well (powers of two)
  }
  unsigned short *d, *db; float* o;
  hipMalloc(&d, sizeof(h16)); hipMalloc(&db, sizeof(hbf)); hipMalloc(&o, 2 * n * sizeof(float));
  hipMemcpy(d, h16, sizeof(h16), hipMemcpyHostToDevice);
  hipMemcpy(db, hbf, sizeof(hbf), hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, db, o, n, mode);
    float r[64];
    hipMemcpy(r, o, 2 * n * sizeof(float), hipMemcpyDeviceToHost);
    printf("%s\n", mode ? "MODE fp16/fp64 denormal bits CLEARED:" : "default MODE:");
    for (int c = 0; c < n; ++c)
      printf("  a = %-14.8g b = %-14.8g  expected 32ab = %-14.8g  mfma f16 = %-14.8g (%s)   mfma bf16 = %-14.8g\n", h[2 * c], h[2 * c + 1],
             32.0 * h[2 * c] * h[2 * c + 1], r[2 * c], r[2 * c] == 32.f * h[2 * c] * h[2 * c + 1] ? "kept" : (r[2 * c] == 0.f ? "FLUSHED" : "differs"), r[2 * c + 1]);
  }
  return 0;
}
