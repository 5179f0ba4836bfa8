// Issue-cost probe for the softmax design of the attention kernel (tools/, not product code): cycles per loop iteration of
//   M x v_mfma_f32_32x32x16_f16  +  F x v_fma_f32  +  P x v_pk_fma_f32  +  E x v_exp_f32      (all independent of each other)
// with one, two or three waves per SIMD (workgroup span per iteration: all waves run the same loop), every instruction an asm volatile statement (program order = the order written: MFMA first,
// then the fillers interleaved E, F, P round-robin).  Answers: what a transcendental costs, and how much vector work hides
// under one MFMA inside ONE wave and across the waves of a SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_mfma_probe tools/valu_mfma_probe.hip && tools/valu_mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// X (round 4): groups of the transcendental-free exp2 the attention softmax could use instead of v_exp_f32 - per score
//   x = fma(s, c, -m); i = cvt_flr_i32(x); r = fract(x); p = ((c3 r + c2) r + c1) r + 1; e = ldexp(p, i)   (7 plain VALU instructions)
template <int M, int F, int P, int E, int SPLIT, int X = 0>
__global__ void probe(float* out, unsigned* cyc, int iters) {
  f32x16 acc[2];
  for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f); b[i] = (_Float16)(i * 0.01f); }
  float f[8];
  f32x2 pk[8];
  float e[8];
  for (int i = 0; i < 8; ++i) { f[i] = threadIdx.x * 1e-3f + i; pk[i] = f32x2{f[i], f[i] + 1.f}; e[i] = -f[i] * 0.01f; }
  const float c1 = 0.999f, c2 = 1e-4f;
  const f32x2 p1 = {0.999f, 0.999f}, p2 = {1e-4f, 1e-4f};
  __shared__ unsigned t0s;
  if (threadIdx.x == 0) t0s = (unsigned)__builtin_amdgcn_s_memtime();
  __syncthreads();
  const unsigned t0 = t0s;
  for (int it = 0; it < iters; ++it) {
    // SPLIT = 0: all MFMAs first, then all fillers; SPLIT = 1: fillers dealt out evenly behind each MFMA
#pragma unroll
    for (int m = 0; m < (M ? M : 1); ++m) {
      if (M) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[m & 1]) : "v"(a), "v"(b));
      const int lo = SPLIT ? m : 0, step = SPLIT ? (M ? M : 1) : 1;
      if (!SPLIT && m + 1 < (M ? M : 1)) continue;
#pragma unroll
      for (int k = lo; k < 64; k += step) {
        if (k < E) asm volatile("v_exp_f32 %0, %0" : "+v"(e[k & 7]));
        if (k < F) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[k & 7]) : "v"(c1), "v"(c2));
        if (k < -F) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(f[k & 7]));   // F < 0: one-source filler
        if (k < P) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pk[k & 7]) : "v"(p1), "v"(p2));
        if (k < X) {
          float x, r, pp; int xi;
          asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x) : "v"(f[k & 7]), "v"(c1), "v"(c2));
          asm volatile("v_cvt_flr_i32_f32 %0, %1" : "=v"(xi) : "v"(x));
          asm volatile("v_fract_f32 %0, %1" : "=v"(r) : "v"(x));
          asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(pp) : "v"(r), "v"(c2), "v"(c1));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(pp) : "v"(r), "v"(c1));
          asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(pp) : "v"(r));
          asm volatile("v_ldexp_f32 %0, %1, %2" : "=v"(f[k & 7]) : "v"(pp), "v"(xi));
        }
      }
    }
  }
  const unsigned t1 = (unsigned)__builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[0][i] + acc[1][i];
  for (int i = 0; i < 8; ++i) s += f[i] + pk[i][0] + pk[i][1] + e[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  // the span of the whole workgroup (the oldest wave of a SIMD wins the issue arbitration: its own loop time says nothing about the others)
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) atomicMax(cyc, t1 - t0);
}

template <int M, int F, int P, int E, int SPLIT, int X = 0>
void run(float* out, unsigned* cyc, const char* tag) {
  const int iters = 2000;
  printf("%-34s M=%d F=%2d P=%2d E=%2d X=%2d split=%d :", tag, M, F, P, E, X, SPLIT);
  for (int waves : {4, 8, 12}) {   // per CU: 1, 2, 3 waves per SIMD
    unsigned best = ~0u;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(cyc, 0, 4));
      hipLaunchKernelGGL((probe<M, F, P, E, SPLIT, X>), dim3(256), dim3(waves * 64), 0, 0, out, cyc, iters);
      CK(hipDeviceSynchronize());
      unsigned h;
      CK(hipMemcpy(&h, cyc, 4, hipMemcpyDeviceToHost));
      if (h < best) best = h;
    }
    printf("  %dw/SIMD %7.1f", waves / 4, (double)best / iters);
  }
  printf("\n");
  fflush(stdout);
}

int main() {
  float* out;
  unsigned* cyc;
  CK(hipMalloc(&out, 256 * 768 * 4));
  CK(hipMalloc(&cyc, 4));
  run<1, 0, 0, 0, 0>(out, cyc, "MFMA alone");
  run<2, 0, 0, 0, 0>(out, cyc, "2 MFMA");
  run<0, 8, 0, 0, 0>(out, cyc, "8 fma");
  run<0, -8, 0, 0, 0>(out, cyc, "8 add (1 vgpr src)");
  run<0, 16, 0, 0, 0>(out, cyc, "16 fma");
  run<0, -16, 0, 0, 0>(out, cyc, "16 add");
  run<0, 0, 8, 0, 0>(out, cyc, "8 pk_fma");
  run<0, 0, 0, 8, 0>(out, cyc, "8 exp");
  run<0, 8, 0, 8, 0>(out, cyc, "8 fma + 8 exp");
  run<1, 4, 0, 0, 0>(out, cyc, "MFMA + 4 fma");
  run<1, 6, 0, 0, 0>(out, cyc, "MFMA + 6 fma");
  run<1, 8, 0, 0, 0>(out, cyc, "MFMA + 8 fma");
  run<1, 12, 0, 0, 0>(out, cyc, "MFMA + 12 fma");
  run<1, 0, 0, 2, 0>(out, cyc, "MFMA + 2 exp");
  run<1, 0, 0, 4, 0>(out, cyc, "MFMA + 4 exp");
  run<1, 0, 0, 8, 0>(out, cyc, "MFMA + 8 exp");
  run<1, 4, 0, 4, 0>(out, cyc, "MFMA + 4 fma + 4 exp");
  run<1, 2, 2, 4, 0>(out, cyc, "MFMA + 2 fma + 2 pk + 4 exp");
  run<8, 32, 0, 0, 0>(out, cyc, "8 MFMA then 32 fma (serial)");
  run<8, 32, 0, 0, 1>(out, cyc, "8 x (MFMA + 4 fma)");
  run<8, 16, 16, 16, 0>(out, cyc, "8 MFMA then 16f 16p 16e");
  run<8, 16, 16, 16, 1>(out, cyc, "8 x (MFMA + 2f 2p 2e)");
  run<8, 24, 8, 16, 1>(out, cyc, "8 x (MFMA + 3f 1p 2e)");
  // round 4: how much PLAIN vector work hides under the MFMAs, and the transcendental-free exp2
  run<1, 16, 0, 0, 0>(out, cyc, "MFMA + 16 fma");
  run<1, 20, 0, 0, 0>(out, cyc, "MFMA + 20 fma");
  run<1, 24, 0, 0, 0>(out, cyc, "MFMA + 24 fma");
  run<0, 0, 0, 0, 0, 2>(out, cyc, "2 poly-exp2 groups (14 VALU)");
  run<1, 0, 0, 0, 0, 2>(out, cyc, "MFMA + 2 poly-exp2 groups");
  run<1, 6, 0, 0, 0, 2>(out, cyc, "MFMA + 2 poly-exp2 + 6 fma");
  run<1, 6, 0, 2, 0>(out, cyc, "MFMA + 2 exp + 6 fma");
  run<8, 48, 0, 0, 1, 16>(out, cyc, "8 x (MFMA + 2 poly-exp2 + 6 fma)");
  run<8, 48, 0, 16, 1>(out, cyc, "8 x (MFMA + 2 exp + 6 fma)");
  return 0;
}
