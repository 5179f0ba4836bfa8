// MFMA power probe (tools/, not product code): sustained TFLOP/s of v_mfma_f32_16x16x32_f16 vs v_mfma_f32_32x32x16_f16 with every CU
// busy (2 waves per SIMD, register operands only, no memory traffic in the loop), on RANDOM and on ZERO operand data.  The chip clocks
// to its power budget (MI355X_MICROARCH.md "DVFS give-back"): the form that reads fewer operand registers per flop may sustain more.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_power_probe tools/mfma_power_probe.hip && tools/mfma_power_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int FORM>
__global__ __launch_bounds__(512) void probe(const f16x8* src, float* sink, int iters) {
  // 8 A and 8 B operand fragments per lane (64 VGPRs), as a K-tile's worth of fragments would be
  f16x8 a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = src[(threadIdx.x + i * 512) % 4096]; b[i] = src[(threadIdx.x * 7 + i * 131 + 17) % 4096]; }
  if constexpr (FORM == 16) {
    f32x4 acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 7], b[(i >> 3) + (i & 3)], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][3];
    if (s == 12345.678f) sink[0] = s;
  } else {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + r) & 7], b[(i * 3 + r) & 7], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15];
    if (s == 12345.678f) sink[0] = s;
  }
}

int main() {
  int ncu = 0;
  CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
  std::vector<_Float16> h(4096 * 8);
  f16x8* d; float* sink;
  CK(hipMalloc(&d, h.size() * 2)); CK(hipMalloc(&sink, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int zero = 0; zero < 2; ++zero) {
    srand(1);
    for (auto& v : h) v = zero ? (_Float16)0.f : (_Float16)((rand() % 2001 - 1000) / 500.0f);
    CK(hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    for (int form : {16, 32, 16, 32}) {
      const int iters = 20000;
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        if (form == 16) hipLaunchKernelGGL(probe<16>, dim3(ncu), dim3(512), 0, 0, d, sink, iters);
        else hipLaunchKernelGGL(probe<32>, dim3(ncu), dim3(512), 0, 0, d, sink, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        // flops per wave per iteration: 32 x 16384 (16x16x32) = 16 x 32768 (32x32x16) = 524288
        const double fl = (double)ncu * 8 * iters * 524288.0;
        if (rep) printf("%s operands, %dx%d form: %.2f ms  %.0f TFLOP/s (%.3f of 2500)\n", zero ? "zero  " : "random", form, form, ms, fl / ms / 1e9, fl / ms / 1e9 / 2500);
      }
    }
  }
  return 0;
}
