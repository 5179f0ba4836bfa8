// LDS-DMA stream probe (tools/, not product code): the operand stream of the 256x256x64 GEMM tile alone - persistent workgroups walk the
// QKV problem's tiles (M = 20800, N = 2304, K = 768) and bring every K-tile's 64 KiB into LDS with buffer_load_dwordx4 ... lds, in
// different PIECE SHAPES (what one 1-KiB wave-instruction covers in memory):
//   shape 0: 16 rows x 64 B (half of each 128-byte line; the st_16x32 sub-tile of ec_gemm8.hip)
//   shape 1:  8 rows x 128 B (whole lines)
//   shape 2: 16 rows x 64 B, but the two halves of the same lines issued back to back by the same wave (as ec_gemm8.hip does)
// and with 4 or 8 waves per workgroup; a barrier per K-tile, vmcnt throttled to PDP pieces in flight per wave.
//   hipcc --offload-arch=gfx950 -O3 -o tools/dma_probe tools/dma_probe.hip && tools/dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((address_space(3))) void* lptr_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#define DEFINE_PROBE(NAME, NW) \
__global__ __launch_bounds__(NW * 64) void NAME(const char* X, const char* W, int M, int N, int K, int alias, int PSH, int PDP) { \
  extern __shared__ __attribute__((aligned(16))) char smem[]; \
  const int lane = threadIdx.x & 63; \
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); \
  const int ntn = N / 256, ntm = (M + 255) / 256, ntiles = ntm * ntn, nk = K / 64; \
  const unsigned ld = (unsigned)K * 2u; \
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(X), 0, -1, 0x00020000); \
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(W), 0, -1, 0x00020000); \
  constexpr int PPW = 64 / NW;         \
  constexpr int PPO = PPW / 2; \
  const int nxcd = 8, chunk = (ntiles + 7) / 8; \
  const int xcd = blockIdx.x % nxcd, slot = blockIdx.x / nxcd, nslot = gridDim.x / nxcd; \
  const int t_end = min(ntiles, (xcd + 1) * chunk); \
  int it = 0; \
  for (int t = xcd * chunk + slot; t < t_end; t += nslot) { \
    const int m0 = alias ? 0 : (t / ntn) * 256, n0 = alias ? 0 : (t % ntn) * 256; \
    unsigned vx[PPO], vw[PPO]; \
_Pragma("unroll") \
    for (int i = 0; i < PPO; ++i) { \
      int row, cb; \
      if (PSH == 3) {          /* 8 rows x 128 B, lanes 0-31 the first 64 B of the 8 lines, lanes 32-63 the second 64 B */ \
        row = (wave * PPO + i) * 8 + ((lane >> 2) & 7); \
        cb = (lane >> 5) * 64 + (lane & 3) * 16; \
      } else if (PSH == 1) {           \
        row = (wave * PPO + i) * 8 + (lane >> 3); \
        cb = (lane & 7) * 16; \
      } else {                    \
        const int p = wave * PPO + i; \
        const int g = PSH == 2 ? p >> 1 : p % 16, h = PSH == 2 ? p & 1 : p / 16; \
        row = g * 16 + (lane >> 2); \
        cb = h * 64 + (lane & 3) * 16; \
      } \
      vx[i] = (unsigned)min(m0 + row, M - 1) * ld + cb; \
      vw[i] = (unsigned)min(n0 + row, N - 1) * ld + cb; \
    } \
    for (int kt = 0; kt < nk; ++kt, ++it) { \
      char* dst = smem + (it & 1) * 65536 + wave * PPW * 1024; \
_Pragma("unroll") \
      for (int i = 0; i < PPO; ++i) { \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lptr_t)(dst + i * 1024), 16, vx[i], kt * 128, 0, 0); \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lptr_t)(dst + (PPO + i) * 1024), 16, vw[i], kt * 128, 0, 0); \
      } \
      if (PDP == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else if (PDP == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); \
      __builtin_amdgcn_s_barrier(); \
    } \
  } \
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
}
DEFINE_PROBE(probe8, 8)
DEFINE_PROBE(probe4, 4)

template <int NW, int PSH, int PDP>
void run(const char* X, const char* W, int M, int N, int K, int alias) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  void (*kern)(const char*, const char*, int, int, int, int, int, int) = NW == 8 ? probe8 : probe4;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(256), dim3(NW * 64), 131072, 0, X, W, M, N, K, alias, PSH, PDP);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double bytes = (double)((M + 255) / 256) * (N / 256) * (K / 64) * 65536.0;
  printf("waves %d shape %d depth %2d alias %d : %7.1f us  %6.2f TB/s  %5.1f B/clk/CU @2.4GHz\n", NW, PSH, PDP, alias, best * 1e3,
         bytes / (best * 1e-3) / 1e12, bytes / (best * 1e-3) / 256 / 2.4e9);
  fflush(stdout);
}

int main() {
  const int M = 20800, N = 2304, K = 768;
  char *X, *W;
  CK(hipMalloc(&X, (size_t)M * K * 2));
  CK(hipMalloc(&W, (size_t)N * K * 2));
  CK(hipMemset(X, 1, (size_t)M * K * 2));
  CK(hipMemset(W, 1, (size_t)N * K * 2));
  for (int alias = 0; alias < 2; ++alias) {
    run<8, 0, 8>(X, W, M, N, K, alias);
    run<8, 2, 8>(X, W, M, N, K, alias);
    run<8, 1, 8>(X, W, M, N, K, alias);
    run<8, 3, 8>(X, W, M, N, K, alias);
    run<4, 3, 16>(X, W, M, N, K, alias);
    run<8, 0, 0>(X, W, M, N, K, alias);
    run<8, 1, 0>(X, W, M, N, K, alias);
    run<4, 0, 16>(X, W, M, N, K, alias);
    run<4, 2, 16>(X, W, M, N, K, alias);
    run<4, 1, 16>(X, W, M, N, K, alias);
    run<4, 1, 0>(X, W, M, N, K, alias);
  }
  return 0;
}
