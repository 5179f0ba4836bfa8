// Store-path probe for the GEMM epilogue design (tools/, not product code): persistent workgroups alternate a busy-wait of D shader
// cycles (the "K loop") with the 16-byte-per-lane stores of one 256x256 16-bit output tile, in different lane->address patterns,
// in lockstep or staggered over the workgroups.  Prints the kernel time and the store-issue cycles per wave.
//   hipcc --offload-arch=gfx950 -O3 -o tools/store_probe tools/store_probe.hip && tools/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// pattern: 0 = 8 rows x 128 B per wave-instruction, 1 = 16 rows x 64 B, 2 = 32 rows x 32 B, 3 = 64 rows x 16 B
template <int NW>
__global__ __launch_bounds__(NW * 64) void probe(char* C, int M, int N, long ldc2, int pattern, long delay, int stagger, int nt_hint,
                                                  unsigned long long* cyc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ntn = N / 256, ntm = (M + 255) / 256, ntiles = ntm * ntn;
  // wave sub-tile: NW = 8: 2 x 4 waves of 128 rows x 64 columns (128 B); NW = 4: 2 x 2 waves of 128 rows x 128 columns (256 B)
  const int wr = NW == 8 ? wave >> 2 : wave >> 1, wc = NW == 8 ? wave & 3 : wave & 1;
  const int wbytes = NW == 8 ? 128 : 256;
  const int nst = NW == 8 ? 16 : 32;
  unsigned long long issue = 0;
  if (stagger) {
    const long d0 = delay * (blockIdx.x % stagger) / stagger;
    const long t0 = __builtin_amdgcn_s_memtime();
    while ((long)__builtin_amdgcn_s_memtime() - t0 < d0) __builtin_amdgcn_s_sleep(2);
  }
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const long t0 = __builtin_amdgcn_s_memtime();
    while ((long)__builtin_amdgcn_s_memtime() - t0 < delay) __builtin_amdgcn_s_sleep(2);
    const int m0 = (t / ntn) * 256, n0 = (t % ntn) * 256;
    const long base = (long)(m0 + wr * 128) * ldc2 + (long)n0 * 2 + wc * wbytes;
    const u32x4 v = {(unsigned)t, (unsigned)lane, 3u, 4u};
    const long s0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < nst; ++i) {
      long off;
      int rowi;                    // row of the store inside the wave's sub-tile
      if (pattern == 0) {          // rows of 128 B: lane>>3 = row, lane&7 = chunk; NW = 4: two column halves
        const int half = NW == 8 ? 0 : i & 1, ii = NW == 8 ? i : i >> 1;
        rowi = ii * 8 + (lane >> 3);
        off = (long)rowi * ldc2 + half * 128 + (lane & 7) * 16;
      } else if (pattern == 1) {   // 16 rows x 64 B
        const int per = wbytes / 64;
        rowi = (i / per) * 16 + (lane & 15);
        off = (long)rowi * ldc2 + (i % per) * 64 + (lane >> 4) * 16;
      } else if (pattern == 2) {   // 32 rows x 32 B
        const int per = wbytes / 32;
        rowi = (i / per) * 32 + (lane & 31);
        off = (long)rowi * ldc2 + (i % per) * 32 + (lane >> 5) * 16;
      } else {                     // 64 rows x 16 B
        const int per = wbytes / 16;
        rowi = (i / per) * 64 + lane;
        off = (long)rowi * ldc2 + (i % per) * 16;
      }
      const int row = m0 + wr * 128 + rowi;
      if (row < M) {
        if (nt_hint) __builtin_nontemporal_store(v, (u32x4*)(C + base + off));
        else *(u32x4*)(C + base + off) = v;
      }
    }
    issue += __builtin_amdgcn_s_memtime() - s0;
  }
  if (lane == 0) atomicAdd(cyc, issue);
}

int main() {
  const int M = 20800, N = 2304;
  const long ldc2 = (long)N * 2;
  char* C;
  unsigned long long* cyc;
  CK(hipMalloc(&C, (size_t)(M + 256) * ldc2));
  CK(hipMalloc(&cyc, 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const long delays[] = {0, 20000, 45000};
  for (int nw : {8, 4})
    for (int grid : {1, 8, 256})
      for (long delay : delays)
        for (int stagger : {0, 4})
          for (int pattern : {0, 1, 2, 3})
            for (int nt : {0}) {
              if (stagger && delay == 0) continue;
              if (grid != 256 && (delay == 20000 || stagger)) continue;
              float best = 1e9f;
              unsigned long long hc = 0;
              for (int rep = 0; rep < 4; ++rep) {
                CK(hipMemset(cyc, 0, 8));
                CK(hipEventRecord(e0));
                if (nw == 8) hipLaunchKernelGGL(probe<8>, dim3(grid), dim3(512), 0, 0, C, M, N, ldc2, pattern, delay, stagger, nt, cyc);
                else hipLaunchKernelGGL(probe<4>, dim3(grid), dim3(256), 0, 0, C, M, N, ldc2, pattern, delay, stagger, nt, cyc);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
                CK(hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost));
              }
              const int ntiles = 82 * 9;
              const int rounds = (ntiles + grid - 1) / grid;
              // store-issue cycles per wave per tile
              const double per_tile = (double)hc / (double)(ntiles * nw);
              printf("waves %d grid %3d delay %5ld stagger %d pattern %d : %8.1f us  (%d tile rounds, pure delay %.1f us @2.4GHz) issue %.0f cyc/wave/tile = %.1f cyc/store\n",
                     nw, grid, delay, stagger, pattern, best * 1e3, rounds, rounds * delay / 2400.0, per_tile, per_tile / (nw == 8 ? 16 : 32));
              fflush(stdout);
            }
  return 0;
}
