// Are scalar atomics (s_atomic_add ... glc) coherent across the eight XCDs of an MI355X, and with agent-scope vector atomics on the same
// address?  256-1024 workgroups each take N tickets from ONE counter - half by s_atomic_add, half by __hip_atomic_fetch_add(agent) -
// and record them; the host checks that every ticket 0 .. total-1 was handed out exactly once.  (The dynamic tile schedule of
// ec_gemm8.hip relies on exactly this.)   hipcc --offload-arch=gfx950 -O2 tools/satomic_probe.hip -o tools/satomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void take(int* ctr, int* out, int n, int mode) {
  const int wg = blockIdx.x;
  if (threadIdx.x != 0) return;
  for (int i = 0; i < n; ++i) {
    int v;
    const bool scalar = mode == 0 || (mode == 2 && ((i + wg) & 1));
    if (scalar) {
      int s = 1;
      asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(s) : "s"(ctr) : "memory");
      v = s;
    } else {
      v = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    out[wg * n + i] = v;
    for (int k = 0; k < (wg % 7) * 50; ++k) asm volatile("s_nop 7");   // de-phase the workgroups
  }
}

int main() {
  int *ctr, *out;
  const int n = 64;
  for (int mode = 0; mode < 3; ++mode)
    for (int grid : {256, 1024, 4096}) {
      hipMalloc(&ctr, 256);
      hipMalloc(&out, (size_t)grid * n * 4);
      hipMemset(ctr, 0, 256);
      for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(take, dim3(grid), dim3(64), 0, 0, ctr, out, n, mode);   // counter runs on across launches
      hipDeviceSynchronize();
      std::vector<int> h((size_t)grid * n);
      int c = 0;
      hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
      hipMemcpy(&c, ctr, 4, hipMemcpyDeviceToHost);
      std::vector<int> seen((size_t)grid * n, 0);
      long bad = 0;
      for (int v : h) { const long t = (long)v - 2L * grid * n; if (t < 0 || t >= (long)grid * n) ++bad; else if (seen[t]++) ++bad; }
      printf("mode %d (%s) grid %4d: counter %d (expected %d), bad tickets in the last launch %ld\n", mode,
             mode == 0 ? "scalar" : mode == 1 ? "vector agent-scope" : "mixed", grid, c, 3 * grid * n, bad);
      hipFree(ctr); hipFree(out);
    }
  return 0;
}
