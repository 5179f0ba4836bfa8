// Probe: semantics of ds_read_b64_tr_b16 on gfx950 (run on the GPU box; prints the lane/element map).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(uint16_t* out, int mode) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;   // value = element index
  __syncthreads();
  const int l = threadIdx.x;
  unsigned addr;
  const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)lds;
  if (mode == 0) addr = l * 8;                                  // lane-linear 8-byte pieces
  else addr = ((l >> 2) & 3) * 128 + (l & 3) * 8 + (l >> 4) * 32;  // 16-lane group g: rows (i>>2) stride 128 B (64 elems), quad (i&3), group col offset 16 elems
  unsigned long long v;
  addr += base;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)(v >> (16 * j));
}
int main() {
  uint16_t* d; printf("malloc: %s\n", hipGetErrorString(hipMalloc(&d, 64 * 4 * 2)));
  uint16_t h[256];
  for (int mode = 0; mode < 2; ++mode) {
    probe<<<1, 64>>>(d, mode);
    printf("launch: %s\n", hipGetErrorString(hipGetLastError()));
    printf("sync: %s\n", hipGetErrorString(hipDeviceSynchronize()));
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  }
  return 0;
}
