// Probe (tools/, not product code): WHICH (lane, byte) of the scale register does v_mfma_scale_f32_16x16x128_f8f6f4 read for WHICH (row, 32-element k block)?
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_scale_layout_probe.hip -o tools/mfma_scale_layout_probe && tools/mfma_scale_layout_probe
// tools/fp8_mfma_probe.hip assumed "lane l = (row l & 15, k block l >> 4) reads byte op_sel of ITS OWN scale register" and found 255 of 256 mismatches
// (0 with uniform scales, which is all the fp16x2 mode needs).  The FP6 lead of DESIGN.md section 10 needs the real layout.  Method: all data = 1.0
// (e4m3 x e5m2), every scale byte 127 (x 1) -> every output is 128; then ONE byte of ONE lane's A-scale (or B-scale) register is set to 128 (x 2) and the
// outputs that move say which row (A) / column (B) and - by how much: + 32 per affected 32-element block - that byte scales, for op_sel = 0 .. 3.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int OPA, int OPB>
__global__ void k(const int* sa, const int* sb, float* out) {
  const int l = threadIdx.x;
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = 0x38383838; b[i] = 0x3c3c3c3c; }   // 1.0 in e4m3 / e5m2
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 1, OPA, sa[l], OPB, sb[l]);
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = acc[e];
}

template <int OP>
static void run(bool side_b, int* dsa, int* dsb, float* dout) {
  printf("op_sel %d, %s-operand scale register: which outputs move when byte B of lane L is doubled\n", OP, side_b ? "B (second)" : "A (first)");
  int hits = 0;
  for (int byte = 0; byte < 4; ++byte)
    for (int L = 0; L < 64; ++L) {
      std::vector<int> s(64, 0x7f7f7f7f), u(64, 0x7f7f7f7f);
      s[L] = (s[L] & ~(0xff << (8 * byte))) | (128 << (8 * byte));
      CK(hipMemcpy(side_b ? dsb : dsa, s.data(), 256, hipMemcpyHostToDevice));
      CK(hipMemcpy(side_b ? dsa : dsb, u.data(), 256, hipMemcpyHostToDevice));
      if (side_b) hipLaunchKernelGGL((k<0, OP>), 1, 64, 0, 0, dsa, dsb, dout); else hipLaunchKernelGGL((k<OP, 0>), 1, 64, 0, 0, dsa, dsb, dout);
      std::vector<float> o(256);
      CK(hipMemcpy(o.data(), dout, 1024, hipMemcpyDeviceToHost));
      // C layout: lane l, reg e -> row (l >> 4) * 4 + e, col l & 15
      int rows[16] = {0}, cols[16] = {0};
      float delta = 0.f;
      int moved = 0;
      for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 4; ++e)
          if (o[l * 4 + e] != 128.f) { ++moved; rows[(l >> 4) * 4 + e]++; cols[l & 15]++; delta = o[l * 4 + e] - 128.f; }
      if (!moved) continue;
      ++hits;
      int r = -1, c = -1, nr = 0, nc = 0;
      for (int i = 0; i < 16; ++i) { if (rows[i]) { r = i; ++nr; } if (cols[i]) { c = i; ++nc; } }
      if (hits <= 70)
        printf("  lane %2d byte %d: %3d outputs moved by %+g (= %g blocks of 32): %s\n", L, byte, moved, delta, delta / 32.f,
               nr == 1 ? (std::string("row ") + std::to_string(r) + " (all columns)").c_str() : nc == 1 ? (std::string("column ") + std::to_string(c) + " (all rows)").c_str() : "several rows and columns");
    }
  printf("  -> %d of 256 (lane, byte) positions are read\n", hits);
}

// Second question: does a lane's scale apply to THAT lane's 32 data bytes?  Data = 1.0 only in the registers of lane group gd (zero elsewhere), the
// A-scale of every lane of group gs doubled: the outputs (32 without any scaling) move to 64 exactly when the scaled block is the one holding the data.
template <int DUMMY>
__global__ void k2(const int* sa, float* out, int gd) {
  const int l = threadIdx.x;
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (l >> 4) == gd ? 0x38383838 : 0; b[i] = 0x3c3c3c3c; }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 1, 0, sa[l], 0, 0x7f7f7f7f);
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = acc[e];
}

int main() {
  {
    int* dsa; float* dout;
    CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dout, 1024));
    printf("data only in lane group gd, A-scale doubled on lane group gs: output value (32 = unscaled, 64 = the scaled block is the one with the data)\n        gs=0  gs=1  gs=2  gs=3\n");
    for (int gd = 0; gd < 4; ++gd) {
      printf("  gd=%d ", gd);
      for (int gs = 0; gs < 4; ++gs) {
        std::vector<int> s(64, 0x7f7f7f7f);
        for (int l = 16 * gs; l < 16 * gs + 16; ++l) s[l] = 0x7f7f7f80;
        CK(hipMemcpy(dsa, s.data(), 256, hipMemcpyHostToDevice));
        hipLaunchKernelGGL((k2<0>), 1, 64, 0, 0, dsa, dout, gd);
        std::vector<float> o(256);
        CK(hipMemcpy(o.data(), dout, 1024, hipMemcpyDeviceToHost));
        float mn = o[0], mx = o[0];
        for (float v : o) { mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
        printf(mn == mx ? "  %4g" : "  %g..%g", mn, mx);
      }
      printf("\n");
    }
  }

  int *dsa, *dsb; float* dout;
  CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dout, 1024));
  run<0>(false, dsa, dsb, dout); run<1>(false, dsa, dsb, dout); run<2>(false, dsa, dsb, dout); run<3>(false, dsa, dsb, dout);
  run<0>(true, dsa, dsb, dout); run<1>(true, dsa, dsb, dout);
  return 0;
}
