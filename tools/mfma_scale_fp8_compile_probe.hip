// Compile probe (tools/, not product code): is the block-scaled FP8 MFMA of gfx950 reachable from this hipcc?
//   hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only tools/mfma_scale_fp8_compile_probe.hip -o - | grep v_mfma_scale
// -> v_mfma_scale_f32_16x16x128_f8f6f4 v[2:5], v[2:9], v[10:17], 0, v1, v18 op_sel_hi:[0,0,0]      (ROCm 7.2.0, checked in round 5)
// Operands: A / B = 8 dwords per lane (32 bytes: this lane's share of a 16 x 128 FP8 tile), cbsz / blgp = the formats of A / B (0 = e4m3),
// the two scale operands = E8M0 exponents (one per 32 K-elements) packed in a VGPR, selected by the op_sel arguments.  Twice the flops
// per instruction of v_mfma_f32_16x16x32_f16 at the same issue cost: what DESIGN.md section 10 (b) would build the correction terms
// a_lo W_hi + a_hi W_lo of the conforming mode on.  Never run this round - numerics are emulated in oracle/correction_terms_study.py.
#include <hip/hip_runtime.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const i32x8* a, const i32x8* b, float* out, int scale_a, int scale_b) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0, scale_a, 0, scale_b);
  out[threadIdx.x * 4] = acc[0] + acc[1] + acc[2] + acc[3];
}
int main() { return 0; }
