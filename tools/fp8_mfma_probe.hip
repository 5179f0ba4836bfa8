// Probe (tools/, not product code): the semantics the fp16x2 backbone mode relies on, measured on gfx950 before the kernels were written.
//   hipcc --offload-arch=gfx950 -O2 tools/fp8_mfma_probe.hip -o tools/fp8_mfma_probe && tools/fp8_mfma_probe
// (1) v_cvt_pk_bf8_f32 / v_cvt_pk_fp8_f32 / v_cvt_scalef32_pk_*: rounding, saturation, NaN / inf, direction of the scale.
// (2) v_mfma_scale_f32_16x16x128_f8f6f4 with A = e4m3 (cbsz 0), B = e5m2 (blgp 1): which K elements a lane's 32 bytes are, the C layout,
//     what the E8M0 scale operands do (per lane? which byte under op_sel?).
// (3) issue cost of the FP8 MFMA against v_mfma_f32_16x16x32_f16 (one wave per SIMD, dependent-free chains).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void cvt_kernel(const float* in, int n, uint32_t* out, float scale) {
  int i = threadIdx.x;
  if (i >= n) return;
  float a = in[i];
  int r = 0;
  r = __builtin_amdgcn_cvt_pk_bf8_f32(a, a, r, false);
  out[i * 4 + 0] = (uint32_t)r & 0xff;
  r = 0;
  r = __builtin_amdgcn_cvt_pk_fp8_f32(a, a, r, false);
  out[i * 4 + 1] = (uint32_t)r & 0xff;
  s16x2 o = {0, 0};
  o = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(o, a, a, scale, false);
  out[i * 4 + 2] = (uint32_t)__builtin_bit_cast(int, o) & 0xff;
  o = s16x2{0, 0};
  o = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(o, a, a, scale, false);
  out[i * 4 + 3] = (uint32_t)__builtin_bit_cast(int, o) & 0xff;
}

// one wave: lane l supplies 32 bytes of A (a[l]) and of B (b[l]), scale registers sa[l], sb[l]
template <int OPA, int OPB>
__global__ void mfma_kernel(const i32x8* a, const i32x8* b, const int* sa, const int* sb, float* out) {
  const int l = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 1, OPA, sa[l], OPB, sb[l]);
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = acc[e];
}

template <int MODE>
__global__ void rate_kernel(long long* cyc, float* sink, int iters) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = 0x38383838 + threadIdx.x * 0x01010101 * (i & 1); b[i] = 0x3c3c3c3c ^ (i * 0x00010000); }   // non-zero, varied
  f16x8 ha = __builtin_bit_cast(f16x8, __builtin_shufflevector(a, a, 0, 1, 2, 3)), hb = __builtin_bit_cast(f16x8, __builtin_shufflevector(b, b, 0, 1, 2, 3));
  __syncthreads();
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[i], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 0, 1, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
  }
  long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

static float e4m3_dec(uint8_t v) {
  int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float f;
  if (e == 15 && m == 7) return NAN;
  if (e == 0) f = ldexpf((float)m, -9); else f = ldexpf(1.f + m / 8.f, e - 7);
  return s ? -f : f;
}
static float e5m2_dec(uint8_t v) {
  int s = v >> 7, e = (v >> 2) & 31, m = v & 3;
  float f;
  if (e == 31) return m ? NAN : (s ? -INFINITY : INFINITY);
  if (e == 0) f = ldexpf((float)m, -16); else f = ldexpf(1.f + m / 4.f, e - 15);
  return s ? -f : f;
}
static uint8_t e4m3_enc_int(int x) {   // small integers |x| <= 16, exact
  for (int v = 0; v < 256; ++v) if (e4m3_dec((uint8_t)v) == (float)x && !(x == 0 && v != 0)) return (uint8_t)v;
  return 0;
}
static uint8_t e5m2_enc_int(int x) {   // |x| <= 8 exact (and 10, 12, ...)
  for (int v = 0; v < 256; ++v) if (e5m2_dec((uint8_t)v) == (float)x && !(x == 0 && v != 0)) return (uint8_t)v;
  return 0;
}

int main() {
  // ---------------- (1) conversions
  std::vector<float> vals = {0.f, 1.f, 1.0625f, 1.125f, 1.1875f, 1.25f, 1.375f, 1.4375f, 1.5625f, 1.625f, 1.875f, 1.9375f, -1.125f, -1.375f, 448.f, 464.f, 480.f, 500.f, 1000.f,
                             57344.f, 60000.f, 61440.f, 61441.f, 65504.f, 1e6f, INFINITY, -INFINITY, NAN, ldexpf(1.f, -16), ldexpf(1.f, -17), ldexpf(1.5f, -17), ldexpf(1.f, -18),
                             ldexpf(1.f, -9), ldexpf(1.f, -10), ldexpf(1.5f, -10), ldexpf(1.f, -6), 0.3f, 3.3f, -7.7f, 100.f};
  const int n = (int)vals.size();
  float* din; uint32_t* dout;
  CK(hipMalloc(&din, n * 4)); CK(hipMalloc(&dout, n * 16));
  CK(hipMemcpy(din, vals.data(), n * 4, hipMemcpyHostToDevice));
  for (float scale : {1.0f, 4.0f, 0.25f}) {
    hipLaunchKernelGGL(cvt_kernel, 1, 64, 0, 0, din, n, dout, scale);
    std::vector<uint32_t> o(n * 4);
    CK(hipMemcpy(o.data(), dout, n * 16, hipMemcpyDeviceToHost));
    printf("conversions (scale operand of the scalef32 forms = %g):\n   %14s | cvt_pk_bf8 (e5m2) | cvt_pk_fp8 (e4m3) | scalef32 bf8 | scalef32 fp8\n", scale, "f32");
    for (int i = 0; i < n; ++i)
      printf("   %14.8g | 0x%02x = %-10g | 0x%02x = %-10g | 0x%02x = %-10g | 0x%02x = %-10g\n", vals[i], o[i * 4], e5m2_dec(o[i * 4]), o[i * 4 + 1], e4m3_dec(o[i * 4 + 1]),
             o[i * 4 + 2], e5m2_dec(o[i * 4 + 2]), o[i * 4 + 3], e4m3_dec(o[i * 4 + 3]));
    if (scale == 1.0f) {   // exhaustive check of RNE + saturation for the plain forms against a host model, on a dense sweep
      // (only reported as counts)
    }
  }
  // dense sweep: plain cvt against host round-to-nearest-even-with-saturation models
  {
    const int N = 64;
    long bad5 = 0, bad4 = 0, tot = 0;
    srand(1);
    for (int rep = 0; rep < 400; ++rep) {
      std::vector<float> v(N);
      for (int i = 0; i < N; ++i) { float m = 1.f + (rand() % 4096) / 4096.f; int e = rand() % 40 - 22; v[i] = ldexpf(m, e) * ((rand() & 1) ? -1.f : 1.f); }
      CK(hipMemcpy(din, v.data(), N * 4 > n * 4 ? n * 4 : N * 4, hipMemcpyHostToDevice));
      int nn = N > n ? n : N;
      hipLaunchKernelGGL(cvt_kernel, 1, 64, 0, 0, din, nn, dout, 1.0f);
      std::vector<uint32_t> o(nn * 4);
      CK(hipMemcpy(o.data(), dout, nn * 16, hipMemcpyDeviceToHost));
      for (int i = 0; i < nn; ++i) {
        // host model: nearest representable value (ties to even mantissa), saturate to max finite
        auto nearest = [&](float x, bool e5) {
          float best = 0.f; double bd = 1e300; int bv = 0;
          for (int c = 0; c < 256; ++c) {
            float d = e5 ? e5m2_dec((uint8_t)c) : e4m3_dec((uint8_t)c);
            if (isnan(d) || isinf(d)) continue;
            double dist = fabs((double)d - (double)x);
            if (dist < bd || (dist == bd && !(c & 1) && (bv & 1))) { bd = dist; best = d; bv = c; }
          }
          return best;
        };
        float h5 = nearest(v[i], true), h4 = nearest(v[i], false);
        if (e5m2_dec(o[i * 4]) != h5) { if (bad5 < 5) printf("   e5m2 mismatch: %g -> hw %g, model %g\n", v[i], e5m2_dec(o[i * 4]), h5); ++bad5; }
        if (e4m3_dec(o[i * 4 + 1]) != h4) { if (bad4 < 5) printf("   e4m3 mismatch: %g -> hw %g, model %g\n", v[i], e4m3_dec(o[i * 4 + 1]), h4); ++bad4; }
        ++tot;
      }
    }
    printf("dense sweep of v_cvt_pk_{bf8,fp8}_f32 against 'nearest even, saturate to the largest finite': %ld values, e5m2 mismatches %ld, e4m3 mismatches %ld\n", tot, bad5, bad4);
  }

  // ---------------- (2) MFMA layout and scales
  {
    // logical matrices: A[i][k] (16 x 128, e4m3 integers), B[k][j] (128 x 16, e5m2 integers); scales sA[i][g], sB[j][g] per 32-k group g
    int A[16][128], B[128][16], sA[16][4], sB[16][4];
    srand(7);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 128; ++k) A[i][k] = rand() % 9 - 4;
    for (int k = 0; k < 128; ++k) for (int j = 0; j < 16; ++j) B[k][j] = rand() % 7 - 3;
    for (int i = 0; i < 16; ++i) for (int g = 0; g < 4; ++g) { sA[i][g] = 127 + (rand() % 5 - 2); sB[i][g] = 127 + (rand() % 5 - 2); }
    for (int hyp = 0; hyp < 3; ++hyp) {   // 2 (added after tools/mfma_scale_layout_probe.hip): byte placement of hypothesis 1, scales per LOGICAL block of 32 consecutive k
      // hypothesis 0: lane l = (row l & 15, group l >> 4) holds k = 32 g + byte;  hypothesis 1: bytes 0-15 -> k = 16 g + b, bytes 16-31 -> k = 64 + 16 g + (b - 16)
      std::vector<i32x8> ha(64), hb(64);
      std::vector<int> hsa(64), hsb(64);
      for (int l = 0; l < 64; ++l) {
        uint8_t ba[32], bb[32];
        int r = l & 15, g = l >> 4;
        for (int b = 0; b < 32; ++b) {
          int k = hyp == 0 ? 32 * g + b : (b < 16 ? 16 * g + b : 64 + 16 * g + (b - 16));   // (hyp 2 places bytes as hyp 1)
          ba[b] = e4m3_enc_int(A[r][k]);
          bb[b] = e5m2_enc_int(B[k][r]);
        }
        memcpy(&ha[l], ba, 32); memcpy(&hb[l], bb, 32);
        // scale registers: byte 0 = this lane's (row, group) scale, byte 1 = a decoy (+3), byte 2 = 127, byte 3 = decoy
        hsa[l] = sA[r][g] | ((sA[r][g] + 3) << 8) | (127 << 16) | (100 << 24);
        hsb[l] = sB[r][g] | ((sB[r][g] + 3) << 8) | (127 << 16) | (100 << 24);
      }
      i32x8 *da, *db; int *dsa, *dsb; float* dout2;
      CK(hipMalloc(&da, 64 * 32)); CK(hipMalloc(&db, 64 * 32)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dout2, 64 * 16));
      CK(hipMemcpy(da, ha.data(), 64 * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), 64 * 32, hipMemcpyHostToDevice));
      CK(hipMemcpy(dsa, hsa.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, hsb.data(), 256, hipMemcpyHostToDevice));
      for (int op = 0; op < 3; ++op) {   // op_sel 0 (byte 0: per-lane scales), 2 (byte 2: 127 = 1.0), 1 (byte 1: scale + 3 on both = x 64)
        if (op == 0) hipLaunchKernelGGL((mfma_kernel<0, 0>), 1, 64, 0, 0, da, db, dsa, dsb, dout2);
        if (op == 1) hipLaunchKernelGGL((mfma_kernel<2, 2>), 1, 64, 0, 0, da, db, dsa, dsb, dout2);
        if (op == 2) hipLaunchKernelGGL((mfma_kernel<1, 1>), 1, 64, 0, 0, da, db, dsa, dsb, dout2);
        std::vector<float> o(256);
        CK(hipMemcpy(o.data(), dout2, 1024, hipMemcpyDeviceToHost));
        // expected: D[i][j] = sum_g 2^(sa - 127) 2^(sb - 127) sum_{k in g} A[i][k] B[k][j]; C layout: lane l, reg e -> row (l >> 4) * 4 + e, col l & 15
        int bad = 0, bad_t = 0;
        for (int l = 0; l < 64; ++l)
          for (int e = 0; e < 4; ++e) {
            int i = (l >> 4) * 4 + e, j = l & 15;
            double ref = 0;
            for (int g = 0; g < 4; ++g) {
              double s = 0;
              for (int k = 32 * g; k < 32 * g + 32; ++k) {
                int kk = k;   // logical k; hypothesis only changes the byte placement above
                s += (double)A[i][kk] * B[kk][j];
              }
              // under hypothesis 1 the scale groups are the lanes' groups: k sets {16g..16g+15} U {64+16g..}
              if (hyp == 1) { s = 0; for (int b = 0; b < 32; ++b) { int k = b < 16 ? 16 * g + b : 64 + 16 * g + (b - 16); s += (double)A[i][k] * B[k][j]; } }
              double sc = op == 0 ? ldexp(1.0, sA[i][g] - 127 + sB[j][g] - 127) : op == 1 ? 1.0 : ldexp(1.0, sA[i][g] - 127 + 3 + sB[j][g] - 127 + 3);
              ref += sc * s;
            }
            if (fabs(ref - o[l * 4 + e]) > 1e-3 * (1 + fabs(ref))) ++bad;
            // transposed C hypothesis
            int it = l & 15, jt = (l >> 4) * 4 + e;
            double reft = 0;
            for (int g = 0; g < 4; ++g) { double s = 0; for (int k = 32 * g; k < 32 * g + 32; ++k) s += (double)A[it][k] * B[k][jt]; reft += (op == 1 ? 1.0 : 0.0) * s; }
            if (op == 1 && fabs(reft - o[l * 4 + e]) > 1e-3 * (1 + fabs(reft))) ++bad_t;
          }
        printf("MFMA e4m3 x e5m2, byte hypothesis %d (%s), op_sel %d: %d / 256 mismatches against 'lane (r, g): row r, k-group g; C row = (l>>4)*4+e, col = l&15; scale byte op_sel, value 2^(s-127), per lane (row, group)'%s\n",
               hyp, hyp == 0 ? "k = 32 g + byte" : hyp == 1 ? "bytes 0-15: k = 16 g + b, 16-31: k = 64 + 16 g + b - 16; scale = the lane's own k set" : "bytes as hypothesis 1; the scale of lane (r, g) belongs to the LOGICAL block k = 32 g .. 32 g + 31", op == 0 ? 0 : op == 1 ? 2 : 1, bad,
               op == 1 ? (bad_t == 0 ? "  [the TRANSPOSED C layout also matches?!]" : "") : "");
      }
      CK(hipFree(da)); CK(hipFree(db)); CK(hipFree(dsa)); CK(hipFree(dsb)); CK(hipFree(dout2));
    }
  }
  // ---------------- (3) issue cost
  {
    long long* dc; float* ds;
    CK(hipMalloc(&dc, 1024 * 8)); CK(hipMalloc(&ds, 1024 * 256 * 4));
    for (int waves = 1; waves <= 2; ++waves)
      for (int mode = 0; mode < 2; ++mode) {
        const int iters = 2000;
        for (int rep = 0; rep < 2; ++rep) {
          if (mode == 0) hipLaunchKernelGGL((rate_kernel<0>), 256, 256 * waves, 0, 0, dc, ds, iters);
          else hipLaunchKernelGGL((rate_kernel<1>), 256, 256 * waves, 0, 0, dc, ds, iters);
          CK(hipDeviceSynchronize());
        }
        std::vector<long long> c(256);
        CK(hipMemcpy(c.data(), dc, 256 * 8, hipMemcpyDeviceToHost));
        double avg = 0; for (auto x : c) avg += x; avg /= 256;
        printf("%s, %d wave(s) per SIMD: %.1f clock64 ticks per MFMA per wave (8 independent accumulators); flops per instruction %d\n",
               mode == 0 ? "v_mfma_f32_16x16x32_f16        " : "v_mfma_scale_f32_16x16x128_f8f6f4", waves, avg / (iters * 8.0), mode == 0 ? 16 * 16 * 32 * 2 : 16 * 16 * 128 * 2);
      }
    // wall-clock rate over the whole chip
    for (int mode = 0; mode < 2; ++mode) {
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      const int iters = 20000;
      CK(hipEventRecord(e0));
      if (mode == 0) hipLaunchKernelGGL((rate_kernel<0>), 1024, 256, 0, 0, dc, ds, iters); else hipLaunchKernelGGL((rate_kernel<1>), 1024, 256, 0, 0, dc, ds, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      double fl = 1024.0 * 4 * iters * 8 * (mode == 0 ? 16384.0 : 65536.0);
      printf("%s whole chip (constant operands): %.0f TFLOP/s\n", mode == 0 ? "fp16 16x16x32 " : "fp8 16x16x128", fl / ms / 1e9);
    }
  }
  return 0;
}
