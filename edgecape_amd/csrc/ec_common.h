// Internal helpers shared by the gfx950 kernels of libedgecape_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

namespace ec {

typedef unsigned short bf16_t;  // storage type for bf16 bit patterns
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;

void set_error(const std::string& s);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define EC_HIP(expr)                                                              \
  do {                                                                            \
    hipError_t _e = (expr);                                                       \
    if (_e != hipSuccess) return ::ec::hip_fail(_e, #expr, __FILE__, __LINE__);   \
  } while (0)

#define EC_LAUNCH_CHECK()                                                                         \
  do {                                                                                            \
    hipError_t _e = hipGetLastError();                                                            \
    if (_e != hipSuccess) return ::ec::hip_fail(_e, "kernel launch", __FILE__, __LINE__);         \
  } while (0)

#define EC_REQUIRE(cond, code, msg)                                  \
  do {                                                               \
    if (!(cond)) {                                                   \
      ::ec::set_error(std::string(msg) + " [" #cond "]");            \
      return (code);                                                 \
    }                                                                \
  } while (0)

__host__ __device__ inline bf16_t f2bf(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_bit_cast(bf16_t, (__bf16)f);   // v_cvt_pk_bf16_f32 (round to nearest even)
#endif
  union { float f; uint32_t u; } v;
  v.f = f;
  uint32_t u = v.u;
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);                                         // round to nearest even
  return (bf16_t)(u >> 16);
}
__host__ __device__ inline float bf2f(bf16_t h) {
  union { float f; uint32_t u; } v;
  v.u = ((uint32_t)h) << 16;
  return v.f;
}

// ---- 16-bit operand formats of the backbone's throughput modes -------------------------------------------------------------
// F16 = false: bfloat16 (8 significand bits); F16 = true: IEEE binary16 (11 significand bits, same MFMA rate on gfx950:
// v_mfma_f32_*_f16 / v_cvt_pk_f16_f32).  Storage is a raw 16-bit pattern (bf16_t) either way; only the conversions and the
// MFMA opcode differ, so every 16-bit kernel takes the format as a template parameter.
template <bool F16> __device__ __forceinline__ f32x4 mfma16x16x32_h(bf16x8 a, bf16x8 b, f32x4 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <bool F16> __device__ __forceinline__ f32x16 mfma32x32x16_h(bf16x8 a, bf16x8 b, f32x16 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// fp16 saturation at +-65504 that keeps NaN: v_med3_f32 alone returns min3 = -65504 for a NaN input (its NaN rule: the minimum of the
// operands, and v_min ignores quiet NaNs), which would turn a NaN activation into a finite value and hide it from the caller
// (round 3 did).  v * 0 is NaN exactly when v is NaN or +-inf and (+-)0 otherwise, so one more full-rate v_fma_f32 puts the NaN back:
// finite v -> clamp(v), NaN -> NaN, an fp32 +-inf (an accumulator overflow: nothing finite to saturate to) -> NaN as well.
__device__ __forceinline__ float sat_h16(float v) { return fmaf(v, 0.f, __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f)); }
// 4 floats -> 4 packed 16-bit values (round to nearest even): 2 x v_cvt_pk_{bf16,f16}_f32
// (fp16: saturated at +-65504 first - sat_h16 - so an activation outlier of a real checkpoint becomes the largest finite half
// instead of inf -> NaN in the next softmax / LayerNorm; NaN inputs stay NaN)
template <bool F16> __device__ __forceinline__ u32x2_t pack4_h(f32x4 v) {
  if constexpr (F16) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = sat_h16(v[e]);
    return __builtin_bit_cast(u32x2_t, __builtin_convertvector(v, f16x4));
  }
  else {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16v4_;
    return __builtin_bit_cast(u32x2_t, __builtin_convertvector(v, bf16v4_));
  }
}
// The same for kernels that have switched the wave's MODE.FP16_OVFL bit on (fp16_ovfl_mode() at kernel entry): with that bit the
// hardware clamps an overflowing fp16 conversion result to +-65504 itself, keeps true infinities and NaN (probed on gfx950:
// tools/fp16_ovfl_probe.hip, profiles/r04_fp16_ovfl_probe.txt) - the saturation costs no instruction at all.  Used by the 8-phase
// GEMM, whose epilogues are VALU-bound (v_med3 + v_fma per value measured +1.5 % on the QKV launch, profiles/r04_sat_ab.txt).
// The bit is switched on for the conversion passes ONLY (the GEMM: around a tile's epilogue): with it set for the whole kernel a NaN
// activation no longer came out as NaN (tests/test_gpu_ops.py::test_linear_h16_fp16_nan_in_nan_out) - the mode is not confined to
// conversions.
template <int ON> __device__ __forceinline__ void fp16_ovfl_mode() { __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), ON); }   // hwreg(HW_REG_MODE, 23, 1) = ON
template <bool F16> __device__ __forceinline__ u32x2_t pack4_h_ovfl(f32x4 v) {
  if constexpr (F16) return __builtin_bit_cast(u32x2_t, __builtin_convertvector(v, f16x4));
  else {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16v4_;
    return __builtin_bit_cast(u32x2_t, __builtin_convertvector(v, bf16v4_));
  }
}
// bf16 SPLIT of four fp32 values (the K-concatenated form of the bf16x3 backbone, ec_model.hip run_backbone): hi = bf16(v) (RNE),
// lo = bf16(v - hi); v = hi + lo to ~2^-17 |v|.  An infinite v gives lo = NaN: the products come out NaN, as with any infinite operand.
__device__ __forceinline__ void split4_bf16(f32x4 v, u32x2_t& hi, u32x2_t& lo) {
  hi = pack4_h_ovfl<false>(v);
  f32x4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) r[e] = v[e] - __uint_as_float((e & 1) ? (hi[e >> 1] & 0xffff0000u) : (hi[e >> 1] << 16));
  lo = pack4_h_ovfl<false>(r);
}
// The same split in the operand format of the GEMM that reads the planes.  F16 (IEEE fp16 planes): hi + lo carries 22 significand bits
// instead of 16 - 4.5 x less error in the backbone's features for the same three MFMAs (profiles/r05_x3_fp16_planes_ab.txt) - and the
// lo part of a small value is an fp16 SUBNORMAL, which the gfx950 matrix pipe keeps (tools/mfma_f16_denorm_probe.hip).  Magnitudes past
// 65504 saturate (NaN stays NaN: sat_h16), first in hi, then in the remainder.  OVFL: the caller has MODE.FP16_OVFL switched on (the
// 8-phase GEMM's epilogue), the conversions saturate by themselves.
template <bool F16, bool OVFL = false> __device__ __forceinline__ void split4_h(f32x4 v, u32x2_t& hi, u32x2_t& lo) {
  if constexpr (!F16) {
    split4_bf16(v, hi, lo);
  } else {
    f32x4 s = v;
    if constexpr (!OVFL) {
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] = sat_h16(v[e]);
    }
    const f16x4 h = __builtin_convertvector(s, f16x4);
    hi = __builtin_bit_cast(u32x2_t, h);
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = OVFL ? v[e] - (float)h[e] : sat_h16(v[e] - (float)h[e]);
    lo = __builtin_bit_cast(u32x2_t, __builtin_convertvector(r, f16x4));
  }
}
// ---- fp16x2 operand format (EC_F16X2 backbone, round 6): TWO MFMA units per product instead of the three of bf16x3 / fp16x3 ------------
//     a W  ~  a_hi W_hi   [fp16 x fp16, v_mfma_f32_16x16x32_f16]
//           + q5(a_lo) q4(W_hi) + q5(a_hi) q4(W_lo)   [FP8, v_mfma_scale_f32_16x16x128_f8f6f4: four times the flops per instruction at twice
//                                                     the issue cost - one pass of depth 2 K for BOTH correction terms = one unit]
// The correction terms are ~2^-12 of the product, so a few significand bits of THEIR operands suffice (oracle/x2_at_scale.py: the CPU
// emulation of this scheme at the conformance sets' scale, before any kernel was written).  An activation row of K values is stored as
//     [ hi: K x fp16 | lo8: K x e5m2 of (a - hi) * 2^11 | hi8: K x e5m2 of a ]          4 K bytes, as the two-plane bf16x3 row it replaces
// - e5m2 has fp16's exponent range and (a - hi) * 2^11 has the exponent range of a itself, so BOTH FP8 planes take FIXED power-of-two
// scales (E8M0 116 = 2^-11 and 127 = 1 in the MFMA's scale operand): no data-dependent scale, no reduction in any producer (LayerNorm,
// attention, the fc1 epilogue each convert their own four values), and nothing can overflow while a fits fp16 (e5m2 conversions
// saturate at +-57344).  A weight row is [ W_hi: K x fp16 | W_hi8: K x e4m3 of W * 2^s1 | W_lo8: K x e4m3 of (W - W_hi) * 2^s2 ] with one
// static power-of-two scale per tensor and plane (ec_finalize knows the weights).  The GEMM's load stream walks both rows straight
// through (no K wrap): K / 64 K-tiles of fp16 MFMAs, then K / 64 K-tiles of FP8 MFMAs - 128 bytes of a row are one 16x16x128 operand.
// Four fp32 values -> four e5m2 bytes of v * 2^SHIFT (round to nearest even).  v_cvt_pk_bf8_f32 does NOT saturate (tools/fp8_mfma_probe.hip:
// >= 61440 -> inf), so the magnitude is clamped first (one v_med3: a NaN comes out finite here - the fp16 plane of the same value carries
// the NaN into the product); the power of two rides in the scaled conversion (v_cvt_scalef32_pk_bf8_f32 divides by its scale operand).
template <int SHIFT> __device__ __forceinline__ unsigned pack4_e5m2(f32x4 v) {
  constexpr float lim = SHIFT == 0 ? 57344.f : 57344.f / (float)(1 << SHIFT);
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], -lim, lim);
  typedef __attribute__((ext_vector_type(2))) short s16x2_;
  if constexpr (SHIFT == 0) {
    int r = __builtin_amdgcn_cvt_pk_bf8_f32(v[0], v[1], 0, false);
    return (unsigned)__builtin_amdgcn_cvt_pk_bf8_f32(v[2], v[3], r, true);
  } else {
    constexpr float inv = 1.f / (float)(1 << SHIFT);
    s16x2_ r = {0, 0};
    r = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(r, v[0], v[1], inv, false);
    r = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(r, v[2], v[3], inv, true);
    return __builtin_bit_cast(unsigned, r);
  }
}
// OVFL: the caller has MODE.FP16_OVFL on (8-phase GEMM epilogue): the fp16 conversion saturates by itself
template <bool OVFL = false> __device__ __forceinline__ void split4_x2(f32x4 v, u32x2_t& hi, unsigned& lo8, unsigned& hi8) {
  f32x4 s = v;
  if constexpr (!OVFL) {
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] = sat_h16(v[e]);
  }
  const f16x4 h = __builtin_convertvector(s, f16x4);
  hi = __builtin_bit_cast(u32x2_t, h);
  f32x4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) r[e] = v[e] - (float)h[e];
  lo8 = pack4_e5m2<11>(r);      // |a - hi| <= 16 for any a inside the fp16 range: the clamp (28) only bites on saturated values
  hi8 = pack4_e5m2<0>(v);
}
constexpr int X2_SCALE_LO8 = 127 - 11, X2_SCALE_HI8 = 127;   // E8M0 scale bytes of the two activation planes
template <bool F16> __device__ __forceinline__ bf16_t f2h(float f) {
  if constexpr (F16) return __builtin_bit_cast(bf16_t, (_Float16)sat_h16(f));
  else return __builtin_bit_cast(bf16_t, (__bf16)f);
}
template <bool F16> __device__ __forceinline__ float h2f(bf16_t h) {
  if constexpr (F16) return (float)__builtin_bit_cast(_Float16, h);
  else return __uint_as_float(((uint32_t)h) << 16);
}
// host: float -> IEEE binary16 bit pattern, round to nearest even (weights of the fp16 mode are converted once at ec_finalize)
inline bf16_t f2half_host(float f) {
  union { float f; uint32_t u; } v;
  v.f = f;
  const uint32_t sign = (v.u >> 16) & 0x8000u;
  const uint32_t a = v.u & 0x7fffffffu;
  if (a >= 0x7f800000u) return (bf16_t)(sign | 0x7c00u | (a > 0x7f800000u ? 0x200u : 0u));   // inf / NaN
  if (a >= 0x477ff000u) return (bf16_t)(sign | 0x7c00u);                                        // rounds to >= 65520: inf
  if (a < 0x33000001u) return (bf16_t)sign;                                                      // < 2^-25 (or == 2^-25: ties to even 0)
  int e = (int)(a >> 23) - 127;
  uint32_t m = (a & 0x7fffffu) | 0x800000u;
  int shift = e < -14 ? (13 + (-14 - e)) : 13;        // subnormal halves lose extra bits
  uint32_t h = m >> shift;
  const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
  if (rem > half || (rem == half && (h & 1u))) ++h;
  if (e < -14) return (bf16_t)(sign | h);             // h may carry into the smallest normal: still the right pattern
  return (bf16_t)(sign | (uint32_t)(((e + 15) << 10) + (h - 0x400u)));   // mantissa carry propagates into the exponent
}

inline float half2f_host(bf16_t h) {
  const uint32_t sign = ((uint32_t)h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
  union { float f; uint32_t u; } v;
  if (e == 0) {
    v.f = (float)m * 5.9604644775390625e-08f;   // m * 2^-24 (zero and subnormals), exact in fp32
    v.u |= sign;
  } else if (e == 31) {
    v.u = sign | 0x7f800000u | (m << 13);
  } else {
    v.u = sign | ((e + 112u) << 23) | (m << 13);
  }
  return v.f;
}

template <typename T> struct Store;
template <> struct Store<float> {
  __device__ static inline void put(float* p, float v) { *p = v; }
  __device__ static inline float get(const float* p) { return *p; }
};
template <> struct Store<bf16_t> {
  __device__ static inline void put(bf16_t* p, float v) { *p = f2bf(v); }
  __device__ static inline float get(const bf16_t* p) { return bf2f(*p); }
};

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ inline float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// erf-GELU (nn.GELU default) to fp32 accuracy without the erff call: x * Phi(x), Phi(x) = 1 / (1 + 2^(x * P(x^2))), P a degree-6
// weighted-minimax polynomial in x^2 (max |x Phi - x Phi_exact| = 6.7e-8 on |x| <= 10; its leading coefficient is negative, so
// x * P -> -/+ inf beyond and the result goes to its exact limits x / -0 without a clamp).  10 plain VALU + exp2 + rcp per value
// against ~40 for erff: the row chains spend up to 2 500 cycles per GELU stage in it (round 3).
__device__ __forceinline__ float gelu_fast32(float x) {
  const float s = x * x;
  float q = fmaf(-5.2099139186e-09f, s, 3.8504522544e-07f);
  q = fmaf(q, s, -1.1452433607e-05f);
  q = fmaf(q, s, 1.5938943863e-04f);
  q = fmaf(q, s, 9.5592844954e-05f);
  q = fmaf(q, s, -1.0483858114e-01f);
  q = fmaf(q, s, -2.3022072036e+00f);
  return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * q));
}

// GELU for 16-bit outputs (erf form: nn.GELU default, dinov2 Mlp) with ONE transcendental (round 4):
//     gelu(x) = x Phi(x) = max(x, 0) - |x| Phi(-|x|),      Phi(-a) = 2^L(a),  L(a) = log2 Phi(-a)  smooth and concave on a >= 0
// (for x > 0 by Phi(x) = 1 - Phi(-x)).  L is replaced by a polynomial in a = |x|, a weighted minimax fit (weight a Phi(-a) ln 2, the
// sensitivity of the result to L; tools/gelu_fit.py: Lawson iteration on [0, 6]) whose leading coefficient is negative, so the
// polynomial runs to -inf beyond the fitted range, 2^ -> 0 and the result goes to its exact limits x / -0 without a clamp.
//   fp16 outputs: degree 5, max |err| 6.4e-7 in fp32 arithmetic on |x| <= 12 (the round-2/3 form 1 / (1 + 2^(x P(x^2))) had 3.0e-6);
//   bf16 outputs: degree 3, max |err| 5.5e-5 (far below the bf16 rounding of the result).
// Cost per element: 5 (3) v_fma + v_exp + v_max + v_fma = 7 (5) full-rate and ONE quarter-rate instruction; the old form was 8 (6)
// full-rate and TWO quarter-rate ones (v_exp + v_rcp = 16 of its ~34 cycles; measured and rejected in round 3: a degree-2 polynomial,
// packed-fp16 polynomials - both kept the two transcendentals).  |x| and -|x| are source modifiers, NaN goes through the last fma.
// The limits hold for FINITE x; an infinite accumulator (an fp32 overflow, or an infinite activation) comes out as NaN for either
// sign: q = -inf, 2^q = 0 and -|x| * 0 = NaN (the exact GELU would give +inf / -0).  Pinned by test_linear_h16_fp16_nan_in_nan_out.
template <bool F16>
__device__ __forceinline__ float gelu_fast8(float x) {
  const float a = fabsf(x);
  float q;
  if constexpr (F16) {
    q = fmaf(-4.732939302e-04f, a, 7.084460654e-03f);
    q = fmaf(q, a, -5.182716738e-02f);
    q = fmaf(q, a, -4.599926465e-01f);
    q = fmaf(q, a, -1.150787770e+00f);
    q = fmaf(q, a, -1.000037632e+00f);
  } else {
    q = fmaf(-2.487393087e-02f, a, -4.988535682e-01f);
    q = fmaf(q, a, -1.129219622e+00f);
    q = fmaf(q, a, -1.003536762e+00f);
  }
  return fmaf(-a, __builtin_amdgcn_exp2f(q), fmaxf(x, 0.f));
}

// ---------------------------------------------------------------------------------------------
// GEMM (ec_gemm.hip):  C = epilogue(A[M,K] @ B[N,K]^T), optional batch via blockIdx.z.
// epilogue:  v = acc + bias[n] + table[m % period][n];  v = act(v);
//            act==TANHGATE: v = (tanh(v) + 1) * aux[m][n];
//            v *= gamma[n];  v += resid[m][n];  store (fp32 or bf16)
// ---------------------------------------------------------------------------------------------
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_TANHGATE = 3 };

struct GemmP {
  const void* A = nullptr;
  const void* B = nullptr;
  void* C = nullptr;
  const float* bias = nullptr;
  const float* gamma = nullptr;
  const float* resid = nullptr;
  const float* table = nullptr;
  const float* aux = nullptr;
  long lda = 0, ldb = 0, ldc = 0, ldr = 0, ldt = 0, ldaux = 0;
  long sA = 0, sB = 0, sC = 0, sBias = 0, sR = 0, sAux = 0;  // batch strides in elements
  int M = 0, N = 0, K = 0, batch = 1;
  int period = 1;
  int act = ACT_NONE;
  int c_bf16 = 0;   // store C as bf16
  int c_x3 = 0;     // store C as the bf16 split [hi | lo] of the fp32 result (split4_bf16): 16-bit rows of stride ldc, the two
                    // planes N elements apart (ldc >= 2 N) - the A operand of a following K-concatenated bf16x3 GEMM; GELU = gelu_fast8<true>
                    // (6.4e-7) in EVERY epilogue that writes this format (8-phase staged / generic, 2-barrier fallback): an image's features
                    // do not change form with the kernel its batch size selects
  int kwrap = 0;    // 16-bit operands only: K-CONCATENATED bf16x3 product over TWO-plane operands.  kwrap = K-steps (64 elements) per plane;
                    // K = 3 * 64 * kwrap logical steps, step kt reads A at plane step (kt < 2 kwrap ? kt : kt - 2 kwrap) - planes
                    // [hi | lo | hi] of A = [hi | lo] - and B at (kt < kwrap ? kt : kt - kwrap) - planes [hi | hi | lo] of B = [W_hi | W_lo]:
                    // a_hi W_hi + a_lo W_hi + a_hi W_lo in one accumulator, no plane stored or fetched from HBM twice.  lda, ldb >= 128 kwrap.
  int x2 = 0;       // fp16x2 operands (split4_x2 above; 8-phase kernel only): rows of K = 2 K_layer 16-bit units = [fp16 plane | FP8 plane | FP8
                    // plane]; x2 = K_layer / 64 = number of fp16 K-tiles, the other K_layer / 64 K-tiles are FP8 (first half: A lo8 x B hi8,
                    // second half: A hi8 x B lo8).  x2_sa: E8M0 scale bytes of B's (the weight's) two FP8 planes, byte 0 | byte 1.
  int x2_sa = 0;
  int c_x2 = 0;     // store C in the fp16x2 row format [hi | lo8 | hi8] (ldc in 16-bit units >= 2 N): the A operand of a following x2 GEMM
  int ab_bf16 = 0;  // A and B are 16-bit (else fp32)
  int h_f16 = 0;    // the 16-bit format (operands and, with c_bf16, the output) is IEEE fp16 instead of bf16
  int split = 0;    // 1 = bf16x3: A fp32, B pre-split into [32 hi | 32 lo] bf16 per 32-k block (split_pack_weights);
                    // 2 = fp16x1: A fp32 rounded to fp16 in registers, B [32 fp16 | unused] per 32-k block (split_pack_weights_h1)
  int tag = 0;      // kernel-symbol tag (profiling only): 1 qkv, 2 proj, 3 fc1, 4 fc2
  int* sched = nullptr;   // 8-phase kernel only: device int[9], all zero between launches - DYNAMIC tile schedule (ec_gemm8.hip); nullptr: static
};
int gemm_nt(const GemmP& p, hipStream_t st);
// host: W [N,K] fp32 -> bf16x3 packing of the same byte size: per row, per 32-k block, 32 hi bf16 then 32 lo bf16
void split_pack_weights(const float* W, long n_rows, long K, float* out);
void split_pack_weights_h1(const float* W, long n_rows, long K, float* out);   // [32 fp16 | 32 x 0] per 32-k block
// 256x256x64 8-phase bf16 kernel (ec_gemm8.hip): 1 = handled, 0 = shape not eligible (use gemm_nt's own kernels), < 0 error
int gemm8_bf16(const GemmP& p, hipStream_t st);

// Small batched fp32 GEMM on the vector ALU (ec_gemm.hip): C[b] = alpha * A[b] @ op(B[b]) + beta * C[b]
// transB = 1: B is [N,K] (NT); transB = 0: B is [K,N] (NN).  Arbitrary sizes/strides.
// Optional GCN epilogue (encoder_decoder.py:517-519): C = relu(C + rowscale[b][m] * self[b][m][n]).
struct BgemmP {
  const float* A = nullptr;
  const float* B = nullptr;
  float* C = nullptr;
  long lda = 0, ldb = 0, ldc = 0;
  long sA = 0, sB = 0, sC = 0;
  int M = 0, N = 0, K = 0, batch = 1;
  int transB = 0;
  float alpha = 1.f, beta = 0.f;
  int modA = 0, modB = 0;          // if > 0: batch index for A / B is (b % mod)
  const float* self = nullptr;     // GCN self term [batch, M, ld_self]
  long ld_self = 0, s_self = 0;
  const float* rowscale = nullptr; // [mod_rs or batch, M]
  int mod_rs = 0;
  int relu = 0;
};
int bgemm_small(const BgemmP& p, hipStream_t st);

// Attention (ec_attn.hip): O = softmax(Q K^T * hd^-0.5 + bias, mask) V, per (batch, head).
struct AttnP {
  const void* Q = nullptr;
  const void* K = nullptr;
  const void* V = nullptr;
  void* O = nullptr;
  long ldq = 0, ldk = 0, ldv = 0, ldo = 0;      // row strides (elements)
  long sQ = 0, sK = 0, sV = 0, sO = 0;          // batch strides (elements)
  const uint8_t* kmask = nullptr;               // [mask_mod or B, mask_len] 1 = masked; applies to keys >= mask_start
  int mask_start = 0, mask_len = 0, mask_mod = 0;
  const float* bias = nullptr;                  // [B, H, Lq, Lk]
  int B = 0, H = 0, Lq = 0, Lk = 0, hd = 0;
  int bf16 = 0;                                 // Q/K/V/O are 16-bit
  int f16 = 0;                                  // ... in IEEE fp16 instead of bf16
  int split = 0;                                // fp32 data, bf16x3 MFMAs (head throughput mode)
  int one = 0;                                  // split mode only: ONE fp16 MFMA per product instead of three bf16 ones (mixed head, see ec_attn.hip)
  int o_x3 = 0;                                 // split mode only: O is written as the bf16 split [hi | lo] (split4_bf16): 16-bit rows of
                                                // stride ldo (sO in the same units), planes H * hd elements apart
  int kv16 = 0;                                 // split mode only: K and V are IEEE fp16 (ldk / ldv / sK / sV in fp16 elements), Q and O fp32
};
int attention(const AttnP& p, hipStream_t st);

}  // namespace ec
