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
  int ab_bf16 = 0;  // A and B are bf16 (else fp32)
  int split = 0;    // bf16x3: A fp32, B pre-split into [32 hi | 32 lo] bf16 per 32-k block (split_pack_weights)
  int tag = 0;      // kernel-symbol tag (profiling only): 1 qkv, 2 proj, 3 fc1, 4 fc2
  int dbg = 0;      // timing experiments (EC_G8_DBG): bit 0 = skip the C stores
};
int gemm_nt(const GemmP& p, hipStream_t st);
// host: W [N,K] fp32 -> bf16x3 packing of the same byte size: per row, per 32-k block, 32 hi bf16 then 32 lo bf16
void split_pack_weights(const float* W, long n_rows, long K, float* out);
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
  int bf16 = 0;                                 // Q/K/V/O are bf16
  int split = 0;                                // fp32 data, bf16x3 MFMAs (head throughput mode)
};
int attention(const AttnP& p, hipStream_t st);

}  // namespace ec
