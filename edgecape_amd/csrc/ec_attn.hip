// Fused multi-head attention for gfx950: O = softmax(Q K^T / sqrt(hd) + bias, key mask) V.
//
// Reference ops: DINOv2 Attention, nn.MultiheadAttention in the encoder / decoder /
// skeleton two-way layers (encoder_decoder.py:444,558,561,573) and BiasedMultiheadAttention
// (bias_attn.py:183-216).  Sequence lengths are 100..829, head dims 32 / 64, so one workgroup owns
// 128 queries of one (batch, head) and streams K/V through LDS in 64-key tiles with an online softmax.
//
// Everything is computed TRANSPOSED so that softmax statistics are lane-local (64-wide waves):
//   S^T[key, query] = K · Q^T        A = K tile (LDS),  B = Q (registers, pre-scaled)
//   O^T[d,   query] = V^T · P^T      A = V tile (LDS),  B = P = exp(S^T - m) straight from the accumulators
// A 32x32 MFMA accumulator holds, in lane (query = lane&31, half = lane>>5), 16 keys of that query;
// the other 16 live in lane^32, so a row max / sum is 16 in-register ops + one cross-half exchange, and
// the per-query rescale of O^T needs no shuffles at all.  The accumulator register->key map of S^T is
// exactly a valid k-slot assignment for the B operand of the second MFMA, so P never moves.
//
//   fp32 path : v_mfma_f32_32x32x2_f32 (exact fp32 products) — parity mode.
#include "ec_common.h"

namespace ec {
namespace {

constexpr int KT = 64;  // keys per LDS tile

// acc register r (0..15) of half `hi` holds row (r&3) + 8*(r>>2) + 4*hi of the 32x32 tile
__device__ inline int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

template <int HD>
__global__ __launch_bounds__(256) void attn_f32_kernel(AttnP p) {
  constexpr int KS = HD + 1;  // K tile row stride (floats): odd -> conflict-free column reads
  constexpr int NM = HD / 2;  // MFMAs per 32x32 S^T tile
  constexpr int DT = HD / 32; // 32-wide d tiles of O^T
  __shared__ float Ks[KT * KS];
  __shared__ float Vs[KT * HD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hi = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const float* Q = (const float*)p.Q + (long)b * p.sQ + h * HD;
  const float* K = (const float*)p.K + (long)b * p.sK + h * HD;
  const float* V = (const float*)p.V + (long)b * p.sV + h * HD;
  const float scale = rsqrtf((float)HD);

  // Q fragment: lane (j, hi) holds Q[q0+j][hi*NM + m], m = 0..NM-1 (contiguous), pre-scaled.
  float qf[NM];
  {
    int qr = q0 + j;
    qr = qr < p.Lq ? qr : p.Lq - 1;
    const float* src = Q + (long)qr * p.ldq + hi * NM;
#pragma unroll
    for (int m = 0; m < NM; m += 4) {
      const f32x4 v = *(const f32x4*)(src + m);
      qf[m] = v[0] * scale; qf[m + 1] = v[1] * scale; qf[m + 2] = v[2] * scale; qf[m + 3] = v[3] * scale;
    }
  }
  f32x16 ot[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[d][r] = 0.f;
  float mrun = -1e30f, lrun = 0.f;

  const uint8_t* km = p.kmask ? p.kmask + (long)(p.mask_mod > 0 ? b % p.mask_mod : b) * p.mask_len : nullptr;
  const float* bias = p.bias ? p.bias + ((long)(b * p.H + h) * p.Lq) * p.Lk : nullptr;
  const int qrow = (q0 + j) < p.Lq ? (q0 + j) : p.Lq - 1;

  for (int k0 = 0; k0 < p.Lk; k0 += KT) {
    __syncthreads();  // previous tile fully consumed
    // stage K,V tile: KT x HD floats each; 256 threads x float4
    for (int idx = tid; idx < KT * HD / 4; idx += 256) {
      const int r = idx / (HD / 4), c = (idx % (HD / 4)) * 4;
      int kr = k0 + r;
      kr = kr < p.Lk ? kr : p.Lk - 1;
      const f32x4 kv = *(const f32x4*)(K + (long)kr * p.ldk + c);
      const f32x4 vv = *(const f32x4*)(V + (long)kr * p.ldv + c);
      float* kd = Ks + r * KS + c;
      kd[0] = kv[0]; kd[1] = kv[1]; kd[2] = kv[2]; kd[3] = kv[3];
      *(f32x4*)(Vs + r * HD + c) = vv;
    }
    __syncthreads();

    // S^T for two 32-key sub-tiles
    f32x16 s[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
      const float* kr = Ks + (t * 32 + j) * KS + hi * NM;
#pragma unroll
      for (int m = 0; m < NM; ++m) s[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[m], qf[m], s[t], 0, 0, 0);
    }
    // bias, masks, tile max
    float tmax = -INFINITY;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kg = k0 + t * 32 + acc_row(r, hi);
        float v = s[t][r];
        if (bias && kg < p.Lk) v += bias[(long)qrow * p.Lk + kg];
        bool masked = kg >= p.Lk;
        if (km && !masked && kg >= p.mask_start) masked = km[kg - p.mask_start] != 0;
        v = masked ? -INFINITY : v;
        s[t][r] = v;
        tmax = fmaxf(tmax, v);
      }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float mnew = fmaxf(mrun, tmax);
    const float alpha = expf(mrun - mnew);
    mrun = mnew;
    float psum = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = expf(s[t][r] - mnew);
        s[t][r] = e;
        psum += e;
      }
    lrun = lrun * alpha + psum;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[d][r] *= alpha;
    // O^T += V^T P^T : MFMA #r of sub-tile t: k-slot `hi` <-> key t*32 + acc_row(r, hi)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float* vr = Vs + (t * 32 + acc_row(r, hi)) * HD + j;
#pragma unroll
        for (int d = 0; d < DT; ++d) ot[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[d * 32], s[t][r], ot[d], 0, 0, 0);
      }
  }
  lrun += __shfl_xor(lrun, 32, 64);
  const float inv = 1.f / lrun;
  if (q0 + j < p.Lq) {
    float* O = (float*)p.O + (long)b * p.sO + (long)(q0 + j) * p.ldo + h * HD;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v;
        v[0] = ot[d][4 * g] * inv; v[1] = ot[d][4 * g + 1] * inv; v[2] = ot[d][4 * g + 2] * inv; v[3] = ot[d][4 * g + 3] * inv;
        *(f32x4*)(O + d * 32 + 8 * g + 4 * hi) = v;
      }
  }
}

}  // namespace

int attention(const AttnP& p, hipStream_t st) {
  EC_REQUIRE(p.B > 0 && p.H > 0 && p.Lq > 0 && p.Lk > 0, -1, "attention: empty problem");
  EC_REQUIRE(p.hd == 32 || p.hd == 64, -1, "attention: head dim must be 32 or 64");
  EC_REQUIRE(!p.bf16, -1, "attention: bf16 path not built");
  EC_REQUIRE(p.ldq % 4 == 0 && p.ldk % 4 == 0 && p.ldv % 4 == 0 && p.ldo % 4 == 0, -1, "attention: strides must be multiples of 4");
  dim3 grid((p.Lq + 127) / 128, p.H, p.B);
  if (p.hd == 64) hipLaunchKernelGGL(attn_f32_kernel<64>, grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL(attn_f32_kernel<32>, grid, dim3(256), 0, st, p);
  EC_LAUNCH_CHECK();
  return 0;
}

}  // namespace ec
