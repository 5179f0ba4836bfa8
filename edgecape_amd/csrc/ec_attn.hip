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
#include <stdlib.h>

#include <type_traits>

#include "ec_common.h"

namespace ec {
namespace {

constexpr int KT = 64;  // keys per LDS tile

// acc register r (0..15) of half `hi` holds row (r&3) + 8*(r>>2) + 4*hi of the 32x32 tile
__device__ inline int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

template <int HD, int NW>
__global__ __launch_bounds__(NW * 64) void attn_f32_kernel(AttnP p) {
  constexpr int KS = HD + 1;  // K tile row stride (floats): odd -> conflict-free column reads
  constexpr int NM = HD / 2;  // MFMAs per 32x32 S^T tile
  constexpr int DT = HD / 32; // 32-wide d tiles of O^T
  __shared__ float Ks[KT * KS];
  __shared__ float Vs[KT * HD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hi = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * (NW * 32) + wave * 32;
  const float* Q = (const float*)p.Q + (long)b * p.sQ + h * HD;
  const float* K = (const float*)p.K + (long)b * p.sK + h * HD;
  const float* V = (const float*)p.V + (long)b * p.sV + h * HD;
  // softmax in the exp2 domain: logits are scaled by hd^-1/2 * log2(e) up front (v_exp_f32 is exact to ~1 ulp)
  constexpr float LOG2E = 1.44269504088896340736f;
  const float scale = rsqrtf((float)HD) * LOG2E;

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
    for (int idx = tid; idx < KT * HD / 4; idx += NW * 64) {
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
        if (bias && kg < p.Lk) v = fmaf(bias[(long)qrow * p.Lk + kg], LOG2E, v);
        bool masked = kg >= p.Lk;
        if (km && !masked && kg >= p.mask_start) masked = km[kg - p.mask_start] != 0;
        v = masked ? -INFINITY : v;
        s[t][r] = v;
        tmax = fmaxf(tmax, v);
      }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float mnew = fmaxf(mrun, tmax);
    const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
    mrun = mnew;
    float psum = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(s[t][r] - mnew);
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


// ------------------------------------------------------------------------------------------------
// bf16 path (backbone, hd = 64): v_mfma_f32_32x32x16_bf16, fp32 softmax statistics, exp2 domain.
//   K tile [64 keys][64 d] and V tile [64 keys][64 d], bf16 = 64 rows x 128 B each, straight from the qkv buffer.
// Both are staged with global_load_lds_dwordx4 (no VGPR round trip) into the same XOR-swizzled 128-byte-row
// LDS image the GEMM uses, double-buffered.  The second MFMA needs V^T (keys contiguous per lane): that is the
// gfx950 transposing LDS read ds_read_b64_tr_b16 — a 16-lane group reads a [4 keys][16 d] block, lane i supplying
// the address of (key i>>2, d-quad i&3) and receiving the 4 keys of column i (probed on hardware:
// tools/probe_tr_read.hip).  With the transposed formulation the S^T accumulator registers r = 8u..8u+7 of a lane
// are exactly the eight k-slots that lane must feed to the second MFMA, so P goes accumulator ->
// v_cvt_pk_bf16_f32 -> B operand without leaving the lane.
// ------------------------------------------------------------------------------------------------
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16v8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_t;

__device__ __forceinline__ float xhalf_max(float v) {   // max over the two half-waves: one v_permlane32_swap instead of an LDS bpermute
  const u32x2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float v) {
  const u32x2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

constexpr int A16_TILE = 64 * 128;           // 8 KiB per tile
constexpr int A16_STAGE = 2 * A16_TILE;      // K + V^T
constexpr int A16_LDS = 2 * A16_STAGE;       // double buffered: 32 KiB

// Negative result kept for the record: a half-size (32-key) instantiation of the tile body for sequence tails <= 32 keys
// (T = 257 / 325 / 730: tails 1 / 5 / 26, i.e. 1/12 fewer sub-tiles at T = 325) did not change the kernel time (52.0 vs 51.2 us):
// at six key tiles per workgroup the per-tile wait-stage-barrier chain, not the tail tile's arithmetic, sets the time.
// And a K/V-RESIDENT form (one 768-thread workgroup per (image, head), all six K/V tiles staged once, no wait or barrier in the
// key loop): 56.1 vs 50.9 us - slower.  Per tile and wave the SIMD spends ~1450 cycles on VALU (265 instructions, v_exp at 4x)
// plus 512 on MFMA, and the two do not overlap across the three waves of a SIMD here: the kernel is bound by that sum, not by
// staging, barriers or load latency.
// Occupancy: 3 workgroups per CU (136 VGPRs) is the optimum - 51.2 us; a 128-VGPR build at 4 per CU 55.5, LDS-capped 2 per CU 57.0,
// 1 per CU 80.1 (so one wave per SIMD already reaches 64 % of the three-wave rate: the waves are busy, not waiting).
// Likewise a 3-stage K/V ring (two tiles ahead, counted vmcnt(4), raw barrier): 54.9 vs 51.4 us - slower; the double buffer stays.
// __launch_bounds__(256, 2): with a 256-register budget hipcc keeps the MFMA accumulators in VGPRs (no v_accvgpr_read/write
// copies around the softmax: -90 of ~410 VALU instructions per key tile; the kernel is VALU-bound at 16 MFMAs per tile).
// TRACE (debug instantiation, EC_ATTN_TRACE=1): lane 0 of every wave of one mid-grid workgroup stamps s_memtime at the
// milestones of every key tile into p.bias (reused as a uint32 buffer); attention() prints the per-segment cycle counts.
#define A16_STAMP(slot)                                                                          \
  do {                                                                                           \
    if constexpr (TRACE) {                                                                       \
      if (tr_on) {                                                                               \
        const unsigned ts_ = (unsigned)__builtin_amdgcn_s_memtime();                             \
        if (lane == 0) ((unsigned*)p.bias)[wave * 128 + (k0 >> 6) * 8 + (slot)] = ts_;           \
      }                                                                                          \
    }                                                                                            \
  } while (0)
template <bool TRACE, bool F16>
__global__ __launch_bounds__(256, 2) void attn_bf16_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, hi = lane >> 5;
  // XCD-aware workgroup map (round 5).  Workgroups are dealt to the eight XCDs round-robin by their linear id, and each XCD has its
  // own L2: with a (query block, head, image) grid the query blocks of one (image, head) - consecutive ids - landed on DIFFERENT XCDs
  // and every one of them pulled that head's K / V through its own L2 (FETCH_SIZE: exactly 7/3 of the Q|K|V bytes at three query
  // blocks, profiles/r04_qkv_gemm_pmc_kernels.csv).  1-D launch; id L -> xcd = L % 8, slot = L / 8; the nqb query blocks of an item
  // take consecutive slots of ONE xcd: item = (slot / nqb) * 8 + xcd, query block = slot % nqb.  Bijective on the first
  // (items / 8) * 8 * nqb ids; the at most seven items left over keep the plain order.  Measured (cfg2: 3 query blocks x 12 heads x 64
  // images, T = 325; profiles/r05_attn_xcd_ab.txt, the plain order behind a switch that is gone again): FETCH_SIZE x 2 223.7 -> 95.9 MB
  // per launch (2.33 x -> 1.00 x the Q|K|V bytes), 48.9 -> 46.0 us in the model trace (47.1 -> 44.7 us in the counter runs): the re-read is
  // gone and the time moved by 6 % - what binds is the vector pipe (32 v_exp_f32 per lane and key tile at quarter rate, see below).
  int item, qb;
  {
    const int nqb = (p.Lq + 127) >> 7, NI = p.H * p.B, L = blockIdx.x;
    const int Gm = (NI >> 3) * 8 * nqb;
    if (L < Gm) {
      const int slot = L >> 3, grp = slot / nqb;
      item = grp * 8 + (L & 7); qb = slot - grp * nqb;
    } else {
      const int Lt = L - Gm, it = Lt / nqb;
      item = (NI & ~7) + it; qb = Lt - it * nqb;
    }
  }
  const int b = item / p.H, h = item - b * p.H;
  const int q0 = qb * 128 + wave * 32;
  const char* Qb = (const char*)p.Q + ((long)b * p.sQ + h * 64) * 2;
  const char* Kb = (const char*)p.K + ((long)b * p.sK + h * 64) * 2;
  const char* Vb = (const char*)p.V + ((long)b * p.sV + h * 64) * 2;
  const long ldk_b = p.ldk * 2, ldv_b = p.ldv * 2;

  bf16x8 qf[4];
  {
    int qr = q0 + j;
    qr = qr < p.Lq ? qr : p.Lq - 1;
    const char* src = Qb + (long)qr * p.ldq * 2 + hi * 16;
#pragma unroll
    for (int m = 0; m < 4; ++m) qf[m] = *(const bf16x8*)(src + m * 32);
  }
  f32x16 ot[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[d][r] = 0.f;
  float mrun = -1e30f, lrun = 0.f;
  const float c = 0.125f * 1.44269504088896340736f;   // hd^-0.5 * log2(e)

  // Staging by buffer loads to LDS: one descriptor per operand that ends with key row Lk - 1 (rows past it read as zeros: K = 0 gives
  // a score the edge mask removes, V = 0 meets P = 0), a fixed 32-bit per-lane offset and the tile as the scalar offset - no 64-bit
  // address arithmetic per tile (host: one head's K / V rows span < 2 GiB)
  const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Kb), 0, (int)((long)(p.Lk - 1) * ldk_b + 128), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Vb), 0, (int)((long)(p.Lk - 1) * ldv_b + 128), 0x00020000);
  unsigned vok[2], vov[2];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int r = (wave * 2 + jj) * 8 + (lane >> 3);
    const int cc = (lane & 7) ^ ((r >> 1) & 7);
    vok[jj] = (unsigned)r * (unsigned)ldk_b + (unsigned)cc * 16u;
    vov[jj] = (unsigned)r * (unsigned)ldv_b + (unsigned)cc * 16u;
  }
  auto stage = [&](int k0, char* buf) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int rb = wave * 2 + jj;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lptr_t)(buf + rb * 1024), 16, vok[jj], k0 * (int)ldk_b, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lptr_t)(buf + A16_TILE + rb * 1024), 16, vov[jj], k0 * (int)ldv_b, 0, 0);
    }
  };

  stage(0, smem);
  int cur = 0;
  const bool active = q0 < p.Lq;   // a wave whose 32 queries are all past Lq only helps staging (wave-uniform)
  const bool tr_on = TRACE && blockIdx.x == gridDim.x / 2;   // (one mid-grid workgroup)
  for (int k0 = 0; k0 < p.Lk; k0 += 64) {
    A16_STAMP(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    A16_STAMP(1);
    __syncthreads();
    A16_STAMP(2);
    if (k0 + 64 < p.Lk) stage(k0 + 64, smem + (cur ^ 1) * A16_STAGE);
    const char* Kt = smem + cur * A16_STAGE;
    const char* Vtt = Kt + A16_TILE;
    cur ^= 1;
    if (!active) continue;
    const bool edge = k0 + 64 > p.Lk;    // only the last tile needs the key-range mask
    // ... and when at most 32 of its keys exist (T = 325: 5 of 64; T = 730: 26) its second 32-key half is skipped altogether
    // (QK MFMAs, maximum, exponentials, PV MFMAs: half of that tile's work, 1/12 of the kernel at T = 325); wave-uniform
    const bool half_tile = k0 + 32 >= p.Lk;
    f32x16 s[2];
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    {  // all eight K fragments first (one LDS latency instead of eight read -> wait -> MFMA round trips), then the two
       // accumulator chains interleaved so consecutive MFMAs never depend on each other
      bf16x8 ka[2][4];
      if (!half_tile) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int row = t * 32 + j;
#pragma unroll
          for (int m = 0; m < 4; ++m) ka[t][m] = *(const bf16x8*)(Kt + row * 128 + (((2 * m + hi) ^ ((row >> 1) & 7)) << 4));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int t = 0; t < 2; ++t)
            s[t] = mfma32x32x16_h<F16>(ka[t][m], qf[m], m == 0 ? zero16 : s[t]);   // C = inline 0 first
      } else {
#pragma unroll
        for (int m = 0; m < 4; ++m) ka[0][m] = *(const bf16x8*)(Kt + j * 128 + (((2 * m + hi) ^ ((j >> 1) & 7)) << 4));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 4; ++m) s[0] = mfma32x32x16_h<F16>(ka[0][m], qf[m], m == 0 ? zero16 : s[0]);
#pragma unroll
        for (int r = 0; r < 16; ++r) s[1][r] = -INFINITY;   // (never read as scores: every use below is skipped too)
      }
    }
    if (edge) {   // last key tile only (a real branch: the empty asm keeps the compiler from if-converting it into 64 selects)
      asm volatile("" ::: "memory");
      const int lim = p.Lk - k0 - 4 * hi;          // key (t, r) is out of range  <=>  32 t + (r&3) + 8 (r>>2) >= lim
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (32 * t + (r & 3) + 8 * (r >> 2) >= lim) s[t][r] = -INFINITY;
    }
    A16_STAMP(3);
    // raw-score maximum (the scale c > 0 is folded into the exponent below)
    float tmax = -INFINITY;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (t == 1 && half_tile) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[t][r]);
    }
    tmax = xhalf_max(tmax) * c;
    // LAZY running maximum: the reference point mrun only moves when some query's tile maximum exceeds it by more than 2^8
    // (softmax is invariant to the reference; exp2 arguments stay <= 8, so P <= 256 and the row sums stay far inside fp32 /
    // bf16 range).  With 32 queries per wave SOME row sets a new maximum in almost every tile, so an exact running maximum
    // rescales O^T (32 multiplies + bookkeeping, ~110 of ~300 VALU instructions per tile) every time; the lazy form does it
    // on the first tile and then practically never.
    if (!__all(tmax <= mrun + 8.f)) {
      const float mnew = fmaxf(mrun, tmax);
      const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
      mrun = mnew;
      lrun *= alpha;
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[d][r] *= alpha;
    }
    A16_STAMP(4);
    // the kernel is VALU-bound (PMC: 16.6 VALU per MFMA, MFMA pipe 20 % busy): scale-and-shift and the row sums run as packed
    // fp32 pairs (v_pk_fma_f32 / v_pk_add_f32: two scores per instruction); only v_exp_f32 stays one per score
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    f32x2 psum2 = {0.f, 0.f};
    const f32x2 c2 = {c, c}, nm2 = {-mrun, -mrun};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (t == 1 && half_tile) continue;
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 x = __builtin_elementwise_fma(f32x2{s[t][r], s[t][r + 1]}, c2, nm2);
        const f32x2 e = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
        s[t][r] = e[0];
        s[t][r + 1] = e[1];
        psum2 += e;
      }
    }
    lrun += psum2[0] + psum2[1];
    A16_STAMP(5);
    {
      // O^T += V^T P^T in four groups (t, uu) of 16 keys.  V^T fragments by transposing reads: this lane's 16-lane group covers
      // d = 32*dt + 16*G .. +15; lane i of the group points at key row (i>>2) of a 4-key block and d-quad (i&3), and gets the 4 keys of
      // column i.  The reads are inline asm (behind the intrinsic the compiler puts s_waitcnt vmcnt(0) - it cannot tell the read from the
      // LDS-DMA writes in flight - i.e. the NEXT tile's loads, issued at the top of this tile, would have to land before this tile's PV
      // MFMAs), closed by hand, and group g + 1 is requested before the MFMAs of group g.
      const int i16 = lane & 15, G = (lane >> 4) & 1;
      const int rr = 4 * hi + (i16 >> 2);
      unsigned aoff[2][2];   // [d][block]: byte offset of this lane's read inside a 16-key group (the key offset 16 g adds 2 KiB)
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const int chunk = 4 * d + 2 * G + ((i16 >> 1) & 1);
        aoff[d][0] = rr * 128 + ((chunk ^ ((rr >> 1) & 7)) << 4) + (i16 & 1) * 8;
        aoff[d][1] = (rr + 8) * 128 + ((chunk ^ (((rr + 8) >> 1) & 7)) << 4) + (i16 & 1) * 8;
      }
      const unsigned vbase = (unsigned)(unsigned long)(__attribute__((address_space(3))) const char*)Vtt;
      u32x2_t vlo[2][2], vhi[2][2];   // [buffer][d]
      auto issue = [&](int g, int bufi) __attribute__((always_inline)) {
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const unsigned a0 = vbase + (unsigned)g * 2048u + aoff[d][0], a1 = vbase + (unsigned)g * 2048u + aoff[d][1];
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(vlo[bufi][d]) : "v"(a0));
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(vhi[bufi][d]) : "v"(a1));
        }
      };
      const int ng = half_tile ? 2 : 4;
      issue(0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (g >= ng) break;
        const int bufi = g & 1, t = g >> 1, uu = g & 1;
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vlo[bufi][0]), "+v"(vhi[bufi][0]), "+v"(vlo[bufi][1]), "+v"(vhi[bufi][1]));
        if (g + 1 < ng) issue(g + 1, bufi ^ 1);
        f32x8 pv;
#pragma unroll
        for (int e = 0; e < 8; ++e) pv[e] = s[t][8 * uu + e];
        bf16x8 pb;   // P <= 2^8 by the lazy maximum: inside the fp16 range as well
        if constexpr (F16) pb = __builtin_bit_cast(bf16x8, __builtin_convertvector(pv, f16x8));
        else pb = __builtin_bit_cast(bf16x8, __builtin_convertvector(pv, bf16v8));
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          typedef __attribute__((ext_vector_type(4))) unsigned u32x4_;
          const bf16x8 av = __builtin_bit_cast(bf16x8, u32x4_{vlo[bufi][d][0], vlo[bufi][d][1], vhi[bufi][d][0], vhi[bufi][d][1]});
          ot[d] = mfma32x32x16_h<F16>(av, pb, ot[d]);
        }
      }
    }
    A16_STAMP(6);
  }
  lrun = xhalf_sum(lrun);
  const float inv = 1.f / lrun;
  // Output: lane (query j, half hi) holds, per 8-column group g of a d-tile, the 4 columns 8 g + 4 hi .. + 3.  One v_permlane32_swap per
  // packed dword between the groups of a pair gives the lower half-wave all 8 columns of the even group and the upper half-wave
  // those of the odd group: FOUR 16-byte stores per lane instead of eight 8-byte ones (the store tail is issue-bound:
  // MI355X_MICROARCH.md "attention epilogue store tail").  Every lane takes part in the swaps; only the store is predicated.
  char* O = (char*)p.O + ((long)b * p.sO + (long)min(q0 + j, p.Lq - 1) * p.ldo + h * 64) * 2;
  const bool q_ok = q0 + j < p.Lq;
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
      u32x2_t a, c;   // packed columns of group 2 gp (a) and 2 gp + 1 (c)
      a = pack4_h<F16>(f32x4{ot[d][8 * gp] * inv, ot[d][8 * gp + 1] * inv, ot[d][8 * gp + 2] * inv, ot[d][8 * gp + 3] * inv});
      c = pack4_h<F16>(f32x4{ot[d][8 * gp + 4] * inv, ot[d][8 * gp + 5] * inv, ot[d][8 * gp + 6] * inv, ot[d][8 * gp + 7] * inv});
      // swap(vdst = a, src = c): a's upper half-wave <-> c's lower half-wave  =>  lower lanes: (a, c) = own | partner columns of the
      // even group; upper lanes: (a, c) = partner | own columns of the odd group
      const u32x2_t s0 = __builtin_amdgcn_permlane32_swap(a[0], c[0], false, false);
      const u32x2_t s1 = __builtin_amdgcn_permlane32_swap(a[1], c[1], false, false);
      typedef __attribute__((ext_vector_type(4))) unsigned u32x4_;
      const u32x4_ o = {s0[0], s1[0], s0[1], s1[1]};
      if (q_ok) *(u32x4_*)(O + (d * 32 + 16 * gp + 8 * hi) * 2) = o;
    }
}


// (round 2 experiment, removed from the product library in round 3: a software-pipelined persistent form of the kernel above -
// QK^T of unit u+1 | softmax of unit u | PV of unit u-1 as one instruction stream, K/V rings of three 64-key stages - was correct but
// slower, 54.5-58.5 us against 48.0 us: with an exp-heavy vector mix the matrix and vector pipes of a SIMD do not run side by side
// (tools/valu_mfma_probe.hip).  Source: git show 2cb8c78:edgecape_amd/csrc/ec_attn.hip.)

// ------------------------------------------------------------------------------------------------
// bf16x3 ("split") path for the HEAD's attentions in throughput mode (head_precision = EC_BF16X3): fp32 Q/K/V/O in memory,
// every MFMA operand is split in registers / at staging time into hi + lo bf16 (hi = RNE(x), lo = RNE(x - hi)) and each
// product is three v_mfma_f32_32x32x16_bf16 (lo*hi, hi*lo, hi*hi; fp32 accumulate): ~2^-17 relative operand error at
// 16/3 of the fp32 MFMA rate.  Same transposed formulation as above; softmax statistics, masks (key padding,
// encoder_decoder.py:301-304,359-360) and the additive Markov bias (bias_attn.py:188-191) in fp32.
//   * K and V tiles (64 keys) are loaded to registers one tile ahead (global latency under the previous tile's MFMAs),
//     split, and written as four bf16 LDS images (K hi/lo, V hi/lo), 16-byte chunks XOR-swizzled by row.
//   * V^T fragments come from the hi and lo images with ds_read_b64_tr_b16 exactly as in the bf16 kernel.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16v2;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ void split4(const f32x4 x, u32x2& hi, u32x2& lo) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x[2 * q], x[2 * q + 1]}, bf16v2));
    const float r0 = x[2 * q] - __uint_as_float(h << 16);
    const float r1 = x[2 * q + 1] - __uint_as_float(h & 0xffff0000u);
    hi[q] = h;
    lo[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, bf16v2));
  }
}
__device__ __forceinline__ void split8v(const f32x4 x0, const f32x4 x1, bf16x8& hi, bf16x8& lo) {
  u32x2 h0, l0, h1, l1;
  split4(x0, h0, l0);
  split4(x1, h1, l1);
  hi = __builtin_bit_cast(bf16x8, u32x4{h0[0], h0[1], h1[0], h1[1]});
  lo = __builtin_bit_cast(bf16x8, u32x4{l0[0], l0[1], l1[0], l1[1]});
}

// KV16: K and V arrive as IEEE fp16 (the image K|V projections of the mixed head's single-pass fp16 layers store them so: half the
// bytes written by the projection and read here; an fp16 value splits EXACTLY into hi + lo bf16, so the three-MFMA product is exact on it)
// ONE: single-pass fp16 instead of bf16x3 (head_precision = EC_MIXED, the attentions of the single-pass fp16 layers): Q (pre-scaled), K, V
// and the probabilities are rounded to IEEE fp16 (P <= 1: exact running maximum), one v_mfma_f32_32x32x16_f16 per product, only the
// "hi" LDS images are written and read.  Softmax statistics, masks and the bias stay fp32.
template <int HD, bool KV16 = false, bool ONE = false>
__global__ __launch_bounds__(256, 2) void attn_split_kernel(AttnP p) {
  typedef typename std::conditional<KV16, _Float16, float>::type kv_t;
  constexpr int ROWB = HD * 2;             // bytes per bf16 row of an LDS image
  constexpr int IMG = 64 * ROWB;           // one image: 64 keys
  constexpr int KS16 = HD / 16;            // k16 MFMA steps of S^T
  constexpr int DT = HD / 32;              // 32-wide d tiles of O^T
  constexpr int PIECES = 64 * HD / 4 / 256;   // float4 pieces per thread per tile (K and V each)
  __shared__ __attribute__((aligned(16))) char lds[4 * IMG];   // K hi | K lo | V hi | V lo
  char* const Khi = lds; char* const Klo = lds + IMG; char* const Vhi = lds + 2 * IMG; char* const Vlo = lds + 3 * IMG;
  auto swz = [](int row) { return HD == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3); };

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, hi = lane >> 5;
  // XCD-aware workgroup map, as in attn_bf16_kernel (1-D launch: the query blocks of one (image, head) on consecutive slots of ONE XCD,
  // so its K / V - fp32 here, twice the bytes - go through one L2 once instead of through three).  bf16x3 backbone, cfg2: see DESIGN 8c.
  int item, qb;
  {
    const int nqb = (p.Lq + 127) >> 7, NI = p.H * p.B, L = blockIdx.x;
    const int Gm = (NI >> 3) * 8 * nqb;
    if (L < Gm) {
      const int slot = L >> 3, grp = slot / nqb;
      item = grp * 8 + (L & 7); qb = slot - grp * nqb;
    } else {
      const int Lt = L - Gm, it = Lt / nqb;
      item = (NI & ~7) + it; qb = Lt - it * nqb;
    }
  }
  const int b = item / p.H, h = item - b * p.H;
  const int q0 = qb * 128 + wave * 32;
  const float* Q = (const float*)p.Q + (long)b * p.sQ + h * HD;
  const kv_t* K = (const kv_t*)p.K + (long)b * p.sK + h * HD;
  const kv_t* V = (const kv_t*)p.V + (long)b * p.sV + h * HD;
  constexpr float LOG2E = 1.44269504088896340736f;
  const float scale = rsqrtf((float)HD) * LOG2E;

  // Q fragments (B operand of S^T): lane (query j, k-half hi) holds k = 16 m + 8 hi .. +7, pre-scaled, split once
  bf16x8 qh[KS16], ql[KS16];
  {
    int qr = q0 + j;
    qr = qr < p.Lq ? qr : p.Lq - 1;
    const float* src = Q + (long)qr * p.ldq + hi * 8;
#pragma unroll
    for (int m = 0; m < KS16; ++m) {
      f32x4 x0 = *(const f32x4*)(src + m * 16), x1 = *(const f32x4*)(src + m * 16 + 4);
      x0 *= scale; x1 *= scale;
      if constexpr (ONE) {
        const u32x2 a = pack4_h<true>(x0), c = pack4_h<true>(x1);
        qh[m] = __builtin_bit_cast(bf16x8, u32x4{a[0], a[1], c[0], c[1]});
      } else {
        split8v(x0, x1, qh[m], ql[m]);
      }
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

  // register-staged tile: piece it of this thread = row r, float4 column c4 (d = 4 c4 .. +3)
  f32x4 kreg[PIECES], vreg[PIECES];
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int it = 0; it < PIECES; ++it) {
      const int idx = tid + it * 256;
      const int r = idx / (HD / 4), c4 = idx % (HD / 4);
      int kr = k0 + r;
      kr = kr < p.Lk ? kr : p.Lk - 1;
      if constexpr (KV16) {
        kreg[it] = __builtin_convertvector(*(const f16x4*)(K + (long)kr * p.ldk + c4 * 4), f32x4);
        vreg[it] = __builtin_convertvector(*(const f16x4*)(V + (long)kr * p.ldv + c4 * 4), f32x4);
      } else {
        kreg[it] = *(const f32x4*)(K + (long)kr * p.ldk + c4 * 4);
        vreg[it] = *(const f32x4*)(V + (long)kr * p.ldv + c4 * 4);
      }
    }
  };
  auto write_tile = [&]() {
#pragma unroll
    for (int it = 0; it < PIECES; ++it) {
      const int idx = tid + it * 256;
      const int r = idx / (HD / 4), c4 = idx % (HD / 4);
      const int off = r * ROWB + (((c4 >> 1) ^ swz(r)) << 4) + (c4 & 1) * 8;
      if constexpr (ONE) {
        *(u32x2*)(Khi + off) = pack4_h<true>(kreg[it]);
        *(u32x2*)(Vhi + off) = pack4_h<true>(vreg[it]);
      } else {
        u32x2 a, c;
        split4(kreg[it], a, c);
        *(u32x2*)(Khi + off) = a; *(u32x2*)(Klo + off) = c;
        split4(vreg[it], a, c);
        *(u32x2*)(Vhi + off) = a; *(u32x2*)(Vlo + off) = c;
      }
    }
  };

  load_tile(0);
  for (int k0 = 0; k0 < p.Lk; k0 += 64) {
    __syncthreads();            // every wave is done reading the previous tile
    write_tile();
    __syncthreads();
    if (k0 + 64 < p.Lk) load_tile(k0 + 64);   // next tile's global loads fly under this tile's MFMAs

    f32x16 s[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
      const int row = t * 32 + j;
#pragma unroll
      for (int m = 0; m < KS16; ++m) {
        const int off = row * ROWB + (((2 * m + hi) ^ swz(row)) << 4);
        if constexpr (ONE) {
          s[t] = mfma32x32x16_h<true>(*(const bf16x8*)(Khi + off), qh[m], s[t]);
        } else {
          const bf16x8 ah = *(const bf16x8*)(Khi + off), al = *(const bf16x8*)(Klo + off);
          s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, qh[m], s[t], 0, 0, 0);
          s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ql[m], s[t], 0, 0, 0);
          s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, qh[m], s[t], 0, 0, 0);
        }
      }
    }
    // key mask of the tile as 64 bits: lane l looks at key k0 + l once (one coalesced byte load instead of 32 scattered ones
    // per lane: vector-memory instructions cost ~64 cycles of issue each), out-of-range keys count as masked
    bool lm = k0 + lane >= p.Lk;
    if (km && !lm && k0 + lane >= p.mask_start) lm = km[k0 + lane - p.mask_start] != 0;
    const unsigned long long mball = __ballot(lm);
    const unsigned long long mbits = mball >> (4 * hi);             // bit (32 t + (r&3) + 8 (r>>2)) <-> accumulator register r
    if (bias) {   // additive bias [Lq, Lk] of this (batch, head): the lane's keys come in runs of four consecutive k
      const float* brow = bias + (long)qrow * p.Lk + k0 + 4 * hi;
      const bool vec = (p.Lk & 3) == 0;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int kb = k0 + 32 * t + 8 * g + 4 * hi;
          if (vec && kb + 3 < p.Lk) {
            const f32x4 bv = *(const f32x4*)(brow + 32 * t + 8 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) s[t][4 * g + e] = fmaf(bv[e], LOG2E, s[t][4 * g + e]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (kb + e < p.Lk) s[t][4 * g + e] = fmaf(brow[32 * t + 8 * g + e], LOG2E, s[t][4 * g + e]);
          }
        }
    }
    float tmax = -INFINITY;
    if (mball == 0ull) {   // (wave-uniform: a full tile without masked keys - every tile of the backbone's attention but the last - skips the selects)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[t][r]);
    } else {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool masked = (mbits >> (32 * t + (r & 3) + 8 * (r >> 2))) & 1ull;
          const float v = masked ? -INFINITY : s[t][r];
          s[t][r] = v;
          tmax = fmaxf(tmax, v);
        }
    }
    tmax = xhalf_max(tmax);
    const float mnew = fmaxf(mrun, tmax);
    const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
    const bool rose = mnew > mrun;   // (per query: false leaves alpha = 1 exactly)
    mrun = mnew;
    float psum = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(s[t][r] - mnew);
        s[t][r] = e;
        psum += e;
      }
    lrun = lrun * alpha + psum;
    if (__builtin_amdgcn_ballot_w64(rose) != 0ull) {   // no query of the wave raised its maximum: the rescale by exactly 1 is skipped
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[d][r] *= alpha;
    }
    // O^T += V^T P^T.  Accumulator registers 8 uu .. 8 uu + 7 of sub-tile t are exactly the eight k-slots this lane feeds.
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int uu = 0; uu < 2; ++uu) {
        bf16x8 ph, pl;
        if constexpr (ONE) {
          const u32x2 a = pack4_h<true>(f32x4{s[t][8 * uu], s[t][8 * uu + 1], s[t][8 * uu + 2], s[t][8 * uu + 3]});
          const u32x2 c = pack4_h<true>(f32x4{s[t][8 * uu + 4], s[t][8 * uu + 5], s[t][8 * uu + 6], s[t][8 * uu + 7]});
          ph = __builtin_bit_cast(bf16x8, u32x4{a[0], a[1], c[0], c[1]});
        } else {
          split8v(f32x4{s[t][8 * uu], s[t][8 * uu + 1], s[t][8 * uu + 2], s[t][8 * uu + 3]},
                  f32x4{s[t][8 * uu + 4], s[t][8 * uu + 5], s[t][8 * uu + 6], s[t][8 * uu + 7]}, ph, pl);
        }
        const int i16 = lane & 15, G = (lane >> 4) & 1;
        const int krow = 32 * t + 16 * uu + 4 * hi + (i16 >> 2);   // second 4-key block: + 8
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          const int chunk = 4 * d + 2 * G + ((i16 >> 1) & 1);
          const int o0 = krow * ROWB + ((chunk ^ swz(krow)) << 4) + (i16 & 1) * 8;
          const int o1 = (krow + 8) * ROWB + ((chunk ^ swz(krow + 8)) << 4) + (i16 & 1) * 8;
          const s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(Vhi + o0));
          const s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(Vhi + o1));
          bf16x8 vh, vl;
#pragma unroll
          for (int e = 0; e < 4; ++e) { vh[e] = h0[e]; vh[4 + e] = h1[e]; }
          if constexpr (ONE) {
            ot[d] = mfma32x32x16_h<true>(vh, ph, ot[d]);
          } else {
            const s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(Vlo + o0));
            const s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(Vlo + o1));
#pragma unroll
            for (int e = 0; e < 4; ++e) { vl[e] = l0[e]; vl[4 + e] = l1[e]; }
            ot[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph, ot[d], 0, 0, 0);
            ot[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl, ot[d], 0, 0, 0);
            ot[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph, ot[d], 0, 0, 0);
          }
        }
      }
  }
  lrun = xhalf_sum(lrun);
  const float inv = 1.f / lrun;
  if (q0 + j < p.Lq) {
    if (p.o_x3 == 3) {   // fp16x2 row [fp16 | e5m2 lo8 | e5m2 hi8] (ec_common.h split4_x2; ldo / sO in 16-bit units, ldo >= 2 H HD): proj's A operand
      char* O = (char*)p.O + ((long)b * p.sO + (long)(q0 + j) * p.ldo) * 2;
      const long C = (long)p.H * HD;
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v;
          v[0] = ot[d][4 * g] * inv; v[1] = ot[d][4 * g + 1] * inv; v[2] = ot[d][4 * g + 2] * inv; v[3] = ot[d][4 * g + 3] * inv;
          u32x2_t vh;
          unsigned l8, h8;
          split4_x2(v, vh, l8, h8);
          const long c = h * HD + d * 32 + 8 * g + 4 * hi;
          *(u32x2_t*)(O + c * 2) = vh;
          *(unsigned*)(O + 2 * C + c) = l8;
          *(unsigned*)(O + 3 * C + c) = h8;
        }
      return;
    }
    if (p.o_x3) {   // bf16 split [hi | lo], planes H * HD elements apart: the A operand of the K-concatenated proj GEMM (bf16x3 backbone)
      bf16_t* O = (bf16_t*)p.O + (long)b * p.sO + (long)(q0 + j) * p.ldo + h * HD;
      const long plane = (long)p.H * HD;
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v;
          v[0] = ot[d][4 * g] * inv; v[1] = ot[d][4 * g + 1] * inv; v[2] = ot[d][4 * g + 2] * inv; v[3] = ot[d][4 * g + 3] * inv;
          u32x2_t vh, vl;
          split4_bf16(v, vh, vl);
          bf16_t* o = O + d * 32 + 8 * g + 4 * hi;
          *(u32x2_t*)o = vh;
          *(u32x2_t*)(o + plane) = vl;
        }
      return;
    }
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
  dim3 grid((p.Lq + 127) / 128, p.H, p.B);
  EC_REQUIRE(!p.o_x3 || (!p.bf16 && p.split), -1, "attention: the split output exists in the bf16x3 mode only");
  if (!p.bf16 && p.split) {
    EC_REQUIRE(p.ldq % 4 == 0 && p.ldk % 4 == 0 && p.ldv % 4 == 0 && p.ldo % 4 == 0, -1, "attention: strides must be multiples of 4");
    const dim3 g1(grid.x * grid.y * grid.z);   // 1-D: the kernel maps the id to (query block, head, image) itself (XCD-aware)
    if (p.kv16) {
      EC_REQUIRE(p.hd == 64, -1, "attention: fp16 K / V only for head dim 64 (the token -> image cross attention)");
      if (p.one) hipLaunchKernelGGL((attn_split_kernel<64, true, true>), g1, dim3(256), 0, st, p);
      else hipLaunchKernelGGL((attn_split_kernel<64, true>), g1, dim3(256), 0, st, p);
    } else if (p.one) {
      if (p.hd == 64) hipLaunchKernelGGL((attn_split_kernel<64, false, true>), g1, dim3(256), 0, st, p);
      else hipLaunchKernelGGL((attn_split_kernel<32, false, true>), g1, dim3(256), 0, st, p);
    } else if (p.hd == 64) hipLaunchKernelGGL((attn_split_kernel<64>), g1, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((attn_split_kernel<32>), g1, dim3(256), 0, st, p);
    EC_LAUNCH_CHECK();
    return 0;
  }
  if (!p.bf16) {
    EC_REQUIRE(p.ldq % 4 == 0 && p.ldk % 4 == 0 && p.ldv % 4 == 0 && p.ldo % 4 == 0, -1, "attention: strides must be multiples of 4");
    // (2-wave workgroups of 64 queries for the few-workgroup case, Lq = 100, measured slower - 80 vs 65 us at Lk = 324 - removed in round 3)
    if (p.hd == 64) hipLaunchKernelGGL((attn_f32_kernel<64, 4>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((attn_f32_kernel<32, 4>), grid, dim3(256), 0, st, p);
    EC_LAUNCH_CHECK();
    return 0;
  }
  if (p.bf16) {
    EC_REQUIRE(p.hd == 64 && !p.kmask && !p.bias, -1, "attention(bf16): hd = 64, no mask / bias (backbone only)");
    EC_REQUIRE(p.ldq % 8 == 0 && p.ldk % 8 == 0 && p.ldv % 8 == 0 && p.ldo % 4 == 0, -1, "attention(bf16): stride alignment");
    static const bool trace = getenv("EC_ATTN_TRACE") != nullptr;
    if (trace && !p.f16) {
      unsigned* d_tr = nullptr;
      EC_HIP(hipMalloc((void**)&d_tr, 4 * 128 * sizeof(unsigned)));
      EC_HIP(hipMemsetAsync(d_tr, 0, 4 * 128 * sizeof(unsigned), st));
      AttnP q = p;
      q.bias = (const float*)d_tr;
      hipLaunchKernelGGL((attn_bf16_kernel<true, false>), dim3(grid.x * grid.y * grid.z), dim3(256), A16_LDS, st, q);
      EC_LAUNCH_CHECK();
      EC_HIP(hipStreamSynchronize(st));
      unsigned h[4 * 128];
      EC_HIP(hipMemcpy(h, d_tr, sizeof(h), hipMemcpyDeviceToHost));
      (void)hipFree(d_tr);
      const int nt = (p.Lk + 63) / 64;
      for (int w = 0; w < 4; ++w) {
        fprintf(stderr, "[attn trace] wave %d:", w);
        for (int t = 0; t < nt && t < 16; ++t) {
          const unsigned* r = h + w * 128 + t * 8;
          fprintf(stderr, " | wait %u bar %u stage+QK %u max %u exp %u PV %u", r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3],
                  r[5] - r[4], r[6] - r[5]);
          if (t + 1 < nt) fprintf(stderr, " (gap %u)", h[w * 128 + (t + 1) * 8] - r[6]);
        }
        fprintf(stderr, " | total %u\n", h[w * 128 + (nt - 1) * 8 + 6] - h[w * 128]);
      }
      return 0;
    }
    EC_REQUIRE((long)p.Lk * p.ldk * 2 < (1l << 31) && (long)p.Lk * p.ldv * 2 < (1l << 31), -1, "attention(bf16): K / V rows of one head beyond 2 GiB");
    if (p.f16) hipLaunchKernelGGL((attn_bf16_kernel<false, true>), dim3(grid.x * grid.y * grid.z), dim3(256), A16_LDS, st, p);
    else hipLaunchKernelGGL((attn_bf16_kernel<false, false>), dim3(grid.x * grid.y * grid.z), dim3(256), A16_LDS, st, p);
    EC_LAUNCH_CHECK();
    return 0;
  }
  return -1;
}

}  // namespace ec
