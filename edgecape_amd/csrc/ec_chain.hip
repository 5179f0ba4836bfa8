// Row-chain kernel for the head of the EdgeCape hot path on gfx950 (MI355X): a SEQUENCE of Linear layers (+ bias, positional
// table, activation, residual, LayerNorm) applied to a slab of 32 token rows by ONE workgroup, the intermediate activations never
// leaving the CU.
//
// Reference ops: every row-wise stretch of a transformer layer of the head between two
// row-mixing operators (attention over the tokens of a sample, GCN aggregation over the skeleton):
//     self-attn out-proj + residual + norm1 -> cross-attn query projection         (encoder_decoder.py:596-611)
//     cross-attn out-proj∘choker + residual + norm2 -> ffn1                        (encoder_decoder.py:618-634)
//     ffn2 + residual + norm3 -> next layer's self-attn in-proj (-> image-to-token K|V)   (encoder_decoder.py:634-649)
//     encoder: out-proj + residual + norm1 -> linear1 + ReLU -> linear2 + residual + norm2 -> next in-proj   (461-483)
// As separate launches each of these M = 3200-row GEMMs (0.4 GFLOP) costs 8-17 us (dependent-launch floor + a cold pass over
// memory) and the LayerNorms 7 us: a decoder layer was 13 launches.  Here a stage's [32, N] output is written straight
// into LDS as the next stage's MFMA operand.
//
// Structure (MI355X-first):
//   * one 512-thread workgroup (8 waves) per 32-row slab; a stage's N output columns are produced in passes of 256; inside a
//     pass wave w owns the two 16-column fragments 2w, 2w+1 and both 16-row fragments: acc[2][2] of v_mfma_f32_16x16x32_bf16.
//   * bf16x3 arithmetic as in the rest of the head (ec_gemm.hip GM_SPLIT): activations and weights are hi + lo bf16 pairs,
//     acc += Whi*Xhi + Whi*Xlo + Wlo*Xhi (fp32 accumulate) - ~2^-17 relative operand error.
//   * activations live in LDS already split: per row, per 32-k block, [32 hi | 32 lo] bf16 (128 B), row pitch K*4 + 16 B; an MFMA
//     B-operand fragment is ONE ds_read_b128 per plane.  Global inputs are split while they are staged; a stage's epilogue
//     writes its output in the same form.
//   * weights stream from L2 straight into registers (no LDS: no two waves share a weight fragment), FRAGMENT-MAJOR packed at
//     ec_finalize (pack_chain_weights) so that every wave-load is one contiguous KiB; the K loop is software-pipelined in batches
//     of four k-blocks (16 loads in flight per wave while the previous batch's 48 MFMAs run).
//   * the MFMA roles are swapped (A-operand <- weight rows, B-operand <- token rows): a lane's accumulator quad is ONE row and 4
//     consecutive columns, so bias / table / residual / output are 16-byte accesses and the split write-back is two ds_write_b64.
//   * LayerNorm (N = 256 = one pass): two-pass statistics (mean, then centred squares) reduced over the four lanes of a row, then
//     over the eight waves through 2 KiB of LDS.
// What bounds it: the weight stream.  A workgroup reads every stage's weights once (1-2 MB per chain) at the per-CU L2 rate.
#include "ec_chain.h"

namespace ec {
namespace {

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16v2;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int CH_RED = CH_LDS0;   // LDS bytes in front of the activation buffers: LayerNorm partials (2 x [32 rows][8 waves] fp32)

// 4 floats -> 4 hi bf16 (RNE) + 4 lo bf16 (RNE of the exact remainder)
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

// 4 floats -> 4 fp16 (RNE): the single-pass form's only plane
__device__ __forceinline__ u32x2 half4(const f32x4 x) {
  return __builtin_bit_cast(u32x2, pack4_h<true>(x));   // (saturating)
}

struct WBatch { bf16x8 w[4][2][2]; };   // [k-block of the batch][n-fragment][plane]

template <bool TRACE>
__global__ __launch_bounds__(512) void chain_kernel(ChainP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  // TRACE (EC_CHAIN_TRACE=1): thread 0 of one mid-grid workgroup stamps s_memtime at the milestones of every stage
  const bool tr_on = TRACE && blockIdx.x == gridDim.x / 2 && tid == 0;
  int tr_i = 0;
  auto stamp = [&]() __attribute__((always_inline)) {
    if constexpr (TRACE) {
      if (tr_on && tr_i < 63) p.trace[tr_i++] = (unsigned)__builtin_amdgcn_s_memtime();
    }
  };
  stamp();
  int fi = 0;   // fine stamps of stage p.trace_stage: trace[64 + fi]
  auto fstamp = [&](bool on) __attribute__((always_inline)) {
    if constexpr (TRACE) {
      if (tr_on && on && fi < 63) p.trace[64 + fi++] = (unsigned)__builtin_amdgcn_s_memtime();
    }
  };
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slab = blockIdx.x / p.split, part = blockIdx.x - slab * p.split;
  const int row0 = slab * CH_BM;
  const int lrow = lane & 15, lq = lane >> 4;
  const bool h1 = p.h1 != 0;
  float* const red = (float*)smem;
  // Row compaction (ChainP::rowmap): slab row r is token row grow(r); rows at or past n_rows are computed on a duplicate of the last
  // row and never stored (as the edge rows of the last slab always were).  The map of the slab's 32 rows sits in LDS behind the
  // LayerNorm / keypoint-tail scratch (first read behind a stage's opening barrier); this lane's own two rows also in registers
  // (the first stage's residual is requested before any barrier).
  const int n_rows = p.n_active ? min(*p.n_active, p.rows) : p.rows;
  if (row0 >= n_rows) return;                       // whole workgroup, before any barrier
  int* const rowm = (int*)(smem + 3072);            // [32] (CH_RED = 4096: LayerNorm partials 0..2047, keypoint tail ..2303)
  auto map_row = [&](int i) { const int c = min(i, n_rows - 1); return p.rowmap ? p.rowmap[c] : c; };
  // Fan-out of the representative rows (ChainP::fan_*): slab row r is also written to the token rows fbase[r] + k of the set bits k
  // of fbits[r] (two 64-bit words: K <= 128); frep: bit r set <=> slab row r has such rows (wave 0's ballot)
  int* const fbase = (int*)(smem + 3200);                                     // [32]
  unsigned long long* const fbits = (unsigned long long*)(smem + 3328);       // [32][2]
  unsigned* const frep = (unsigned*)(smem + 3840);
  if (tid < 64) {
    unsigned long long b0 = 0ull, b1 = 0ull;
    if (tid < CH_BM) {
      rowm[tid] = map_row(row0 + tid);
      const bool on = p.fan_bits != nullptr && row0 + tid < n_rows;
      if (on) { b0 = p.fan_bits[(long)(row0 + tid) * 2]; b1 = p.fan_bits[(long)(row0 + tid) * 2 + 1]; }
      fbase[tid] = on ? p.fan_base[row0 + tid] : 0;
      fbits[tid * 2] = b0;
      fbits[tid * 2 + 1] = b1;
    }
    const unsigned long long any = __ballot((b0 | b1) != 0ull);
    if (tid == 0) *frep = (unsigned)any;
  }
  int grow_l[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) grow_l[mi] = map_row(row0 + mi * 16 + lrow);

  f32x4 keep[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) keep[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int s = 0; s < p.n_stages; ++s) {
    const ChainStage& S = p.st[s];
    const int nfr = S.N >> 4, nkb = S.K >> 5, nkb1 = S.k1 >> 5;
    const long pitch_a = (long)S.k1 * 4 + 16, pitch_b = (long)(S.K - S.k1) * 4 + 16;
    const char* xa = smem + S.a_off + lrow * pitch_a + lq * 16;
    const char* xb = smem + S.b_off + lrow * pitch_b + lq * 16;
    const bool ln = S.ln_w != nullptr;

    // ---- this wave's weight stream: passes of 256 columns (wave w owns fragments pass*16 + 2w, +1 while they exist; host:
    // N % 32 == 0), each pass nb = K/128 BATCHES of four k-blocks (16 x 1 KiB wave-loads).  The stream is one loop over all
    // batches of the stage, double-buffered in registers, running across pass boundaries; every load is issued
    // UNCONDITIONALLY - past the end of the stream the buffer offset is out of range, which returns zeros without touching
    // memory - so the compiler's own vmcnt bookkeeping stays exact (16 loads in flight under every batch of 48 MFMAs; with the
    // prefetch inside an `if` it falls back to vmcnt(0) before every batch).
    const int nb = nkb >> 2;
    const int npass_all = wave * 2 < nfr ? (nfr - wave * 2 + 15) >> 4 : 0;   // passes in which this wave has fragments
    // two workgroups per slab (p.split = 2): a stage that later stages read is computed by both, any other is dealt out by passes
    const bool shared = S.s_off >= 0 || S.keep || ln || S.kp_w;
    const int pstep = shared ? 1 : p.split, p0 = shared ? 0 : part;
    const int npass = npass_all > p0 ? (npass_all - p0 + pstep - 1) / pstep : 0;
    const bool store_out = S.out != nullptr && (!shared || part == 0);
    const int T = npass * nb;
    // Every workgroup walks the SAME weights: started together they would all pull the same few KiB through the same L2 channels at
    // the same time, so workgroup b starts its column passes rot_p passes further on.  (The k order is NOT rotated although that
    // measured another 5 % on the chain: a row's fp32 summation order must not depend on the slab it sits in - identical tokens,
    // e.g. the padded keypoint slots of a sample, produce bit-identical outputs, as they do in the reference.)
    const int rot_b = 0, rot_p = npass ? slab % npass : 0;
    auto pass_of = [&](int ps) { const int q = ps + rot_p; return p0 + (q >= npass ? q - npass : q) * pstep; };
    auto batch_of = [&](int bb) { const int q = bb + rot_b; return q >= nb ? q - nb : q; };
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(S.W), 0, S.N * S.K * 4, 0x00020000);
    auto load_batch = [&](WBatch& b, int i) {
      const int ps = i / nb, kb0 = batch_of(i - ps * nb) << 2;
      // fragment (f, kb, plane) = 1 KiB at ((f * nkb + kb) * 2 + plane) * 1024, lane-linear
      const unsigned v0 = i < T ? (unsigned)(((pass_of(ps) * 16 + wave * 2) * nkb + kb0) * 2048 + lane * 16) : 0x80000000u;
      const unsigned v1 = v0 + (unsigned)nkb * 2048u;
      // single-pass fp16: the lo planes are requested out of range (zeros, no memory traffic; the load count stays the same)
      const unsigned l0 = h1 ? 0x80000000u : v0, l1 = h1 ? 0x80000000u : v1;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        b.w[k][0][0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsW, v0 + k * 2048, 0, 0));
        b.w[k][0][1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsW, l0 + (1024 + k * 2048), 0, 0));
        b.w[k][1][0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsW, v1 + k * 2048, 0, 0));
        b.w[k][1][1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsW, l1 + (1024 + k * 2048), 0, 0));
      }
    };

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute_batch = [&](const WBatch& b, int kb0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int kb = kb0 + k;
        const char* x = kb < nkb1 ? xa + kb * 128 : xb + (kb - nkb1) * 128;
        const long pitch = kb < nkb1 ? pitch_a : pitch_b;
        bf16x8 xh[2], xl[2];
        if (h1) {   // (wave-uniform) single-pass fp16: hi planes only, one MFMA per product
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) xh[mi] = *(const bf16x8*)(x + mi * 16 * pitch);
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) acc[mi][j] = mfma16x16x32_h<true>(b.w[k][j][0], xh[mi], acc[mi][j]);
          continue;
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          xh[mi] = *(const bf16x8*)(x + mi * 16 * pitch);
          xl[mi] = *(const bf16x8*)(x + mi * 16 * pitch + 64);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) acc[mi][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b.w[k][j][0], xh[mi], acc[mi][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) acc[mi][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b.w[k][j][0], xl[mi], acc[mi][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) acc[mi][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b.w[k][j][1], xh[mi], acc[mi][j], 0, 0, 0);
      }
    };

    f32x4 pbias[2], presid[2][2];
    auto load_bias = [&](int f0) {   // bias of fragments f0, f0 + 1 (clamped past the end: loaded, never used)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        pbias[j] = S.bias ? *(const f32x4*)(S.bias + min((f0 + j) * 16, S.N - 16) + lq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    };

    // ---- the end of a pass: epilogue on acc (fragments f0, f0 + 1), then acc = 0
    auto finish_pass = [&](int f0, int f0_next) {
      // part 1: bias, positional table, activation, residual
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = (f0 + j) * 16 + lq * 4;
        const f32x4 bias = pbias[j];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const int gr = grow_l[mi];
          f32x4 v = acc[mi][j] + bias;
          if (S.table) v += *(const f32x4*)(S.table + (long)(gr % S.period) * S.ldt + n);
          if (S.act == ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          } else if (S.act == ACT_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_fast32(v[e]);
          }
          if (S.resid) v += presid[mi][j];
          if (S.resid_keep) v += keep[mi][j];
          acc[mi][j] = v;
        }
      }
      fstamp(s == p.trace_stage);   // F: part 1 done (bias / act / resid)
      if (ln) {
        // LayerNorm over the N = 256 columns of the (only) pass - every wave is here: mean, then centred squares
        float mean[2], rstd[2];
        f32x4 lg[2], lb[2];   // scale / shift: requested here, they land under the two reductions
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          lg[j] = *(const f32x4*)(S.ln_w + (f0 + j) * 16 + lq * 4);
          lb[j] = *(const f32x4*)(S.ln_b + (f0 + j) * 16 + lq * 4);
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          float sm = 0.f;
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) sm += acc[mi][j][e];
          sm += __shfl_xor(sm, 16, 64);
          sm += __shfl_xor(sm, 32, 64);
          if (lq == 0) red[(mi * 16 + lrow) * 8 + wave] = sm;
        }
        __syncthreads();
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const f32x4 a = *(const f32x4*)(red + (mi * 16 + lrow) * 8), b = *(const f32x4*)(red + (mi * 16 + lrow) * 8 + 4);
          mean[mi] = (((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3]))) * (1.f / 256.f);
          float sq = 0.f;
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float dlt = acc[mi][j][e] - mean[mi];
              sq = fmaf(dlt, dlt, sq);
            }
          sq += __shfl_xor(sq, 16, 64);
          sq += __shfl_xor(sq, 32, 64);
          if (lq == 0) red[256 + (mi * 16 + lrow) * 8 + wave] = sq;
        }
        __syncthreads();
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const f32x4 a = *(const f32x4*)(red + 256 + (mi * 16 + lrow) * 8), b = *(const f32x4*)(red + 256 + (mi * 16 + lrow) * 8 + 4);
          const float var = (((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3]))) * (1.f / 256.f);
          rstd[mi] = 1.f / sqrtf(var + S.eps);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) acc[mi][j] = (acc[mi][j] - mean[mi]) * rstd[mi] * lg[j] + lb[j];
        }
      }
      fstamp(s == p.trace_stage);   // F: LayerNorm done
      const bool kp_sine = S.kp_w && S.kp_dim_t;
      if (S.kp_w) {
        // keypoint-branch tail (see ChainStage): two dot products per row over the 256 columns - 4 columns x 2 fragments per lane, the
        // 4 lanes of a row by shuffles, the 8 waves through LDS - then the reference-point update by one thread per (row, coordinate)
        float d0[2] = {0.f, 0.f}, d1[2] = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int n = (f0 + j) * 16 + lq * 4;
          const f32x4 w0 = *(const f32x4*)(S.kp_w + n), w1 = *(const f32x4*)(S.kp_w + 256 + n);
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              d0[mi] = fmaf(acc[mi][j][e], w0[e], d0[mi]);
              d1[mi] = fmaf(acc[mi][j][e], w1[e], d1[mi]);
            }
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          d0[mi] += __shfl_xor(d0[mi], 16, 64); d0[mi] += __shfl_xor(d0[mi], 32, 64);
          d1[mi] += __shfl_xor(d1[mi], 16, 64); d1[mi] += __shfl_xor(d1[mi], 32, 64);
          if (lq == 0) {
            red[(mi * 16 + lrow) * 8 + wave] = d0[mi];
            red[256 + (mi * 16 + lrow) * 8 + wave] = d1[mi];
          }
        }
        __syncthreads();
        float* const coord = red + 512;   // [32 rows][2]: (x, y) of b_next, read by the sine embedding below
        if (tid < 64) {
          const int r = tid >> 1, c = tid & 1;
          const float* q = red + c * 256 + r * 8;
          const float dl = (((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]))) + S.kp_b[c];
          const int gr = rowm[r];
          float x = S.kp_prev[(long)gr * 2 + c];   // inverse_sigmoid, eps 1e-3 (head.py:27-31)
          x = fminf(fmaxf(x, 0.f), 1.f);
          const float z = dl + logf(fmaxf(x, 1e-3f) / fmaxf(1.f - x, 1e-3f));
          const float bn = 1.f / (1.f + expf(-z));
          coord[r * 2 + c] = bn;
          if (row0 + r < n_rows && part == 0) S.kp_next[(long)gr * 2 + c] = bn;
        }
        __syncthreads();
        if (p.fan_bits && part == 0 && tid < 256) {
          // (row compaction) the sample's other masked tokens take the representative's point: thread (k, coordinate) per representative
          const int k = tid >> 1, cc = tid & 1;
          for (unsigned rm = __builtin_amdgcn_readfirstlane(*frep); rm; rm &= rm - 1u) {
            const int r = __builtin_ctz(rm);
            const unsigned long long w = fbits[r * 2 + (k >> 6)];
            if ((w >> (k & 63)) & 1ull) S.kp_next[(long)(fbase[r] + k) * 2 + cc] = coord[r * 2 + cc];
          }
        }
        if (kp_sine) {   // sine embedding of b_next -> the operand buffer of the next stage (positional_encoding.py; sincos_kernel)
          for (int qd = tid; qd < CH_BM * 64; qd += 512) {
            const int r = qd >> 6, c = (qd & 63) << 2;           // 4 consecutive features of row r
            const int isx = c >= 128, i0 = c & 127;
            const float e = coord[r * 2 + (isx ? 0 : 1)] * 6.283185307179586f;
            const f32x4 dt = *(const f32x4*)(S.kp_dim_t + i0);
            const f32x4 v = {sinf(e / dt[0]), cosf(e / dt[1]), sinf(e / dt[2]), cosf(e / dt[3])};   // even feature: sin, odd: cos
            char* dst = smem + S.s_off + r * (256l * 4 + 16) + (c >> 5) * 128 + (c & 31) * 2;
            if (h1) {
              *(u32x2*)dst = half4(v);
            } else {
              u32x2 hi, lo;
              split4(v, hi, lo);
              *(u32x2*)dst = hi;
              *(u32x2*)(dst + 64) = lo;
            }
          }
        }
      }
      // part 2: global store, split write-back for the next stage, register copy for a later residual
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = (f0 + j) * 16 + lq * 4;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const int r = mi * 16 + lrow;
          if (S.post_table)   // (encoder: src = norm2(..) + pos is what the next layer's q, k AND v read, encoder_decoder.py:461-470)
            acc[mi][j] += *(const f32x4*)(S.post_table + (long)(grow_l[mi] % S.post_period) * S.ldpt + n);
          if (store_out && row0 + r < n_rows) *(f32x4*)(S.out + (long)grow_l[mi] * S.ldo + n) = acc[mi][j];
          if (S.s_off >= 0 && !kp_sine) {
            char* dst = smem + S.s_off + r * ((long)S.N * 4 + 16) + (n >> 5) * 128 + (n & 31) * 2;
            if (h1) {
              *(u32x2*)dst = half4(acc[mi][j]);
            } else {
              u32x2 hi, lo;
              split4(acc[mi][j], hi, lo);
              *(u32x2*)dst = hi;
              *(u32x2*)(dst + 64) = lo;
            }
          }
          if (S.keep) keep[mi][j] = acc[mi][j];
          acc[mi][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      load_bias(f0_next);   // the next pass's, a whole pass ahead
      fstamp(s == p.trace_stage);   // F: part 2 done (stores, LDS write-back)
    };

    WBatch b0, b1;
    int bi = 0, ps = 0;                 // batch inside the pass, pass (both in stream order)
    auto step = [&]() {
      if (++bi == nb) {
        finish_pass(pass_of(ps) * 16 + wave * 2, pass_of(ps + 1 < npass ? ps + 1 : 0) * 16 + wave * 2);
        bi = 0;
        ++ps;
      }
    };
    // The first weight batch and the first pass's epilogue operands are requested BEFORE the stage's input is staged: they
    // land under the staging and its two barriers instead of in front of the first MFMA / inside the epilogue.
    stamp();   // stage start
    // (round 3, measured and reverted: requesting BOTH register buffers here.  EC_CHAIN_TRACE_STAGE stamps: a batch takes ~2 400
    //  cycles to arrive whether it carries 16 or 48 MFMAs; with two batches in flight at the stage start the first barrier simply
    //  waits for both - "loads + barrier" 3 000 -> 4 500-5 000 cycles, K loop -1 000, whole chains +2...+6 k cycles, and the second
    //  buffer's longer live range spills 8 VGPRs.  The batches arrive at the CU's ~53 B/clk one behind the other.)
    load_batch(b0, 0);
    load_bias(pass_of(0) * 16 + wave * 2);
    if (S.resid) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)   // (LayerNorm stages, N = 256: every wave has its two fragments)
          presid[mi][j] = *(const f32x4*)(S.resid + (long)grow_l[mi] * S.ldr + (wave * 2 + j) * 16 + lq * 4);
    }
    __syncthreads();   // every wave is done with the previous stage's operand buffers (this stage may re-stage one of them)
    stamp();   // barrier passed
    if (S.g_k > 0) {
      // ---- stage a global fp32 input [rows, g_k] into LDS, split: one f32x4 per thread and step
      // (four loads per thread in flight before the first split: one load per trip paid a full memory latency per 8 KiB)
      const int q4 = S.g_k >> 2, total = CH_BM * q4;
      const long pitch = (long)S.g_k * 4 + 16;
      for (int base = tid; base < total; base += 4 * 512) {
        f32x4 v[4];
        int r[4], c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = min(base + u * 512, total - 1);
          r[u] = idx / q4; c[u] = (idx - r[u] * q4) << 2;
          const int gr = rowm[r[u]];   // rows past the edge: duplicated, computed, never stored
          v[u] = *(const f32x4*)(S.g_in + (long)gr * S.ld_in + c[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (base + u * 512 >= total) continue;
          char* dst = smem + S.g_off + r[u] * pitch + (c[u] >> 5) * 128 + (c[u] & 31) * 2;
          if (h1) {
            *(u32x2*)dst = half4(v[u]);
          } else {
            u32x2 hi, lo;
            split4(v[u], hi, lo);
            *(u32x2*)dst = hi;
            *(u32x2*)(dst + 64) = lo;
          }
        }
      }
    }
    __syncthreads();
    stamp();   // input staged

    // (sched_barrier: the machine scheduler otherwise sinks the prefetch three quarters into the batch it should run under)
    for (int i = 0; i < T;) {
      load_batch(b1, i + 1);
      __builtin_amdgcn_sched_barrier(0);
      compute_batch(b0, batch_of(bi) << 2);
      fstamp(s == p.trace_stage);   // F: batch computed
      step();
      if (++i >= T) break;
      load_batch(b0, i + 1);
      __builtin_amdgcn_sched_barrier(0);
      compute_batch(b1, batch_of(bi) << 2);
      fstamp(s == p.trace_stage);   // F: batch computed
      step();
      ++i;
    }
    stamp();   // stage done (K loop + epilogues)
    if (p.fan_bits && store_out) {
      // Row compaction: the masked token rows that were not computed equal their sample's representative row.  Its freshly stored
      // output row (this workgroup's own stores: visible to the whole workgroup behind the barrier) is copied to them.  Thread t holds
      // the 16-byte chunk t % (N/4) of the row in registers; the 512 / (N/4) thread groups deal the sample's K tokens out between them
      // (token k to group k % G); a group is whole waves (N % 256 == 0), so "is token k masked" is a scalar bit test on the sample's mask
      // words and a wave only issues the stores that exist - nothing is read from memory between them.
      // (Round 4 did this with a bcast_rows launch behind every chain: 27 launches per step on the head's dependent lanes.  In-kernel
      //  forms that read a destination list from memory, or walked the mask bits with per-lane arithmetic, cost 15-30 us per chain.)
      __syncthreads();
      const int q4 = S.N >> 2;                          // host: N % 256 == 0, N <= 2048
      const int G = 512 / q4;
      const int c = tid % q4, g = __builtin_amdgcn_readfirstlane(tid / q4);
      // two workgroups per slab: a dealt-out stage's 256-column passes alternate between the parts, and each part copies the columns
      // it computed itself (a wave's 64 chunks are one pass's 256 columns)
      const bool cols_mine = shared || __builtin_amdgcn_readfirstlane(((c >> 6) % p.split) == part);
      if (g < G && cols_mine) {
        for (unsigned rm = __builtin_amdgcn_readfirstlane(*frep); rm; rm &= rm - 1u) {
          const int r = __builtin_ctz(rm);
          const unsigned long long w0 = fbits[r * 2], w1 = fbits[r * 2 + 1];   // (LDS broadcast reads: the same value in every lane)
          // (readfirstlane returns a signed int: without the casts the low word is sign-extended over the high one)
          const unsigned long long b0 = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(w0 >> 32)) << 32) |
                                        (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)w0);
          const unsigned long long b1 = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(w1 >> 32)) << 32) |
                                        (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)w1);
          const int base = __builtin_amdgcn_readfirstlane(fbase[r]);
          const f32x4 v = *(const f32x4*)(S.out + (long)__builtin_amdgcn_readfirstlane(rowm[r]) * S.ldo + c * 4);
          float* const dst0 = S.out + (long)base * S.ldo + c * 4;
          for (int k = g; k < 128; k += G) {
            const unsigned long long w = k < 64 ? b0 : b1;
            if ((w >> (k & 63)) & 1ull) *(f32x4*)(dst0 + (long)k * S.ldo) = v;
          }
        }
      }
    }
  }
}

}  // namespace

// host: W [N, K] fp32 (nn.Linear layout) -> fragment-major bf16x3 packing, N*K*4 bytes:
//   fragment (f = n / 16, kb = k / 32, plane) is 1 KiB: lane l holds the 8 bf16 W[f*16 + (l & 15)][kb*32 + (l >> 4)*8 .. +7]
//   (hi plane: RNE(w); lo plane: RNE(w - hi)); fragments ordered ((f * K/32 + kb) * 2 + plane).
void pack_chain_weights(const float* W, long N, long K, void* out, bool h1) {
  bf16_t* o = (bf16_t*)out;
  const long nkb = K / 32;
  for (long f = 0; f < N / 16; ++f)
    for (long kb = 0; kb < nkb; ++kb)
      for (int l = 0; l < 64; ++l) {
        const float* src = W + (f * 16 + (l & 15)) * K + kb * 32 + (l >> 4) * 8;
        bf16_t* hi = o + ((f * nkb + kb) * 2) * 512 + l * 8;
        bf16_t* lo = hi + 512;
        for (int i = 0; i < 8; ++i) {
          if (h1) { hi[i] = f2half_host(src[i]); lo[i] = 0; continue; }
          const bf16_t h = f2bf(src[i]);
          hi[i] = h;
          lo[i] = f2bf(src[i] - bf2f(h));
        }
      }
}

int chain_layout_bytes(int k) { return CH_BM * (k * 4 + 16); }

namespace {
struct ChDev { bool attr_done = false; };
ChDev ch_dev[64];
}  // namespace

int run_chain(const ChainP& p, hipStream_t st) {
  EC_REQUIRE(p.rows > 0 && p.n_stages >= 1 && p.n_stages <= CH_MAX_STAGES, -1, "chain: bad stage count");
  EC_REQUIRE(p.split == 1 || p.split == 2, -1, "chain: split");
  EC_REQUIRE(!p.fan_bits || (p.rowmap && p.fan_base), -1, "chain: row fan-out needs a row map");
  EC_REQUIRE(p.lds_bytes >= CH_RED && p.lds_bytes <= 160 * 1024, -1, "chain: LDS layout does not fit");
  for (int s = 0; s < p.n_stages; ++s) {
    const ChainStage& S = p.st[s];
    EC_REQUIRE(S.W && S.N % 32 == 0 && S.K % 128 == 0 && S.k1 % 32 == 0 && S.k1 > 0 && S.k1 <= S.K, -1, "chain: stage shape");
    EC_REQUIRE(!S.ln_w || (S.N == 256 && S.ln_b), -1, "chain: LayerNorm needs N = 256");
    EC_REQUIRE(!(S.keep || S.resid_keep) || S.N == 256, -1, "chain: register-kept tiles need N = 256");
    EC_REQUIRE(!S.resid || S.N == 256, -1, "chain: a global residual is loaded once per stage for the single 256-column pass: N = 256");
    EC_REQUIRE(S.a_off >= CH_RED && S.a_off + chain_layout_bytes(S.k1) <= p.lds_bytes, -1, "chain: operand buffer A out of range");
    EC_REQUIRE(S.k1 == S.K || (S.b_off >= CH_RED && S.b_off + chain_layout_bytes(S.K - S.k1) <= p.lds_bytes), -1, "chain: operand buffer B out of range");
    EC_REQUIRE(S.g_k == 0 || (S.g_in && S.g_k % 32 == 0 && S.g_off >= CH_RED && S.g_off + chain_layout_bytes(S.g_k) <= p.lds_bytes), -1, "chain: staged input out of range");
    EC_REQUIRE(S.s_off < 0 || (S.s_off >= CH_RED && S.s_off + chain_layout_bytes(S.N) <= p.lds_bytes), -1, "chain: output buffer out of range");
    EC_REQUIRE(!S.table || S.period > 0, -1, "chain: table period");
    EC_REQUIRE(!S.post_table || S.post_period > 0, -1, "chain: post-table period");
    EC_REQUIRE(!S.kp_w || (S.N == 256 && !S.ln_w && S.kp_b && S.kp_prev && S.kp_next && (!S.kp_dim_t || S.s_off >= 0)), -1, "chain: keypoint tail needs N = 256, no LayerNorm");
    EC_REQUIRE((S.h1 != 0) == (p.h1 != 0), -1, "chain: stages packed for different arithmetic");
    EC_REQUIRE(!p.fan_bits || !S.out || (S.N <= 2048 && S.N % 256 == 0), -1, "chain: row fan-out needs N % 256 == 0, N <= 2048");
    EC_REQUIRE(p.split == 1 || !(S.s_off >= 0 || S.keep || S.ln_w) || !S.resid || !S.out || S.resid != S.out, -1,
               "chain: split chains need out != resid in the stages both workgroups compute");
  }
  int dev = 0;
  EC_HIP(hipGetDevice(&dev));
  EC_REQUIRE(dev >= 0 && dev < 64, -1, "chain: device ordinal out of range");
  if (!ch_dev[dev].attr_done) {
    EC_HIP(hipFuncSetAttribute((const void*)chain_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    EC_HIP(hipFuncSetAttribute((const void*)chain_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    ch_dev[dev].attr_done = true;
  }
  const dim3 grid(((p.rows + CH_BM - 1) / CH_BM) * p.split);
  static const bool trace = getenv("EC_CHAIN_TRACE") != nullptr;
  if (trace) {   // diagnostics: per-stage cycle counts of one workgroup (synchronises the stream: not for timing runs)
    unsigned* d_tr = nullptr;
    EC_HIP(hipMalloc((void**)&d_tr, 128 * sizeof(unsigned)));
    EC_HIP(hipMemsetAsync(d_tr, 0, 128 * sizeof(unsigned), st));
    ChainP q = p;
    q.trace = d_tr;
    q.trace_stage = getenv("EC_CHAIN_TRACE_STAGE") ? atoi(getenv("EC_CHAIN_TRACE_STAGE")) : -1;
    static const int warm = atoi(getenv("EC_CHAIN_TRACE"));   // 2: run the launch once untraced first (weights warm in the L2s)
    if (warm == 2) {
      hipLaunchKernelGGL(chain_kernel<false>, grid, dim3(512), p.lds_bytes, st, p);
      EC_HIP(hipStreamSynchronize(st));
    }
    hipLaunchKernelGGL(chain_kernel<true>, grid, dim3(512), p.lds_bytes, st, q);
    EC_LAUNCH_CHECK();
    EC_HIP(hipStreamSynchronize(st));
    unsigned h[128];
    EC_HIP(hipMemcpy(h, d_tr, sizeof(h), hipMemcpyDeviceToHost));
    (void)hipFree(d_tr);
    fprintf(stderr, "[chain trace] rows %d stages %d split %d h1 %d:", p.rows, p.n_stages, p.split, p.h1);
    for (int s = 0; s < p.n_stages; ++s) {
      const unsigned* t = h + 1 + 4 * s;
      fprintf(stderr, " | N%d K%d%s: loads+barrier %u stage-in %u K-loop+epilogue %u", p.st[s].N, p.st[s].K, p.st[s].ln_w ? " LN" : "", t[1] - t[0],
              t[2] - t[1], t[3] - t[2]);
    }
    fprintf(stderr, " | total %u (entry -> first stage %u)\n", h[4 * p.n_stages] - h[0], h[1] - h[0]);
    if (q.trace_stage >= 0 && q.trace_stage < p.n_stages) {   // fine stamps: deltas from "input staged" of that stage
      fprintf(stderr, "[chain fine] stage %d (N%d K%d):", q.trace_stage, p.st[q.trace_stage].N, p.st[q.trace_stage].K);
      unsigned prev = h[1 + 4 * q.trace_stage + 2];
      for (int i = 64; i < 128 && h[i]; ++i) { fprintf(stderr, " %u", h[i] - prev); prev = h[i]; }
      fprintf(stderr, "\n");
    }
    return 0;
  }
  hipLaunchKernelGGL(chain_kernel<false>, grid, dim3(512), p.lds_bytes, st, p);
  EC_LAUNCH_CHECK();
  return 0;
}

}  // namespace ec
