// GEMM kernels for gfx950 (MI355X).
//
// gemm_nt:  C[M,N] = epilogue(A[M,K] @ B[N,K]^T)  — every Linear / 1x1-conv / patch-embed of the
// hot path (reference ops: SURVEY.md §2.3 rows B1,B4,B6,B7,H1,H4,H6,H7,H10-H14).  Both operands are
// K-contiguous ("NT"), which is torch's nn.Linear weight layout, so no transposes are needed.
//
//   * operands fp32  -> v_mfma_f32_32x32x2_f32   (exact fp32 products, fp32 accumulate: parity mode)
//   * operands bf16  -> v_mfma_f32_32x32x16_bf16 (fp32 accumulate: throughput mode)
//
// Tiling (MI355X-first, 64-wide waves): 128x128 output tile per 256-thread workgroup (4 waves, each a
// 64x64 sub-tile = 2x2 MFMA 32x32 accumulators = 64 acc VGPRs), K step = 128 BYTES of K per row for
// either dtype (32 fp32 / 64 bf16), so the LDS image and all address arithmetic are dtype-independent.
// Global->LDS staging is the gfx950 direct path (global_load_lds_dwordx4: 64 lanes x 16 B = 8 rows x
// 128 B per wave-instruction, no VGPR round trip), double-buffered.  The LDS destination is lane-linear,
// so the bank-conflict swizzle is applied to the SOURCE address and undone on the ds_read_b128 side
// (both use phys_chunk = chunk ^ ((row >> 1) & 7), an involution; conflict-free for the 16-lane groups
// ds_read_b128 is serviced in).
#include <stdlib.h>

#include "ec_common.h"

namespace ec {

namespace {

constexpr int KBYTES = 128;   // bytes of K per row per stage (32 fp32 / 64 bf16)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// Stage ROWS rows x 128 B of a K-contiguous operand into LDS with NW waves (ROWS/8 wave-instructions).
template <int ROWS, int NW>
__device__ inline void stage_rows(const char* __restrict__ base, long ld_bytes, int row0, int nrows_total, long kbyte0,
                                  char* lds_tile, int wave, int lane) {
  constexpr int PER = ROWS / 8 / NW;
  static_assert(ROWS % (8 * NW) == 0, "tile rows must split evenly over the waves");
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int rb = wave * PER + j;           // 8-row block inside the tile
    const int r = rb * 8 + (lane >> 3);      // row inside the tile
    const int c = (lane & 7) ^ ((r >> 1) & 7);   // logical 16-B chunk this lane fetches (swizzle on the SOURCE side)
    int gr = row0 + r;
    gr = gr < nrows_total ? gr : nrows_total - 1;   // clamp: rows past the edge are computed but never stored
    const char* src = base + (long)gr * ld_bytes + kbyte0 + c * 16;
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds_tile + rb * 1024), 16, 0, 0);   // + lane*16 by hardware
  }
}

// erf via Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7): used for GELU when the result is rounded to bf16 anyway.
__device__ inline float gelu_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float e = 1.f - poly * t * __builtin_amdgcn_exp2f(-z * z * 1.44269504088896340736f);
  const float erfv = x < 0.f ? -e : e;
  return 0.5f * x * (1.f + erfv);
}

// Persistent NT GEMM.  Workgroup = WGM x WGN waves; tile BM x BN; each wave owns (BM/WGM) x (BN/WGN).
// The MFMA roles are SWAPPED (A-operand <- weight rows n, B-operand <- activation rows m) so that an
// accumulator lane holds ONE output row m and 4 consecutive columns n per register quad: the epilogue
// then writes 16-byte row segments (after a v_permlane32_swap pairing of the two half-waves) instead of
// scalar elements.  TAG only separates kernel symbols for rocprof (1 qkv, 2 proj, 3 fc1, 4 fc2).
// MODE 0: fp32 operands, exact (v_mfma_f32_32x32x2_f32).  MODE 1: bf16 operands.  MODE 2 ("bf16x3"): A is fp32 in memory and
// is split in registers into hi + lo bf16 (hi = RNE(a), lo = RNE(a - hi), a - hi exact), B is the weight matrix PRE-SPLIT on
// the host into the same 128-byte K-blocks ([32 hi | 32 lo] bf16 per 32 k) so staging and addressing are those of fp32;
// acc += hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 (fp32 accumulate): ~2^-17 relative operand error at 16/3 of the
// fp32 MFMA rate — the head's throughput mode (the head is 6 % of the FLOPs; keeping it fp32-class keeps the 1e-3 gate).
// MODE 4 ("fp16x1", the single-pass form of the head's mixed precision): the same data as MODE 2 - A fp32 in memory, B in 128-byte
// K-blocks of 32 k - but ONE v_mfma_f32_32x32x16_f16 per product: A is rounded to fp16 in registers, the first 64 bytes of a B
// block hold fp16(W) (split_pack_weights_h1; the second half is unused).  Used where oracle/head_precision_study.py shows the
// 1e-3 gate has room for it (skeleton head, decoder): a third of the MFMA work of bf16x3.
enum { GM_F32 = 0, GM_BF16 = 1, GM_SPLIT = 2, GM_F16 = 3, GM_SPLIT1 = 4 };   // GM_F16: as GM_BF16 with IEEE fp16 operands / output

__device__ __forceinline__ void split8(const f32x4 x0, const f32x4 x1, bf16x8& hi, bf16x8& lo) {
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16v2;
  unsigned h[4], l[4];
  const float f[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    h[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{f[2 * q], f[2 * q + 1]}, bf16v2));
    const float r0 = f[2 * q] - __uint_as_float(h[q] << 16);
    const float r1 = f[2 * q + 1] - __uint_as_float(h[q] & 0xffff0000u);
    l[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r0, r1}, bf16v2));
  }
  hi = __builtin_bit_cast(bf16x8, u32x4{h[0], h[1], h[2], h[3]});
  lo = __builtin_bit_cast(bf16x8, u32x4{l[0], l[1], l[2], l[3]});
}

// NS = depth of the LDS stage ring.  The load stream (LDS-DMA) runs NS-1 stages ahead of the compute position and crosses
// tile boundaries; each K-step waits with a COUNTED vmcnt for its own stage only, so NS-2 stages stay in flight across
// the barrier.  The head's GEMMs are short (K = 256 = 8 stages) and small: with NS = 2 every K-step pays a full load
// latency; with NS = 4 the latencies overlap (cdna_hip_programming.md §5 "Pipelining across barriers").
template <int MODE, int BM, int BN, int WGM, int WGN, int NS, int TAG>
__global__ __launch_bounds__(WGM* WGN * 64) void gemm_nt_kernel(GemmP p) {
  constexpr bool BF16 = MODE == GM_BF16 || MODE == GM_F16;   // 16-bit operands
  constexpr bool F16 = MODE == GM_F16;
  constexpr int LPS = BM / 8 / (WGM * WGN) + BN / 8 / (WGM * WGN);   // LDS-DMA instructions per thread per stage
  constexpr int NW = WGM * WGN;
  constexpr int MT = BM / WGM / 32, NT = BN / WGN / 32;
  constexpr int A_BYTES = BM * KBYTES, B_BYTES = BN * KBYTES, STAGE = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int esz = BF16 ? 2 : 4;
  const int lrow = lane & 31, hi = lane >> 5;

  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
  const int per_batch = ntm * ntn;
  const int ntiles = per_batch * p.batch;
  const long lda_b = p.lda * esz, ldb_b = p.ldb * esz;
  const int nk = (p.K * esz) / KBYTES;

  // XCD-aware persistent schedule: workgroup b runs on XCD b % 8 (observed); XCD x walks the contiguous tile
  // range [x*chunk, (x+1)*chunk) (row-major, n fastest) so tiles in flight on one L2 share operand panels.
  const int nxcd = (gridDim.x >= 8 && gridDim.x % 8 == 0) ? 8 : 1;
  const int chunk = (ntiles + nxcd - 1) / nxcd;
  const int xcd = blockIdx.x % nxcd, slot = blockIdx.x / nxcd, nslot = gridDim.x / nxcd;
  const int t_end = min(ntiles, (xcd + 1) * chunk);

  auto tile_coords = [&](int t, int& bz, int& m0, int& n0) {
    bz = t / per_batch;
    const int r = t - bz * per_batch;
    m0 = (r / ntn) * BM;
    n0 = (r % ntn) * BN;
  };
  auto stage = [&](int bz, int m0, int n0, int kt, char* buf) {
    const char* A = (const char*)p.A + (long)bz * p.sA * esz;
    const char* B = (const char*)p.B + (long)bz * p.sB * esz;
    // (GemmP::kwrap, 16-bit modes: the K-concatenated bf16x3 product over two-plane operands - A planes hi | lo | hi, B planes hi | hi | lo)
    const int kta = (BF16 && p.kwrap && kt >= 2 * p.kwrap) ? kt - 2 * p.kwrap : kt;
    const int ktb = (BF16 && p.kwrap && kt >= p.kwrap) ? kt - p.kwrap : kt;
    stage_rows<BM, NW>(A, lda_b, m0, p.M, (long)kta * KBYTES, buf, wave, lane);
    stage_rows<BN, NW>(B, ldb_b, n0, p.N, (long)ktb * KBYTES, buf + A_BYTES, wave, lane);
  };

  int t = xcd * chunk + slot;
  if (t >= t_end) return;
  int bz, m0, n0;
  tile_coords(t, bz, m0, n0);
  // ---- load stream: (ls_t, ls_kt) = next stage to issue, into ring slot ls_slot.  Past the last stage of the workgroup's
  // last tile the stream stops advancing and re-issues that stage into its own slot (identical bytes), which keeps the
  // per-step load count - and therefore the counted waits - uniform without a dummy buffer.
  int ls_t = t, ls_kt = 0, ls_slot = 0, lbz = bz, lm0 = m0, ln0 = n0;
  auto issue_next = [&]() {
    stage(lbz, lm0, ln0, ls_kt, smem + ls_slot * STAGE);
    const int nslot_ = (ls_slot + 1 == NS) ? 0 : ls_slot + 1;
    if (ls_kt + 1 < nk) {
      ++ls_kt; ls_slot = nslot_;
    } else if (ls_t + nslot < t_end) {
      ls_t += nslot; ls_kt = 0; ls_slot = nslot_;
      tile_coords(ls_t, lbz, lm0, ln0);
    }
  };
#pragma unroll
  for (int i = 0; i < NS - 1; ++i) issue_next();
  int cur = 0;
  for (; t < t_end; t += nslot) {
    f32x16 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    tile_coords(t, bz, m0, n0);

    for (int kt = 0; kt < nk; ++kt) {
      // this wave's part of the stage about to be computed has landed; NS-2 younger stages stay in flight
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * LPS) : "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();      // raw barrier: __syncthreads() would drain the LDS-DMA queue (vmcnt(0))
      __builtin_amdgcn_sched_barrier(0);
      issue_next();                      // into the slot computed in the previous step (all waves are past it)
      const char* At = smem + cur * STAGE;
      const char* Bt = At + A_BYTES;
      if constexpr (MODE == GM_SPLIT) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {   // two 16-k MFMA steps per 32-k stage
          bf16x8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
          for (int j = 0; j < MT; ++j) {
            const int ra = wm * (BM / WGM) + j * 32 + lrow;
            const int sw = (ra >> 1) & 7, c0 = ks * 4 + hi * 2;
            const f32x4 x0 = *(const f32x4*)(At + ra * KBYTES + ((c0 ^ sw) << 4));
            const f32x4 x1 = *(const f32x4*)(At + ra * KBYTES + (((c0 + 1) ^ sw) << 4));
            split8(x0, x1, ah[j], al[j]);
          }
#pragma unroll
          for (int i = 0; i < NT; ++i) {
            const int rb = wn * (BN / WGN) + i * 32 + lrow;
            const int sw = (rb >> 1) & 7, c0 = ks * 2 + hi;
            bh[i] = *(const bf16x8*)(Bt + rb * KBYTES + ((c0 ^ sw) << 4));
            bl[i] = *(const bf16x8*)(Bt + rb * KBYTES + (((4 + c0) ^ sw) << 4));
          }
#pragma unroll
          for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[i], ah[j], acc[i][j], 0, 0, 0);   // small terms first
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[i], al[j], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[i], ah[j], acc[i][j], 0, 0, 0);
            }
        }
      } else if constexpr (MODE == GM_SPLIT1) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {   // two 16-k MFMA steps per 32-k stage, one MFMA per product
          bf16x8 ah[MT], bh[NT];
#pragma unroll
          for (int j = 0; j < MT; ++j) {
            const int ra = wm * (BM / WGM) + j * 32 + lrow;
            const int sw = (ra >> 1) & 7, c0 = ks * 4 + hi * 2;
            const f32x4 x0 = *(const f32x4*)(At + ra * KBYTES + ((c0 ^ sw) << 4));
            const f32x4 x1 = *(const f32x4*)(At + ra * KBYTES + (((c0 + 1) ^ sw) << 4));
            typedef __attribute__((ext_vector_type(8))) float f32x8_;
            const f32x8_ x = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
            ah[j] = __builtin_bit_cast(bf16x8, __builtin_convertvector(x, f16x8));   // 4 x v_cvt_pk_f16_f32 (RNE)
          }
#pragma unroll
          for (int i = 0; i < NT; ++i) {
            const int rb = wn * (BN / WGN) + i * 32 + lrow;
            const int sw = (rb >> 1) & 7, c0 = ks * 2 + hi;
            bh[i] = *(const bf16x8*)(Bt + rb * KBYTES + ((c0 ^ sw) << 4));
          }
#pragma unroll
          for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) acc[i][j] = mfma32x32x16_h<true>(bh[i], ah[j], acc[i][j]);
        }
      } else
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int chunkk = 2 * kk + hi;
        f32x4 xa[MT], wb[NT];
#pragma unroll
        for (int j = 0; j < MT; ++j) {
          const int ra = wm * (BM / WGM) + j * 32 + lrow;
          xa[j] = *(const f32x4*)(At + ra * KBYTES + ((chunkk ^ ((ra >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          const int rb = wn * (BN / WGN) + i * 32 + lrow;
          wb[i] = *(const f32x4*)(Bt + rb * KBYTES + ((chunkk ^ ((rb >> 1) & 7)) << 4));
        }
        if constexpr (BF16) {
#pragma unroll
          for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j)
              acc[i][j] = mfma32x32x16_h<F16>(__builtin_bit_cast(bf16x8, wb[i]), __builtin_bit_cast(bf16x8, xa[j]), acc[i][j]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
              for (int j = 0; j < MT; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[i][e], xa[j][e], acc[i][j], 0, 0, 0);
        }
      }
      cur = (cur + 1 == NS) ? 0 : cur + 1;
    }

    // ---- epilogue.  acc[i][j][r], lane (m_local = lane & 31, hi): column n_local = (r&3) + 8*(r>>2) + 4*hi.
    // Pair the register quads g = 2q, 2q+1 across the half-waves (v_permlane32_swap): afterwards lane hi=0 holds
    // columns 16q..16q+7 and lane hi=1 columns 16q+8..16q+15 of its row -> 8 consecutive outputs per lane.
    const float* bias = p.bias ? p.bias + (long)bz * p.sBias : nullptr;
    const float* resid = p.resid ? p.resid + (long)bz * p.sR : nullptr;
    const float* aux = p.aux ? p.aux + (long)bz * p.sAux : nullptr;
    char* Cb = (char*)p.C + (long)bz * p.sC * (p.c_bf16 ? 2 : 4);
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int m = m0 + wm * (BM / WGM) + j * 32 + lrow;
      const bool mok = m < p.M;
      const int mt = mok && p.table ? m % p.period : 0;
#pragma unroll
      for (int i = 0; i < NT; ++i) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float a = acc[i][j][8 * q + e], b = acc[i][j][8 * q + 4 + e];
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
            v[e] = __uint_as_float(sw[0]);
            v[4 + e] = __uint_as_float(sw[1]);
          }
          const int n = n0 + wn * (BN / WGN) + i * 32 + 16 * q + 8 * hi;   // first of this lane's 8 columns
          if (!mok || n >= p.N) continue;                                   // N is a multiple of 8 (checked on the host)
          if (bias) {
            const f32x4 b0 = *(const f32x4*)(bias + n), b1 = *(const f32x4*)(bias + n + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
          }
          if (p.table) {
            const float* tp = p.table + (long)mt * p.ldt + n;
            const f32x4 t0 = *(const f32x4*)tp, t1 = *(const f32x4*)(tp + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += t0[e]; v[4 + e] += t1[e]; }
          }
          if (p.act == ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
          } else if (p.act == ACT_GELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = p.c_bf16 ? gelu_fast(v[e]) : p.c_x3 ? gelu_fast8<true>(v[e]) : gelu_erf(v[e]);   // (c_x3: the form of the 8-phase epilogues, ec_common.h)
          } else if (p.act == ACT_TANHGATE) {
            const float* ap = aux + (long)m * p.ldaux + n;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (tanhf(v[e]) + 1.f) * ap[e];
          }
          if (p.gamma) {
            const f32x4 g0 = *(const f32x4*)(p.gamma + n), g1 = *(const f32x4*)(p.gamma + n + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] *= g0[e]; v[4 + e] *= g1[e]; }
          }
          if (resid) {
            const float* rp = resid + (long)m * p.ldr + n;
            const f32x4 r0 = *(const f32x4*)rp, r1 = *(const f32x4*)(rp + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
          }
          if (p.c_x3) {   // bf16 split [hi | lo], planes N apart (GemmP::c_x3; the small-M fallback of the bf16x3 backbone's fc1)
            u32x2_t h0, l0, h1, l1;
            split4_h<F16>(f32x4{v[0], v[1], v[2], v[3]}, h0, l0);
            split4_h<F16>(f32x4{v[4], v[5], v[6], v[7]}, h1, l1);
            char* cp = Cb + ((long)m * p.ldc + n) * 2;
            *(u32x4*)cp = u32x4{h0[0], h0[1], h1[0], h1[1]};
            *(u32x4*)(cp + (long)p.N * 2) = u32x4{l0[0], l0[1], l1[0], l1[1]};
          } else if (p.c_bf16) {
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e)   // (16-bit output of the single-pass fp16 mode: IEEE fp16, like its operands)
              o[e] = (unsigned)f2h<F16 || MODE == GM_SPLIT1>(v[2 * e]) | ((unsigned)f2h<F16 || MODE == GM_SPLIT1>(v[2 * e + 1]) << 16);
            *(u32x4*)(Cb + ((long)m * p.ldc + n) * 2) = o;
          } else {
            f32x4 o0, o1;
#pragma unroll
            for (int e = 0; e < 4; ++e) { o0[e] = v[e]; o1[e] = v[4 + e]; }
            float* cp = (float*)Cb + (long)m * p.ldc + n;
            *(f32x4*)cp = o0;
            *(f32x4*)(cp + 4) = o1;
          }
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // duplicate tail stages of the load stream must land before the LDS is released
}

// ------------------------------------------------------------------------------------------------
// Small batched fp32 GEMM on the vector ALU: 64x64 tile, 256 threads, 4x4 outputs per thread.
// Used for the per-sample 100x100 / 100xHW contractions of the head (support-keypoint pooling,
// cosine-similarity adjacency, Markov powers, GCN aggregation, similarity map) where M = K = 100.
// ------------------------------------------------------------------------------------------------
constexpr int SB = 64, SK = 16;

__global__ __launch_bounds__(256) void bgemm_small_kernel(BgemmP p) {
  __shared__ float As[SK][SB + 4];
  __shared__ float Bs[SK][SB + 4];
  const int tid = threadIdx.x;
  const int bz = blockIdx.z;
  const int m0 = blockIdx.y * SB, n0 = blockIdx.x * SB;
  const float* A = p.A + (long)(p.modA > 0 ? bz % p.modA : bz) * p.sA;
  const float* B = p.B + (long)(p.modB > 0 ? bz % p.modB : bz) * p.sB;
  float* C = p.C + (long)bz * p.sC;
  const int tx = tid & 15, ty = tid >> 4;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

  for (int k0 = 0; k0 < p.K; k0 += SK) {
    // A tile: 64 rows x 16 k -> As[k][m]
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = tid + it * 256;
      const int r = idx >> 4, k = idx & 15;
      const int gm = m0 + r, gk = k0 + k;
      As[k][r] = (gm < p.M && gk < p.K) ? A[(long)gm * p.lda + gk] : 0.f;
    }
    if (p.transB) {  // B [N,K]
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int idx = tid + it * 256;
        const int r = idx >> 4, k = idx & 15;
        const int gn = n0 + r, gk = k0 + k;
        Bs[k][r] = (gn < p.N && gk < p.K) ? B[(long)gn * p.ldb + gk] : 0.f;
      }
    } else {  // B [K,N]
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int idx = tid + it * 256;
        const int k = idx >> 6, c = idx & 63;
        const int gn = n0 + c, gk = k0 + k;
        Bs[k][c] = (gn < p.N && gk < p.K) ? B[(long)gk * p.ldb + gn] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SK; ++k) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[k][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bs[k][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx * 4 + j;
      if (n >= p.N) continue;
      float v = p.alpha * acc[i][j];
      if (p.beta != 0.f) v += p.beta * C[(long)m * p.ldc + n];
      if (p.self) {
        const float rs = p.rowscale[(long)(p.mod_rs > 0 ? bz % p.mod_rs : bz) * p.M + m];
        v += rs * p.self[(long)bz * p.s_self + (long)m * p.ld_self + n];
      }
      if (p.relu) v = fmaxf(v, 0.f);
      C[(long)m * p.ldc + n] = v;
    }
  }
}


// ------------------------------------------------------------------------------------------------
// The same small batched GEMMs on the matrix pipe, exact fp32 (v_mfma_f32_32x32x2_f32): one wave per 32x32 output tile,
// operands straight from global memory (they are a few hundred KB per sample and L2-resident; no reuse inside a wave worth
// an LDS round trip).  The contraction index is permuted inside blocks of 8 (MFMA step s, k-half h <-> k = k0 + 4 h + s)
// so a lane's four A values are one 16-byte load; B is four coalesced row loads (NN) or one 16-byte load (NT).
// ------------------------------------------------------------------------------------------------
template <bool TRANSB>
__global__ __launch_bounds__(256) void bgemm_mfma_kernel(BgemmP p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bz = blockIdx.z;
  const int m0 = blockIdx.y * 32, n0 = (blockIdx.x * 4 + wave) * 32;
  if (n0 >= p.N) return;
  const float* A = p.A + (long)(p.modA > 0 ? bz % p.modA : bz) * p.sA;
  const float* B = p.B + (long)(p.modB > 0 ? bz % p.modB : bz) * p.sB;
  float* C = p.C + (long)bz * p.sC;
  const int i = lane & 31, h = lane >> 5;
  const int am = min(m0 + i, p.M - 1);          // clamp: rows / columns past the edge are computed, never stored
  const int bn = min(n0 + i, p.N - 1);
  const float* arow = A + (long)am * p.lda + 4 * h;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int kfull = p.K & ~7;
  auto load_b = [&](int k0) {
    f32x4 b;
    if constexpr (TRANSB) {
      b = *(const f32x4*)(B + (long)bn * p.ldb + k0 + 4 * h);
    } else {
      const float* bp = B + (long)(k0 + 4 * h) * p.ldb + bn;
#pragma unroll
      for (int s = 0; s < 4; ++s) b[s] = bp[(long)s * p.ldb];
    }
    return b;
  };
  // software pipeline: the operands of block k0 + 8 are in flight while block k0 is multiplied (without the scheduling
  // fence hipcc sinks every load next to its MFMA: five dependent memory round trips per block).  Deeper staging (32 k per
  // stage) measured slower: the kernel is bound by vector-memory instruction issue (5 per 4 MFMAs), not by latency.
  f32x4 a_nx = kfull > 0 ? *(const f32x4*)arow : f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 b_nx = kfull > 0 ? load_b(0) : f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < kfull; k0 += 8) {
    const f32x4 a = a_nx, b = b_nx;
    if (k0 + 8 < kfull) {
      a_nx = *(const f32x4*)(arow + k0 + 8);
      b_nx = load_b(k0 + 8);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (kfull < p.K) {   // ragged tail (K = 100, 324): same permutation, out-of-range k contribute zeros
    f32x4 a, b;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int k = kfull + 4 * h + s;
      const bool ok = k < p.K;
      a[s] = ok ? A[(long)am * p.lda + k] : 0.f;
      b[s] = ok ? (TRANSB ? B[(long)bn * p.ldb + k] : B[(long)k * p.ldb + bn]) : 0.f;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
  }
  // D[row][col]: lane holds column n0 + i, rows m0 + (r&3) + 8 (r>>2) + 4 h
  const int n = n0 + i;
  if (n >= p.N) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (m >= p.M) continue;
    float v = p.alpha * acc[r];
    if (p.beta != 0.f) v += p.beta * C[(long)m * p.ldc + n];
    if (p.self) {
      const float rs = p.rowscale[(long)(p.mod_rs > 0 ? bz % p.mod_rs : bz) * p.M + m];
      v += rs * p.self[(long)bz * p.s_self + (long)m * p.ld_self + n];
    }
    if (p.relu) v = fmaxf(v, 0.f);
    C[(long)m * p.ldc + n] = v;
  }
}

}  // namespace

namespace {
struct Cfg { int bm, bn, threads, lds, per_cu; };
template <int MODE, int BM, int BN, int WGM, int WGN, int NS>
int launch_cfg(const GemmP& p, hipStream_t st, int per_cu) {
  typedef void (*kern_t)(GemmP);
  constexpr bool TAGGED = MODE != GM_SPLIT && MODE != GM_SPLIT1 && MODE != GM_F16 && BM == 256;   // per-tag symbols only where rocprof needs to tell the backbone GEMMs apart
  static const kern_t table[5] = {gemm_nt_kernel<MODE, BM, BN, WGM, WGN, NS, 0>, gemm_nt_kernel<MODE, BM, BN, WGM, WGN, NS, TAGGED ? 1 : 0>,
                                  gemm_nt_kernel<MODE, BM, BN, WGM, WGN, NS, TAGGED ? 2 : 0>, gemm_nt_kernel<MODE, BM, BN, WGM, WGN, NS, TAGGED ? 3 : 0>,
                                  gemm_nt_kernel<MODE, BM, BN, WGM, WGN, NS, TAGGED ? 4 : 0>};
  constexpr int LDS = NS * (BM + BN) * KBYTES;
  // per-device: the dynamic-LDS limit is a per-device function attribute (a process may drive several GPUs)
  static bool attr_done[64] = {};
  static int ncu_dev[64] = {};
  int dev = 0;
  EC_HIP(hipGetDevice(&dev));
  EC_REQUIRE(dev >= 0 && dev < 64, -1, "gemm_nt: device ordinal out of range");
  if (!attr_done[dev]) {
    for (int t = 0; t < 5; ++t)
      EC_HIP(hipFuncSetAttribute((const void*)table[t], hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    EC_HIP(hipDeviceGetAttribute(&ncu_dev[dev], hipDeviceAttributeMultiprocessorCount, dev));
    attr_done[dev] = true;
  }
  const int ncu = ncu_dev[dev];
  const long ntiles = (long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * p.batch;
  long grid = (long)ncu * per_cu;
  if (ntiles < grid) grid = ntiles;
  hipLaunchKernelGGL(table[p.tag], dim3((unsigned)grid), dim3(WGM * WGN * 64), LDS, st, p);
  EC_LAUNCH_CHECK();
  return 0;
}
}  // namespace

int gemm_nt(const GemmP& p, hipStream_t st) {
  const int esz = p.ab_bf16 ? 2 : 4;
  EC_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0 && p.batch > 0, -1, "gemm_nt: empty problem");
  EC_REQUIRE((p.K * esz) % KBYTES == 0, -1, "gemm_nt: K must be a multiple of 128 bytes");
  EC_REQUIRE((p.lda * esz) % 16 == 0 && (p.ldb * esz) % 16 == 0, -1, "gemm_nt: row strides must be 16-byte multiples");
  EC_REQUIRE(((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.B % 16) == 0, -1, "gemm_nt: operands must be 16-byte aligned");
  EC_REQUIRE(p.N % 8 == 0 && p.ldc % 8 == 0 && ((uintptr_t)p.C % 16) == 0, -1, "gemm_nt: N and ldc must be multiples of 8");
  EC_REQUIRE(!p.resid || (p.ldr % 4 == 0), -1, "gemm_nt: residual stride must be a multiple of 4");
  EC_REQUIRE(!p.table || (p.ldt % 4 == 0), -1, "gemm_nt: table stride must be a multiple of 4");
  EC_REQUIRE(p.act != ACT_TANHGATE || p.aux, -1, "gemm_nt: tanh-gate epilogue needs aux");
  EC_REQUIRE(p.tag >= 0 && p.tag < 5, -1, "gemm_nt: bad tag");
  EC_REQUIRE(!(p.split && p.ab_bf16), -1, "gemm_nt: split (bf16x3) mode takes fp32 A and a pre-split B");
  EC_REQUIRE(!p.c_x3 || (p.ab_bf16 && !p.c_bf16 && p.batch == 1 && p.ldc >= 2l * p.N), -1,
             "gemm_nt: split output (c_x3) takes 16-bit operands, one batch and ldc >= 2 N");
  EC_REQUIRE(!p.kwrap || (p.ab_bf16 && p.K == 3 * 64 * p.kwrap && p.lda >= 128l * p.kwrap && p.ldb >= 128l * p.kwrap), -1,
             "gemm_nt: kwrap takes 16-bit two-plane operands of 64 * kwrap elements per plane and K = 3 planes");
  if (p.ab_bf16) {   // large 16-bit problems: the 8-phase 256x256x64 kernel (block GEMMs of the backbone)
    const int rc = gemm8_bf16(p, st);
    if (rc != 0) return rc < 0 ? rc : 0;
  }
  EC_REQUIRE(!p.x2 && !p.c_x2, -1, "gemm_nt: fp16x2 operands exist on the 8-phase kernel only (K_layer % 128 == 0, N >= 256, N % 16 == 0)");
  // Tile choice.  The fp32 MFMA rate (64 FLOP/clk/SIMD = 614 GFLOP/s per CU) is reached by any of these tiles, so what
  // matters for the head's small problems (M = bs*K = 3200 rows, N = 256) is how evenly the tiles spread over the 256
  // CUs: pick the tile minimising  ceil(tiles / CUs) * tile_area / efficiency.  256x256 only pays for the big backbone
  // GEMMs (fp32 parity mode) where operand re-reads dominate.
#ifdef EC_G8_LAB   // kernel-lab build only (python -m edgecape_amd.build --lab): force a tile configuration, EC_GEMM_TILE=256/256128/128/12864/64
  static const int force = getenv("EC_GEMM_TILE") ? atoi(getenv("EC_GEMM_TILE")) : 0;
#else
  constexpr int force = 0;
#endif
  // (a larger stage ring measured neutral on the head's shapes - kp 3200x256x256: 8.0 vs 8.2 us; 10368x256x768: 30 vs 37 us: those
  // kernels sit at the per-launch floor, not at the load latency - and was removed in round 3: the 2-stage ring is the only form)
  static int ncu_dev[64] = {};
  int dev = 0;
  EC_HIP(hipGetDevice(&dev));
  EC_REQUIRE(dev >= 0 && dev < 64, -1, "gemm_nt: device ordinal out of range");
  if (!ncu_dev[dev]) EC_HIP(hipDeviceGetAttribute(&ncu_dev[dev], hipDeviceAttributeMultiprocessorCount, dev));
  const int ncu = ncu_dev[dev];
  struct Opt { int bm, bn; float eff; };
  static const Opt opts[5] = {{256, 256, 1.00f}, {256, 128, 1.00f}, {128, 128, 0.95f}, {128, 64, 0.85f}, {64, 64, 0.75f}};
  int sel = 2;
  float best = 1e30f;
  for (int i = 0; i < 5; ++i) {
    const long nt = (long)((p.M + opts[i].bm - 1) / opts[i].bm) * ((p.N + opts[i].bn - 1) / opts[i].bn) * p.batch;
    const float cost = (float)((nt + ncu - 1) / ncu) * (float)(opts[i].bm * opts[i].bn) / opts[i].eff;
    if (cost < best) { best = cost; sel = i; }
  }
  if (force == 256) sel = 0; else if (force == 256128) sel = 1; else if (force == 128) sel = 2; else if (force == 12864) sel = 3;
  else if (force == 64) sel = 4;
  if (p.ab_bf16 && p.h_f16) {   // fp16 operands: the small-shape fallbacks of the fp16 backbone mode (2-stage ring only)
    switch (sel) {
      case 0: return launch_cfg<GM_F16, 256, 256, 2, 4, 2>(p, st, 1);
      case 1: return launch_cfg<GM_F16, 256, 128, 4, 2, 2>(p, st, 1);
      case 2: return launch_cfg<GM_F16, 128, 128, 2, 2, 2>(p, st, 2);
      case 3: return launch_cfg<GM_F16, 128, 64, 2, 2, 2>(p, st, 3);
      default: return launch_cfg<GM_F16, 64, 64, 2, 2, 2>(p, st, 4);
    }
  }
  if (p.ab_bf16) {
    switch (sel) {
      case 0: return launch_cfg<GM_BF16, 256, 256, 2, 4, 2>(p, st, 1);
      case 1: return launch_cfg<GM_BF16, 256, 128, 4, 2, 2>(p, st, 1);
      case 2: return launch_cfg<GM_BF16, 128, 128, 2, 2, 2>(p, st, 2);
      case 3: return launch_cfg<GM_BF16, 128, 64, 2, 2, 2>(p, st, 3);
      default: return launch_cfg<GM_BF16, 64, 64, 2, 2, 2>(p, st, 4);
    }
  }
  if (p.split == 2) {   // fp16x1: the single-pass form of the head's mixed precision
    switch (sel) {
      case 0: return launch_cfg<GM_SPLIT1, 256, 256, 2, 4, 2>(p, st, 1);
      case 1: return launch_cfg<GM_SPLIT1, 256, 128, 4, 2, 2>(p, st, 1);
      case 2: return launch_cfg<GM_SPLIT1, 128, 128, 2, 2, 2>(p, st, 2);
      case 3: return launch_cfg<GM_SPLIT1, 128, 64, 2, 2, 2>(p, st, 3);
      default: return launch_cfg<GM_SPLIT1, 64, 64, 2, 2, 2>(p, st, 4);
    }
  }
  if (p.split) {
    // experiment kept for A/B: wide-N wave tiles (32 x 128 per wave) share the in-register hi/lo split of an A fragment
    // (~30 VALU) between 4 x 3 MFMAs instead of 1-2 x 3 (PMC at 128x64: 16.7 VALU per MFMA, MFMA pipe 19 % busy, waves one third
    // parked / one third issue-stalled / one third issuing).  Measured 0-15 % SLOWER than the 2x2 configurations on every head
    // shape (13568x768x256: 35.5 / 33.7 vs 30.9 us) - the split is not what bounds these kernels.
    if (force == 64256) return launch_cfg<GM_SPLIT, 64, 256, 2, 2, 2>(p, st, 2);
    if (force == 1281) return launch_cfg<GM_SPLIT, 128, 128, 4, 1, 2>(p, st, 2);
    switch (sel) {
      case 0: return launch_cfg<GM_SPLIT, 256, 256, 2, 4, 2>(p, st, 1);
      case 1: return launch_cfg<GM_SPLIT, 256, 128, 4, 2, 2>(p, st, 1);
      case 2: return launch_cfg<GM_SPLIT, 128, 128, 2, 2, 2>(p, st, 2);
      case 3: return launch_cfg<GM_SPLIT, 128, 64, 2, 2, 2>(p, st, 3);
      default: return launch_cfg<GM_SPLIT, 64, 64, 2, 2, 2>(p, st, 4);
    }
  }
  switch (sel) {
    case 0: return launch_cfg<GM_F32, 256, 256, 2, 4, 2>(p, st, 1);
    case 1: return launch_cfg<GM_F32, 256, 128, 4, 2, 2>(p, st, 1);
    case 2: return launch_cfg<GM_F32, 128, 128, 2, 2, 2>(p, st, 2);
    case 3: return launch_cfg<GM_F32, 128, 64, 2, 2, 2>(p, st, 3);
    default: return launch_cfg<GM_F32, 64, 64, 2, 2, 2>(p, st, 4);
  }
}

#ifdef EC_G8_LAB
// Lab build only: time one fixed configuration of gemm_nt_kernel (bf16) - cfg 0: 256x256, 8 waves (2x4); cfg 1: 256x256, 4 waves (2x2,
// 128x128 per wave, one wave per SIMD)
extern "C" int ec_lab_gemm_nt(const void* A, const void* W, const float* bias, void* C, int M, int N, int K, int cfg, int iters,
                              void* stream, float* ms) {
  GemmP p;
  p.A = A; p.B = W; p.C = C; p.bias = bias; p.M = M; p.N = N; p.K = K; p.lda = K; p.ldb = K; p.ldc = N; p.ab_bf16 = 1; p.c_bf16 = 1;
  hipStream_t st = (hipStream_t)stream;
  hipEvent_t e0, e1;
  EC_HIP(hipEventCreate(&e0));
  EC_HIP(hipEventCreate(&e1));
  for (int i = 0; i <= iters; ++i) {
    if (i == 1) EC_HIP(hipEventRecord(e0, st));
    int rc = cfg == 1 ? launch_cfg<GM_BF16, 256, 256, 2, 2, 2>(p, st, 1) : launch_cfg<GM_BF16, 256, 256, 2, 4, 2>(p, st, 1);
    if (rc) return rc;
  }
  EC_HIP(hipEventRecord(e1, st));
  EC_HIP(hipEventSynchronize(e1));
  float t = 0.f;
  EC_HIP(hipEventElapsedTime(&t, e0, e1));
  *ms = t / (float)iters;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return 0;
}
#endif

void split_pack_weights(const float* W, long n_rows, long K, float* out) {
  for (long r = 0; r < n_rows; ++r)
    for (long kb = 0; kb < K / 32; ++kb) {
      const float* src = W + r * K + kb * 32;
      bf16_t* dst = (bf16_t*)(out + r * K + kb * 32);
      for (int j = 0; j < 32; ++j) {
        const bf16_t h = f2bf(src[j]);
        dst[j] = h;
        dst[32 + j] = f2bf(src[j] - bf2f(h));
      }
    }
}

void split_pack_weights_h1(const float* W, long n_rows, long K, float* out) {
  for (long r = 0; r < n_rows; ++r)
    for (long kb = 0; kb < K / 32; ++kb) {
      const float* src = W + r * K + kb * 32;
      bf16_t* dst = (bf16_t*)(out + r * K + kb * 32);
      for (int j = 0; j < 32; ++j) {
        dst[j] = f2half_host(src[j]);
        dst[32 + j] = 0;
      }
    }
}

int bgemm_small(const BgemmP& p, hipStream_t st) {
  EC_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0 && p.batch > 0, -1, "bgemm_small: empty problem");
  const bool aligned = p.lda % 4 == 0 && p.sA % 4 == 0 && ((uintptr_t)p.A % 16) == 0 &&
                       (!p.transB || (p.ldb % 4 == 0 && p.sB % 4 == 0 && ((uintptr_t)p.B % 16) == 0));
  if (aligned) {   // 16-byte operand loads need 4-float aligned rows (K = 100: yes; the demos' odd K: VALU kernel below)
    dim3 g((p.N + 127) / 128, (p.M + 31) / 32, p.batch);
    if (p.transB) hipLaunchKernelGGL(bgemm_mfma_kernel<true>, g, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(bgemm_mfma_kernel<false>, g, dim3(256), 0, st, p);
    EC_LAUNCH_CHECK();
    return 0;
  }
  dim3 grid((p.N + SB - 1) / SB, (p.M + SB - 1) / SB, p.batch);
  hipLaunchKernelGGL(bgemm_small_kernel, grid, dim3(256), 0, st, p);
  EC_LAUNCH_CHECK();
  return 0;
}

}  // namespace ec
