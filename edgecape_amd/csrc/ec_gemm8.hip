// bf16 NT GEMM for the backbone of the EdgeCape hot path on gfx950 (MI355X): the north-star kernel.
//
//   C[M,N] = epilogue(A[M,K] @ B[N,K]^T)      A = activations (bf16, K contiguous), B = nn.Linear weight (bf16)
//
// Reference ops: B4 `qkv` Linear (the roofline kernel), B6 `proj`, B7 `fc1`/`fc2` of every DINOv2 block
// (facebookresearch/dinov2 Attention / Mlp, called from EdgeCape/models/detectors/EdgeCape.py:188-189).
//
// Structure (MI355X-first; cdna_hip_programming.md §5 "256² 8-phase"):
//   * 256x256 output tile, K step 64, one 512-thread workgroup per CU (8 waves = 2 (M) x 4 (N), 128x64 per wave,
//     v_mfma_f32_16x16x32_bf16, 128 fp32 accumulator registers per lane), persistent over an XCD-contiguous tile range.
//   * operands go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip) as HALF-TILES of 128 rows x 128 B;
//     a K-tile is four half-tiles (A-h0, B-h0, B-h1, A-h1), LDS holds two K-tiles (128 KiB) + one 16 KiB dummy slot.
//     The LDS image is [16 rows][64 B] sub-tiles, 32-byte XOR-swizzled for rows 8..15 (conflict-free ds_read_b128); since
//     LDS-DMA writes lane-linear, the swizzle is applied to the per-lane SOURCE address and undone by the readers.
//   * the K loop is a sequence of PHASES, one accumulator quadrant (64x32, 16 MFMAs) each:
//         R: ds_read the fragments this phase needs, issue ONE half-tile of the load stream, s_waitcnt vmcnt(6)
//         barrier;  M: 16 MFMAs;  barrier
//     The two wave groups (wr = 0 / 1; they share each SIMD pairwise) run ONE BARRIER apart, so while one group is in its
//     MFMA phase the other does its LDS reads and DMA issue: matrix pipe beside memory pipe on every SIMD.
//   * the load stream runs 5 half-tiles ahead of the compute stream and crosses output-tile boundaries (the next tile's
//     first K-tile lands under this tile's epilogue); vmcnt is never drained inside the K loop: after issuing stream index
//     q+5 in phase q, vmcnt(6) leaves the three newest half-tiles in flight and retires everything phase q+1 reads.
//     RAW: a half-tile is read one phase after the wait that retires it, with a barrier in between for both groups.
//     WAR: a slot is re-staged >= 3 phases after its last ds_read.
//   * 16x16x32 MFMAs, not 32x32x16: 32x32 variants of this kernel measured 8-9 % slower (1245-1265 vs 1360-1375 TFLOP/s at
//     8192^3, with one quadrant per phase and with two) although the 32x32 form has the higher isolated rate.  Phase
//     timestamps (EC_G8_TRACE, tools/g8_trace.py): a 16-MFMA block issues in ~320 cycles (20 per MFMA = the 16x16 form's
//     own rate), barrier-to-barrier ~450; two 32-MFMA phases per K-tile (EC_G8_2PH=1) gain 2 % at 8192^3, nothing at K = 768.
//   * MFMA roles are swapped (A-operand <- weight rows n, B-operand <- activation rows m) so an accumulator lane holds one
//     output row and 4 consecutive columns: the epilogue (bias / pos-table / GELU / LayerScale / residual / bf16 pack)
//     works on 16-byte row segments.
#include <stdlib.h>

#include "ec_common.h"

namespace ec {
namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16v4;

constexpr int G8_HALF = 16384;            // half-tile: 128 rows x 128 B
constexpr int G8_KT = 4 * G8_HALF;        // K-tile: A-h0 | B-h0 | B-h1 | A-h1
constexpr int G8_STAGE = 2 * G8_KT;       // 8 x 4 KiB: per-wave output staging; also the dummy target of the stream's tail
constexpr int G8_LDS = 2 * G8_KT + 8 * 4096;   // 160 KiB, the whole CU
constexpr int G8_AHEAD = 5;               // half-tiles the load stream runs ahead

#define G8_SB() __builtin_amdgcn_sched_barrier(0)
#define G8_BAR()                      \
  do {                                \
    G8_SB();                          \
    __builtin_amdgcn_s_barrier();     \
    G8_SB();                          \
  } while (0)

// Epilogue kinds (compile-time): the three shapes of the bf16 backbone blocks + a generic one (every GemmP option).
enum { G8_GENERIC = 0, G8_BIAS_BF16 = 1, G8_SCALE_BF16 = 2, G8_GELU_BF16 = 3 };

// GELU for 16-bit outputs: x * Phi(x) with Phi(x) = 1 / (1 + 2^(x * P(x^2))), P an even minimax polynomial fitted to the erf
// form (nn.GELU default, dinov2 Mlp) on |x| <= 9.  bf16 outputs: degree 2 in x^2, max |err| 2.5e-5 (far below the bf16 rounding
// of the result), 6 plain VALU ops + exp2 + rcp per element (the epilogue of fc1 is VALU-bound: 128 GELUs per lane per tile).
// fp16 outputs carry three more significand bits: degree 4, max |err| 3.0e-6 (two more FMAs).
template <bool F16>
__device__ __forceinline__ float gelu_fast8(float x) {
  const float s = fminf(x * x, 81.f);
  float q;
  if constexpr (F16) {
    q = fmaf(-3.229071e-06f, s, 8.82395e-05f);
    q = fmaf(q, s, 3.6026796e-04f);
    q = fmaf(q, s, -1.0522668e-01f);
    q = fmaf(q, s, -2.3020453e+00f);
  } else {
    q = fmaf(1.01453915e-03f, s, -1.06777424e-01f);
    q = fmaf(q, s, -2.30111947e+00f);
  }
  const float e = __builtin_amdgcn_exp2f(x * q);
  return x * __builtin_amdgcn_rcpf(1.f + e);
}

// acc[mi][ni] (f32x4) of lane l: row m = m0 + wr*128 + (mi>>2)*64 + (mi&3)*16 + (l&15),
//                                cols n = n0 + wc*64 + (ni>>1)*32 + (ni&1)*16 + (l>>4)*4 .. +3
template <int KIND, bool FULL, bool F16>
__device__ __forceinline__ void g8_epilogue(const GemmP& p, f32x4 (&acc)[8][4], char* smem, int m0, int n0, int wr, int wc, int lane) {
  const int ncol = n0 + wc * 64 + (lane >> 4) * 4;
  const int mrow = m0 + wr * 128 + (lane & 15);
  if constexpr (KIND == G8_GENERIC) {
    f32x4 bias4[4], gam4[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = ncol + (ni >> 1) * 32 + (ni & 1) * 16;
      const bool nok = n < p.N;
      bias4[ni] = (p.bias && nok) ? *(const f32x4*)(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
      gam4[ni] = (p.gamma && nok) ? *(const f32x4*)(p.gamma + n) : f32x4{1.f, 1.f, 1.f, 1.f};
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tile-seam drain (see the kernel), after the bias loads were issued
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
      const int m = mrow + (mi >> 2) * 64 + (mi & 3) * 16;
      if (m >= p.M) continue;
      const float* trow = p.table ? p.table + (long)(m % p.period) * p.ldt : nullptr;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int n = ncol + (ni >> 1) * 32 + (ni & 1) * 16;
        if (n >= p.N) continue;
        f32x4 v = acc[mi][ni] + bias4[ni];
        if (trow) v += *(const f32x4*)(trow + n);
        if (p.act == ACT_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        } else if (p.act == ACT_GELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = p.c_bf16 ? gelu_fast8<F16>(v[e]) : gelu_erf(v[e]);
        }
        v *= gam4[ni];
        if (p.resid) v += *(const f32x4*)(p.resid + (long)m * p.ldr + n);
        if (p.c_bf16) {
          *(u32x2*)((char*)p.C + ((long)m * p.ldc + n) * 2) = pack4_h<F16>(v);
        } else {
          *(f32x4*)((float*)p.C + (long)m * p.ldc + n) = v;
        }
      }
    }
  } else {
    // bf16 output, bias (+ LayerScale gamma | GELU): no runtime branches, 32-bit offsets from a uniform base,
    // immediate column offsets.  (host checks M * ldc * 2 < 2^31)
    f32x4 bias4[4], gam4[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = ncol + (ni >> 1) * 32 + (ni & 1) * 16;
      const bool nok = FULL || n < p.N;
      bias4[ni] = nok ? *(const f32x4*)(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (KIND == G8_SCALE_BF16) gam4[ni] = nok ? *(const f32x4*)(p.gamma + n) : f32x4{1.f, 1.f, 1.f, 1.f};
    }
    // tile-seam drain: one wait covers the bias loads just issued AND every LDS-DMA half-tile still in flight
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // Stores go through a per-wave 4 KiB LDS staging slot: a piece = 32 rows x 64 columns of bf16 (two m-fragments) is
    // written fragment-wise (8 x ds_write_b64, 16-byte chunks XOR-swizzled by row) and read back row-wise
    // (4 x ds_read_b128), so every global store instruction writes 8 full 128-byte lines (16 B per lane) instead of 16
    // quarter lines (8 B per lane): the dwordx2 form is store-issue bound at ~7 B/clk/CU (MI355X_MICROARCH.md).
    char* const Cb = (char*)p.C;
    const unsigned ldc2 = (unsigned)p.ldc * 2u;
    const int wave = wr * 4 + wc;
    char* const stg = smem + G8_STAGE + wave * 4096;
    const int wrow = lane & 15, wq = lane >> 4;                       // writer: fragment row, column quad
    const int rrow = lane >> 3, rch = lane & 7;                       // reader: row within 8, 16-byte chunk
    const unsigned goff0 = (unsigned)(m0 + wr * 128 + rrow) * ldc2 + (unsigned)(n0 + wc * 64) * 2u + (unsigned)rch * 16u;
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) {
#pragma unroll
      for (int mm = 0; mm < 2; ++mm) {
        const int mi = pc * 2 + mm;
        const int row = mm * 16 + wrow;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          f32x4 v = acc[mi][ni] + bias4[ni];
          if constexpr (KIND == G8_GELU_BF16) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_fast8<F16>(v[e]);
          }
          if constexpr (KIND == G8_SCALE_BF16) v *= gam4[ni];
          const u32x2 o = pack4_h<F16>(v);   // 2 x v_cvt_pk_{bf16,f16}_f32 (RNE)
          const int chunk = (ni >> 1) * 4 + (ni & 1) * 2 + (wq >> 1);
          *(u32x2*)(stg + row * 128 + ((chunk ^ (row & 7)) << 4) + (wq & 1) * 8) = o;
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = j * 8 + rrow;
        const u32x4 o = *(const u32x4*)(stg + row * 128 + ((rch ^ (row & 7)) << 4));
        const int mloc = pc * 32 + row;                                // row inside the wave's 128
        const bool ok = FULL || ((m0 + wr * 128 + mloc < p.M) && (n0 + wc * 64 + rch * 8 < p.N));
        if (ok) *(u32x4*)(Cb + goff0 + (unsigned)(pc * 32 + j * 8) * ldc2) = o;
      }
    }
  }
}

// TRACE (debug instantiation only, EC_G8_TRACE=1): lane 0 of every wave of workgroup 0 stamps s_memtime at five points of
// every phase of one K-tile pair into its LDS staging slot; dumped to p.aux after the first tile (tools/g8_trace.py).
// ONEBAR (EC_G8_1BAR=1, A/B only - measured SLOWER: 8192^3 1300 vs 1355 TFLOP/s, QKV 1029 vs 1111-1139): one barrier per phase instead of two.  Group 0 keeps only the barrier AFTER its MFMA block, group 1 only the one BEFORE
// its MFMA block, so physical barrier #p is {group 0 done with M_p, group 1 done with R_p}: inside one barrier interval group 0
// runs R_p then M_p while group 1 runs M_(p-1) then R_p - the same matrix-pipe / memory-pipe alternation on every SIMD with half
// the barrier releases.  RAW: half-tile i is waited for (vmcnt) in every wave's R_(i-2), barrier #(i-2) follows that wait in
// both groups and precedes every read of it (R_(i-1) at the earliest).  WAR: slot of half-tile i-8 is re-staged in R_(i-5);
// its last reads (R_(<=i-8)) were retired before barrier #(i-7) in both groups.
template <int KIND, int TAG, bool F16 = false, bool TRACE = false>
__global__ __launch_bounds__(512) void gemm8_bf16_kernel(GemmP p) {
  constexpr bool TWOPH = false, ONEBAR = false;   // rejected schedules (see the header comment), kept out of the build
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  const int ntm = (p.M + 255) >> 8, ntn = (p.N + 255) >> 8;
  const int ntiles = ntm * ntn;
  const int nk = p.K >> 6;                       // K-tiles per output tile (even: checked on the host)
  const long lda_b = p.lda * 2, ldb_b = p.ldb * 2;

  // XCD-aware persistent schedule: workgroup b runs on XCD b % 8 (observed, speed only); XCD x walks the contiguous tile
  // range [x*chunk, (x+1)*chunk) (row-major, n fastest) so the tiles in flight on one L2 share operand panels.
  const int nxcd = (gridDim.x >= 8 && gridDim.x % 8 == 0) ? 8 : 1;
  const int chunk = (ntiles + nxcd - 1) / nxcd;
  const int xcd = blockIdx.x % nxcd, slot = blockIdx.x / nxcd, nslot = gridDim.x / nxcd;
  const int t_end = min(ntiles, (xcd + 1) * chunk);
  const int t_first = xcd * chunk + slot;
  if (t_first >= t_end) return;                 // whole workgroup leaves: no barrier has been executed yet

  // ---- load stream (LDS-DMA) state -------------------------------------------------------------------------------
  // wave w stages row-group w (16 rows) of every half-tile, both 64-byte K halves (2 wave-instructions of 1 KiB).
  // lane -> row r = lane>>2 of the row-group, physical 16-B chunk lane&3 which holds logical chunk (lane&3) ^ 2*(r>=8).
  const int ld_r = lane >> 2;
  const int ld_c = ((lane & 3) ^ ((lane >> 5) << 1)) << 4;
  const char* rp0; const char* rp1; const char* rp2; const char* rp3;   // A-h0, B-h0, B-h1, A-h1 row pointers (+ chunk)
  int ls_kt = 0, ls_tile = t_first;
  bool ls_live = true;
  auto set_rows = [&](int t) {
    const int m0 = (t / ntn) << 8, n0 = (t % ntn) << 8;
    const int ra = m0 + (wave >> 2) * 128 + (wave & 3) * 16 + ld_r;     // A-h0 row; A-h1 = +64
    const int rb = n0 + (wave >> 1) * 64 + (wave & 1) * 16 + ld_r;      // B-h0 row; B-h1 = +32
    const char* A = (const char*)p.A + ld_c;
    const char* B = (const char*)p.B + ld_c;
    rp0 = A + (long)min(ra, p.M - 1) * lda_b;
    rp3 = A + (long)min(ra + 64, p.M - 1) * lda_b;
    rp1 = B + (long)min(rb, p.N - 1) * ldb_b;
    rp2 = B + (long)min(rb + 32, p.N - 1) * ldb_b;
  };
  auto issue = [&](const char* rp, int half) {
    const char* src = rp + (long)ls_kt * 128;
    // (dead stream: each wave's dummy loads land in its OWN staging slot, which it only uses after draining its own loads)
    char* dst = smem + (ls_live ? ((ls_kt & 1) * G8_KT + half * G8_HALF + wave * 2048) : (G8_STAGE + wave * 4096));
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(src + 64), (lptr_t)(dst + 1024), 16, 0, 0);
  };
  auto advance = [&]() {
    if (++ls_kt == nk) {
      ls_kt = 0;
      ls_tile += nslot;
      if (ls_tile < t_end) set_rows(ls_tile);
      else ls_live = false;                      // stream exhausted: keep issuing (valid addresses) into the dummy slot
    }
  };
  set_rows(t_first);
  issue(rp0, 0); issue(rp1, 1); issue(rp2, 2); issue(rp3, 3);
  advance();
  issue(rp0, 0);
  if constexpr (TWOPH) issue(rp1, 1);                // two-phase schedule: the stream runs 6 half-tiles ahead
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the prologue half-tiles have landed (this wave's part)
  G8_BAR();
  if constexpr (!ONEBAR) {
    if (wr == 1) G8_BAR();                           // stagger: group 1 runs one barrier behind group 0
  }
#define G8_BAR_A() do { if constexpr (ONEBAR) { if (wr == 1) G8_BAR(); } else G8_BAR(); } while (0)   /* before the MFMA block */
#define G8_BAR_B() do { if constexpr (ONEBAR) { if (wr == 0) G8_BAR(); } else G8_BAR(); } while (0)   /* after the MFMA block */

  // ---- fragment read addresses ------------------------------------------------------------------------------------
  // reader lane: row r = lane&15 of the 16-row sub-tile, logical chunk lane>>4 at physical chunk (lane>>4) ^ 2*(r>=8)
  const int rd_off = ((lane & 15) << 6) + ((((lane >> 4) ^ (((lane & 15) >> 3) << 1))) << 4);
  const char* a_base = smem + rd_off + wr * 8192;                // row-groups 4*wr.. of the A halves
  const char* b_base = smem + rd_off + wc * 4096 + G8_HALF;      // row-groups 2*wc.. of the B halves (B-h0 is slot 1)

#define G8_STAMP(slot)                                                                                   \
  do {                                                                                                    \
    if constexpr (TRACE) {                                                                                \
      if (trace_on) {                                                                                     \
        const unsigned ts_ = (unsigned)__builtin_amdgcn_s_memtime();                                      \
        if (lane == 0) ((unsigned*)(smem + G8_STAGE + wave * 4096))[slot] = ts_;                          \
      }                                                                                                   \
    }                                                                                                     \
  } while (0)
  for (int t = t_first; t < t_end; t += nslot) {
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 af[4][2], b0[2][2], b1[2][2];

    for (int kt2 = 0; kt2 < nk; kt2 += 2) {
      const bool trace_on = TRACE && blockIdx.x == 0 && t == t_first && kt2 == 4;
#pragma unroll
      for (int buf = 0; buf < 2; ++buf) {
        const char* ab = a_base + buf * G8_KT;
        const char* bb = b_base + buf * G8_KT;
        G8_STAMP(buf * 20 + 0);
        if constexpr (TWOPH) {
          // ---- two phases per K-tile (32 MFMAs each: half as many barrier transitions per MFMA).  Stream 6 half-tiles ahead:
          //      phase A issues B-h1/A-h1 of K-tile +1 and waits vmcnt(8) (A-h1 of this K-tile has landed);
          //      phase B issues A-h0/B-h0 of K-tile +2 and waits vmcnt(6) (A-h0, B-h0, B-h1 of K-tile +1 have landed).
#pragma unroll
          for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
              b0[f][kh] = *(const bf16x8*)(bb + f * 2048 + kh * 1024);
              b1[f][kh] = *(const bf16x8*)(bb + G8_HALF + f * 2048 + kh * 1024);
            }
#pragma unroll
          for (int fi = 0; fi < 4; ++fi)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) af[fi][kh] = *(const bf16x8*)(ab + fi * 2048 + kh * 1024);
          issue(rp2, 2);
          issue(rp3, 3);
          if (buf == 1 || kt2 > 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // first K-tile after a seam: already drained
          G8_BAR();
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int fi = 0; fi < 4; ++fi)
#pragma unroll
              for (int f = 0; f < 2; ++f)
                acc[fi][f] = mfma16x16x32_h<F16>(b0[f][kh], af[fi][kh], acc[fi][f]);
#pragma unroll
          for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int fi = 0; fi < 4; ++fi)
#pragma unroll
              for (int f = 0; f < 2; ++f)
                acc[fi][2 + f] = mfma16x16x32_h<F16>(b1[f][kh], af[fi][kh], acc[fi][2 + f]);
          __builtin_amdgcn_s_setprio(0);
          G8_BAR();
#pragma unroll
          for (int fi = 0; fi < 4; ++fi)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) af[fi][kh] = *(const bf16x8*)(ab + 3 * G8_HALF + fi * 2048 + kh * 1024);
          advance();
          issue(rp0, 0);
          issue(rp1, 1);
          asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
          G8_BAR();
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int fi = 0; fi < 4; ++fi)
#pragma unroll
              for (int f = 0; f < 2; ++f)
                acc[4 + fi][2 + f] = mfma16x16x32_h<F16>(b1[f][kh], af[fi][kh], acc[4 + fi][2 + f]);
#pragma unroll
          for (int kh = 0; kh < 2; ++kh)
#pragma unroll
            for (int fi = 0; fi < 4; ++fi)
#pragma unroll
              for (int f = 0; f < 2; ++f)
                acc[4 + fi][f] = mfma16x16x32_h<F16>(b0[f][kh], af[fi][kh], acc[4 + fi][f]);
          __builtin_amdgcn_s_setprio(0);
        } else {
        // ---------------- phase 0: quadrant (m-half 0, n-half 0)
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int kh = 0; kh < 2; ++kh) b0[f][kh] = *(const bf16x8*)(bb + f * 2048 + kh * 1024);
#pragma unroll
        for (int fi = 0; fi < 4; ++fi)
#pragma unroll
          for (int kh = 0; kh < 2; ++kh) af[fi][kh] = *(const bf16x8*)(ab + fi * 2048 + kh * 1024);
        G8_STAMP(40 + buf * 8 + 0);
        issue(rp1, 1);
        G8_STAMP(40 + buf * 8 + 1);
        if (buf == 1 || kt2 > 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // see "tile seam" below
        G8_STAMP(buf * 20 + 1);
        G8_BAR_A();
        G8_STAMP(buf * 20 + 2);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
          for (int fi = 0; fi < 4; ++fi)
#pragma unroll
            for (int f = 0; f < 2; ++f)
              acc[fi][f] = mfma16x16x32_h<F16>(b0[f][kh], af[fi][kh], acc[fi][f]);
        __builtin_amdgcn_s_setprio(0);
        G8_STAMP(buf * 20 + 3);
        G8_BAR_B();
        G8_STAMP(buf * 20 + 4);
        G8_STAMP(buf * 20 + 5);
        // ---------------- phase 1: quadrant (0, 1)
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int kh = 0; kh < 2; ++kh) b1[f][kh] = *(const bf16x8*)(bb + G8_HALF + f * 2048 + kh * 1024);
        G8_STAMP(40 + buf * 8 + 2);
        issue(rp2, 2);
        G8_STAMP(40 + buf * 8 + 3);
        if (buf == 1 || kt2 > 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // see "tile seam" below
        G8_STAMP(buf * 20 + 6);
        G8_BAR_A();
        G8_STAMP(buf * 20 + 7);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
          for (int fi = 0; fi < 4; ++fi)
#pragma unroll
            for (int f = 0; f < 2; ++f)
              acc[fi][2 + f] = mfma16x16x32_h<F16>(b1[f][kh], af[fi][kh], acc[fi][2 + f]);
        __builtin_amdgcn_s_setprio(0);
        G8_STAMP(buf * 20 + 8);
        G8_BAR_B();
        G8_STAMP(buf * 20 + 9);
        G8_STAMP(buf * 20 + 10);
        // ---------------- phase 2: quadrant (1, 1)
#pragma unroll
        for (int fi = 0; fi < 4; ++fi)
#pragma unroll
          for (int kh = 0; kh < 2; ++kh) af[fi][kh] = *(const bf16x8*)(ab + 3 * G8_HALF + fi * 2048 + kh * 1024);
        G8_STAMP(40 + buf * 8 + 4);
        issue(rp3, 3);
        G8_STAMP(40 + buf * 8 + 5);
        if (buf == 1 || kt2 > 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // see "tile seam" below
        G8_STAMP(buf * 20 + 11);
        G8_BAR_A();
        G8_STAMP(buf * 20 + 12);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
          for (int fi = 0; fi < 4; ++fi)
#pragma unroll
            for (int f = 0; f < 2; ++f)
              acc[4 + fi][2 + f] = mfma16x16x32_h<F16>(b1[f][kh], af[fi][kh], acc[4 + fi][2 + f]);
        __builtin_amdgcn_s_setprio(0);
        G8_STAMP(buf * 20 + 13);
        G8_BAR_B();
        G8_STAMP(buf * 20 + 14);
        G8_STAMP(buf * 20 + 15);
        // ---------------- phase 3: quadrant (1, 0); the load stream moves on to the next K-tile
        G8_STAMP(40 + buf * 8 + 6);
        advance();
        issue(rp0, 0);
        G8_STAMP(40 + buf * 8 + 7);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        G8_STAMP(buf * 20 + 16);
        G8_BAR_A();
        G8_STAMP(buf * 20 + 17);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
          for (int fi = 0; fi < 4; ++fi)
#pragma unroll
            for (int f = 0; f < 2; ++f)
              acc[4 + fi][f] = mfma16x16x32_h<F16>(b0[f][kh], af[fi][kh], acc[4 + fi][f]);
        __builtin_amdgcn_s_setprio(0);
        }
        G8_STAMP(buf * 20 + 18);
        if constexpr (ONEBAR) {
          G8_BAR_B();
        } else {
          if (buf == 0 || kt2 + 2 < nk) G8_BAR();
        }
        G8_STAMP(buf * 20 + 19);   // the tile's last barrier is placed around the epilogue below
      }
    }

    // ---- epilogue.  Both groups run it concurrently: group 0 passes the tile's last barrier first, group 1 after.
    // Tile seam: every half-tile issued so far (stream indices up to 4 of the NEXT tile) is drained at the top of the
    // epilogue (g8_epilogue, right after its bias loads are issued), before any store is issued, so the first three phases of the next tile need no wait; from its phase 3 on, vmcnt(6) covers
    // loads issued after this point (loads retire in order among themselves; the epilogue's stores, also counted by
    // vmcnt, can only make that wait stricter) while the stores drain in the background under the next tile's MFMAs.
    if constexpr (TRACE) {
      if (blockIdx.x == 0 && t == t_first && lane < 56)
        ((unsigned*)p.aux)[wave * 64 + lane] = ((const unsigned*)(smem + G8_STAGE + wave * 4096))[lane];
    }
    if constexpr (!ONEBAR) {
      if (wr == 0) G8_BAR();
    }
    {
      const int m0 = (t / ntn) << 8, n0 = (t % ntn) << 8;
      if (KIND != G8_GENERIC && m0 + 256 <= p.M && n0 + 256 <= p.N) g8_epilogue<KIND, true, F16>(p, acc, smem, m0, n0, wr, wc, lane);
      else g8_epilogue<KIND, false, F16>(p, acc, smem, m0, n0, wr, wc, lane);
    }
    if constexpr (!ONEBAR) {
      if (wr == 1) G8_BAR();
    }
  }
  if constexpr (!ONEBAR) {
    if (wr == 0) G8_BAR();   // balances group 1's extra barrier at the start
  }
}


}  // namespace

// Per-device launch state: the 160 KiB dynamic-LDS attribute is a per-device function attribute and the CU count differs per
// device, so a process that drives several GPUs (one engine per device) gets both for every device it touches.
namespace {
struct G8Dev { bool attr_done = false; bool trace_attr = false; int ncu = 0; };
G8Dev g8_dev[64];
}  // namespace

// Returns 1 if this kernel handled the problem, 0 if the shape is not eligible (caller falls back), <0 on error.
int gemm8_bf16(const GemmP& p, hipStream_t st) {
  static const int disable = getenv("EC_GEMM8_OFF") ? atoi(getenv("EC_GEMM8_OFF")) : 0;
  if (disable) return 0;
  if (!p.ab_bf16 || p.batch != 1 || p.act == ACT_TANHGATE) return 0;
  if (p.K % 128 != 0 || p.N % 16 != 0 || p.M < 1024 || p.N < 256) return 0;
  typedef void (*kern_t)(GemmP);
  // epilogue kind from the options; TAG only names the symbol for rocprof (1 qkv, 2 proj, 3 fc1, 4 fc2)
  int kind = G8_GENERIC;
  if (p.c_bf16 && p.bias && !p.resid && !p.table && (long)p.M * p.ldc * 2 < (1l << 31)) {
    if (p.act == ACT_NONE && !p.gamma) kind = G8_BIAS_BF16;
    else if (p.act == ACT_NONE && p.gamma) kind = G8_SCALE_BF16;
    else if (p.act == ACT_GELU && !p.gamma) kind = G8_GELU_BF16;
  }
#define G8_ROW(F) \
      {gemm8_bf16_kernel<0, 0, F>, gemm8_bf16_kernel<0, 1, F>, gemm8_bf16_kernel<0, 2, F>, gemm8_bf16_kernel<0, 3, F>, gemm8_bf16_kernel<0, 4, F>}, \
      {gemm8_bf16_kernel<1, 0, F>, gemm8_bf16_kernel<1, 1, F>, gemm8_bf16_kernel<1, 0, F>, gemm8_bf16_kernel<1, 0, F>, gemm8_bf16_kernel<1, 0, F>}, \
      {gemm8_bf16_kernel<2, 0, F>, gemm8_bf16_kernel<2, 0, F>, gemm8_bf16_kernel<2, 2, F>, gemm8_bf16_kernel<2, 0, F>, gemm8_bf16_kernel<2, 4, F>}, \
      {gemm8_bf16_kernel<3, 0, F>, gemm8_bf16_kernel<3, 0, F>, gemm8_bf16_kernel<3, 0, F>, gemm8_bf16_kernel<3, 3, F>, gemm8_bf16_kernel<3, 0, F>}
  static const kern_t table[2][4][5] = {{G8_ROW(false)}, {G8_ROW(true)}};
#undef G8_ROW
  int dev = 0;
  EC_HIP(hipGetDevice(&dev));
  EC_REQUIRE(dev >= 0 && dev < 64, -1, "gemm8: device ordinal out of range");
  G8Dev& ds = g8_dev[dev];
  if (!ds.attr_done) {
    for (int f = 0; f < 2; ++f)
      for (int k = 0; k < 4; ++k)
        for (int t = 0; t < 5; ++t)
          EC_HIP(hipFuncSetAttribute((const void*)table[f][k][t], hipFuncAttributeMaxDynamicSharedMemorySize, G8_LDS));
    EC_HIP(hipDeviceGetAttribute(&ds.ncu, hipDeviceAttributeMultiprocessorCount, dev));
    ds.attr_done = true;
  }
  const long ntiles = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
  long grid = ds.ncu;
  if (ntiles < grid) grid = ntiles;
  static const bool trace = getenv("EC_G8_TRACE") != nullptr;
  if (trace && kind == G8_BIAS_BF16 && !p.h_f16 && p.aux) {   // debug: p.aux = device buffer of 8 x 64 uint32 timestamps
    if (!ds.trace_attr) {
      EC_HIP(hipFuncSetAttribute((const void*)gemm8_bf16_kernel<1, 0, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, G8_LDS));
      ds.trace_attr = true;
    }
    hipLaunchKernelGGL((gemm8_bf16_kernel<1, 0, false, true>), dim3((unsigned)grid), dim3(512), G8_LDS, st, p);
    EC_LAUNCH_CHECK();
    return 1;
  }
  hipLaunchKernelGGL(table[p.h_f16 ? 1 : 0][kind][p.tag], dim3((unsigned)grid), dim3(512), G8_LDS, st, p);
  EC_LAUNCH_CHECK();
  return 1;
}

}  // namespace ec
