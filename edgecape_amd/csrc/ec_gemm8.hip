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
//   * operands go HBM -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, no VGPR round trip) as HALF-TILES of 128 rows x 128 B;
//     a K-tile is four half-tiles (A-h0, B-h0, B-h1, A-h1), LDS holds two K-tiles (128 KiB), the per-wave output staging
//     slots (16 KiB) and the bias / LayerScale slices of two tiles (8 KiB).
//     One wave-instruction of the stream (a PIECE, 1 KiB) covers 8 rows x 128 B = eight WHOLE cache lines: lanes 0-31 the first
//     64 bytes of the rows, lanes 32-63 the second (round 2; tools/dma_probe.hip: the stream of the QKV problem alone runs in 29 us
//     with whole-line pieces and in 42 us with the 16 rows x 64 B pieces of round 1, which fetched every line in two halves).
//     LDS image of a 16-row group (2 KiB): [8-row octet][k half][8 rows][64 B]; the 16-byte chunk index of rows 8..15 is XORed with 2
//     (conflict-free ds_read_b128 of a 16-row x 32-k fragment); since LDS-DMA writes lane-linear, the swizzle is applied to the
//     per-lane SOURCE address and undone by the readers.
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
//     timestamps (round 1): a 16-MFMA block issues in ~320 cycles (20 per MFMA = the 16x16 form's own rate), barrier-to-barrier
//     ~450; two 32-MFMA phases per K-tile gain 2 % at 8192^3, nothing at K = 768.
//   * MFMA roles are swapped (A-operand <- weight rows n, B-operand <- activation rows m) so an accumulator lane holds one
//     output row and 4 consecutive columns: the epilogue (bias / pos-table / GELU / LayerScale / residual / bf16 pack)
//     works on 16-byte row segments.
#include <stdlib.h>

#include <type_traits>

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
constexpr int G8_STAGE = 2 * G8_KT;       // 8 x 2 KiB: per-wave output staging (one 16-row x 64-column piece at a time)
constexpr int G8_BIAS = G8_STAGE + 8 * 2048;    // 2 tile parities x 8 waves x 256 B: this wave's 64 bias values (LDS-DMA, one tile ahead)
constexpr int G8_GAMMA = G8_BIAS + 2 * 8 * 256;  // same for the LayerScale vector
constexpr int G8_SCHED = G8_GAMMA + 2 * 8 * 256;  // two ints: the dynamic schedule's hand-over slots (tile parity)
constexpr int G8_LDS = G8_SCHED + 64;            // 152 KiB + 64 B
constexpr int G8_TRACE = G8_LDS;                 // lab build, LAB & 512: 2 wave groups x 32 s_memtime stamps (64-bit) behind the product's LDS
constexpr int G8_LDS_LAB = G8_TRACE + 512;
constexpr int G8_AHEAD = 5;               // half-tiles the load stream runs ahead

#define G8_SB() __builtin_amdgcn_sched_barrier(0)
#define G8_BAR()                      \
  do {                                \
    G8_SB();                          \
    __builtin_amdgcn_s_barrier();     \
    G8_SB();                          \
  } while (0)

// Epilogue kinds (compile-time): the three shapes of the bf16 backbone blocks + a generic one (every GemmP option).
// G8_TAB_*: C = acc + bias[n] + table[m % period][n] (the patch embedding's positional table; the head's image K|V / image-query
// projections, whose positional half is folded into such a table), fp32 or 16-bit output, through the staged whole-line epilogue.
// G8_F32 / G8_RES_F32 / G8_GELU_X3 (round 5): the epilogues of the K-CONCATENATED bf16x3 backbone (ec_model.hip run_backbone: A = bf16
// [hi | lo | hi] planes, B = [W_hi | W_hi | W_lo], K = 3 x the layer's depth), through the same staged whole-line pieces:
//   G8_F32      C = acc + bias, fp32 (QKV)
//   G8_RES_F32  C = (acc + bias) * gamma + C, fp32, in place (proj / fc2: LayerScale + the fp32 residual stream; resid == C)
//   G8_GELU_X3  C = bf16 split planes [hi | lo] of gelu(acc + bias), N elements apart (fc1: the A operand of fc2)
// Their operands are TWO-plane (GemmP::kwrap): the load stream's K position wraps per operand - A walks its planes hi | lo | hi, B its
// planes hi | hi | lo - so no plane is written, stored or fetched from HBM twice (scalar arithmetic in the issue slot only; compiled
// into these kinds, the table-fp32 kind of the patch embedding and the generic kind).
// (the generic kind did this first: 12 spilled VGPRs, fragment-wise quarter-line stores - QKV 227 us, fc1 343 us at cfg2)
// G8_GELU_X2 (round 6): fc1 of the fp16x2 backbone (ec_common.h split4_x2): C = [fp16 plane | e5m2 lo8 plane | e5m2 hi8 plane] of gelu(acc + bias).
// The fp16x2 operands themselves are a template flag of the kernel (X2), not a kind: G8_F32 / G8_RES_F32 / G8_GELU_X2 are its epilogues.
enum { G8_GENERIC = 0, G8_BIAS_BF16 = 1, G8_SCALE_BF16 = 2, G8_GELU_BF16 = 3, G8_TAB_H16 = 4, G8_TAB_F32 = 5, G8_F32 = 6, G8_RES_F32 = 7, G8_GELU_X3 = 8, G8_GELU_X2 = 9 };
constexpr bool g8_f32_out(int kind) { return kind == G8_TAB_F32 || kind == G8_F32 || kind == G8_RES_F32; }

// (gelu_fast8, the single-transcendental GELU of the 16-bit and split-plane epilogues: ec_common.h)

// acc[mi][ni] (f32x4) of lane l: row m = m0 + wr*128 + mi*16 + (l&15),
//                                cols n = n0 + wc*64 + (ni>>1)*32 + (ni&1)*16 + (l>>4)*4 .. +3
// GENERIC epilogue (fp32 output, residual, positional table, any activation): the patch embedding and the op-level tests.  It
// runs at the end of the tile behind a full drain of the load stream (its bias / table / residual loads retire in order behind
// the LDS-DMA half-tiles) - the seam the pipelined epilogue below removes for the backbone's block GEMMs.
template <bool F16>
__device__ __forceinline__ void g8_epilogue_generic(const GemmP& p, f32x4 (&acc)[8][4], int m0, int n0, int wr, int wc, int lane) {
  const int ncol = n0 + wc * 64 + (lane >> 4) * 4;
  const int mrow = m0 + wr * 128 + (lane & 15);
  f32x4 bias4[4], gam4[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int n = ncol + (ni >> 1) * 32 + (ni & 1) * 16;
    const bool nok = n < p.N;
    bias4[ni] = (p.bias && nok) ? *(const f32x4*)(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    gam4[ni] = (p.gamma && nok) ? *(const f32x4*)(p.gamma + n) : f32x4{1.f, 1.f, 1.f, 1.f};
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tile-seam drain, after the bias loads were issued
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) {
    const int m = mrow + mi * 16;
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
        for (int e = 0; e < 4; ++e) v[e] = p.c_bf16 ? gelu_fast8<F16>(v[e]) : p.c_x3 ? gelu_fast8<true>(v[e]) : gelu_erf(v[e]);   // (c_x3: ONE form in every epilogue of the mode)
      }
      v *= gam4[ni];
      if (p.resid) v += *(const f32x4*)(p.resid + (long)m * p.ldr + n);
      if (p.c_x3) {   // bf16 split [hi | lo], planes N apart: the A operand of the next K-concatenated GEMM (fc1 -> fc2, bf16x3 backbone)
        u32x2_t vh, vl;
        split4_h<F16, true>(v, vh, vl);   // (F16: inside the epilogue's FP16_OVFL window)
        bf16_t* c = (bf16_t*)p.C + (long)m * p.ldc + n;
        *(u32x2_t*)c = vh;
        *(u32x2_t*)(c + p.N) = vl;
      } else if (p.c_bf16) {
        *(u32x2*)((char*)p.C + ((long)m * p.ldc + n) * 2) = pack4_h_ovfl<F16>(v);
      } else {
        *(f32x4*)((float*)p.C + (long)m * p.ldc + n) = v;
      }
    }
  }
}

// One PIECE of the pipelined epilogue (16-bit output kinds): the m-fragment `a` = acc[mi][0..3] of this lane's wave, i.e. 16
// rows x 64 columns.  bias (+ GELU | LayerScale) from this wave's LDS slices, 16-bit pack (RNE), transposition through the wave's
// 2 KiB staging slot (fragment-wise ds_write_b64, 16-byte chunks XOR-swizzled by row; row-wise ds_read_b128), then TWO global
// stores of 16 B per lane = 8 full 128-byte lines each (fragment-wise 8-byte stores would be quarter lines and twice the
// instructions: global stores issue at ~70 cycles per wave-instruction per CU whatever their width).  The accumulators are
// zeroed on the way out: the next tile accumulates into them.
//   rsC: buffer descriptor of C with num_records = M * ldc bytes (rows past M are dropped by its range check); goff: byte offset in C of
//   (row m0 + wr*128 + mi*16 + (lane>>3), column n0 + wc*64 + (lane&7)*8); col_ok: this lane's 8 columns are inside N.
//   trow (G8_TAB_* kinds): this lane's row of the table, at the wave's first column (table + (m % period) * ldt + n0 + wc*64); ncols: how
//   many of the wave's 64 columns lie inside N (a multiple of 16: a lane's four columns are all inside or all outside).
//   Bias kinds (G8_BIAS_BF16 / G8_SCALE_BF16 / G8_GELU_BF16, round 4): the accumulators START from the bias - they are initialised with the
//   tile's bias slice instead of zero (the MFMA chain adds the products on top), so a piece neither adds the bias nor zeroes its
//   registers: it re-loads them with the NEXT tile's slice `bias_next` (staged a whole tile ahead) by four ds_read_b128 - per value one
//   packed add and one move fewer on the VALU, which is what bounds the epilogue.
template <int KIND, bool F16, int LAB>
__device__ __forceinline__ void g8_piece(f32x4 (&a)[4], const __amdgpu_buffer_rsrc_t rsC, unsigned goff, unsigned ldc2, bool col_ok, char* stg,
                                         const char* bias_lds, const char* gam_lds, int lane, const float* trow = nullptr, int ncols = 64,
                                         const char* bias_next = nullptr, unsigned plane = 0, unsigned goff8 = 0, bool col_ok8 = false) {
  const int wrow = lane & 15, wq = lane >> 4;                       // writer: fragment row, column quad
  const int rrow = lane >> 3, rch = lane & 7;                       // reader: row within 8, 16-byte chunk
  if constexpr (KIND == G8_F32 || KIND == G8_RES_F32) {
    // fp32 output as in G8_TAB_F32 (two halves of 16 rows x 32 columns through the 2 KiB slot, whole-line stores).  G8_RES_F32: the
    // residual is C itself - it is loaded in the STORE layout (the same 16 bytes per lane that the lane writes afterwards: whole lines,
    // no address arithmetic of its own, in-place safe) at the top of the piece and added behind the transposition
    u32x4 r[2][2];
    if constexpr (KIND == G8_RES_F32) {
#pragma unroll
      for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          r[half][j] = __builtin_amdgcn_raw_buffer_load_b128(rsC, half * 32 + rch * 4 < ncols ? goff + (unsigned)(j * 8) * ldc2 + (unsigned)(half * 128) : 0xFFFFFFF0u, 0, 0);
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int ni = half * 2 + q;
        const int c0 = half * 32 + q * 16 + wq * 4;
        f32x4 v = a[ni] + *(const f32x4*)(bias_lds + c0 * 4);
        if constexpr (KIND == G8_RES_F32) v *= *(const f32x4*)(gam_lds + c0 * 4);
        *(f32x4*)(stg + wrow * 128 + (((q * 4 + wq) ^ (wrow & 7)) << 4)) = v;
        a[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = j * 8 + rrow;
        f32x4 o = *(const f32x4*)(stg + row * 128 + ((rch ^ (row & 7)) << 4));
        if constexpr (KIND == G8_RES_F32) o += __builtin_bit_cast(f32x4, r[half][j]);
        const bool ok = half * 32 + rch * 4 < ncols;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rsC, ok ? goff + (unsigned)(j * 8) * ldc2 + (unsigned)(half * 128) : 0xFFFFFFF0u, 0, 0);
      }
    }
    return;
  }
  if constexpr (KIND == G8_GELU_X3) {
    // fc1 of the K-concatenated bf16x3 backbone: gelu(acc + bias) (single-transcendental form, 6.4e-7), split into bf16 hi + lo, the hi
    // plane staged and stored, then the lo plane: 4 whole-line stores per piece
    u32x2_t vh[4], vl[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int c0 = (ni >> 1) * 32 + (ni & 1) * 16 + wq * 4;
      f32x4 v = a[ni] + *(const f32x4*)(bias_lds + c0 * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = gelu_fast8<true>(v[e]);
      split4_h<F16, true>(v, vh[ni], vl[ni]);
      a[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int chunk = (ni >> 1) * 4 + (ni & 1) * 2 + (wq >> 1);
        *(u32x2_t*)(stg + wrow * 128 + ((chunk ^ (wrow & 7)) << 4) + (wq & 1) * 8) = pl ? vl[ni] : vh[ni];
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = j * 8 + rrow;
        const u32x4 o = *(const u32x4*)(stg + row * 128 + ((rch ^ (row & 7)) << 4));
        const unsigned off = goff + (unsigned)(j * 8) * ldc2;
        __builtin_amdgcn_raw_buffer_store_b128(o, rsC, col_ok ? off + (pl ? plane : 0u) : 0xFFFFFFF0u, 0, 0);
      }
    }
    return;
  }
  if constexpr (KIND == G8_GELU_X2) {
    // fc1 of the fp16x2 backbone: gelu(acc + bias) (single-transcendental form, 6.4e-7), split4_x2.  The fp16 plane leaves as in
    // G8_GELU_X3 (two whole-line stores); then BOTH FP8 planes go through the same 2 KiB slot as [16 rows][64 B lo8 | 64 B hi8] and
    // leave in two stores of 8 rows x (64 + 64) B: four stores per piece, as the bf16x3 kind.  plane = N bytes (one FP8 plane of a row),
    // goff8 = byte offset in C of (row r0 + (lane >> 3), lo8 / hi8 plane by lane & 4, column n0 + wc*64 + (lane & 3) * 16).
    u32x2_t vh[4];
    unsigned l8[4], h8[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int c0 = (ni >> 1) * 32 + (ni & 1) * 16 + wq * 4;
      f32x4 v = a[ni] + *(const f32x4*)(bias_lds + c0 * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = gelu_fast8<true>(v[e]);
      split4_x2<true>(v, vh[ni], l8[ni], h8[ni]);
      a[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int chunk = (ni >> 1) * 4 + (ni & 1) * 2 + (wq >> 1);
      *(u32x2_t*)(stg + wrow * 128 + ((chunk ^ (wrow & 7)) << 4) + (wq & 1) * 8) = vh[ni];
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = j * 8 + rrow;
      const u32x4 o = *(const u32x4*)(stg + row * 128 + ((rch ^ (row & 7)) << 4));
      __builtin_amdgcn_raw_buffer_store_b128(o, rsC, col_ok ? goff + (unsigned)(j * 8) * ldc2 : 0xFFFFFFF0u, 0, 0);
    }
    // FP8 planes: dword (wq ^ 2 (wrow >> 3)) of 16-byte chunk ((ni >> 1) * 2 + (ni & 1) + 4 * plane) ^ (wrow & 7) - rows r and r + 8 of a
    // 32-lane write group land on different banks (ds_write_b32: 32 banks = one 128-byte row); the reader of rows 8..15 swaps the halves back
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int chunk = (ni >> 1) * 2 + (ni & 1) + 4 * pl;
        *(unsigned*)(stg + wrow * 128 + ((chunk ^ (wrow & 7)) << 4) + ((wq ^ ((wrow >> 3) << 1)) << 2)) = pl ? h8[ni] : l8[ni];
      }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = j * 8 + rrow;
      u32x4 o = *(const u32x4*)(stg + row * 128 + ((rch ^ (row & 7)) << 4));
      if (j) o = u32x4{o[2], o[3], o[0], o[1]};
      __builtin_amdgcn_raw_buffer_store_b128(o, rsC, col_ok8 ? goff8 + (unsigned)(j * 8) * ldc2 : 0xFFFFFFF0u, 0, 0);
    }
    return;
  }
  if constexpr (KIND == G8_TAB_F32) {
    // fp32 output: the 2 KiB staging slot takes 16 rows x 32 columns, so a piece leaves in two halves of two stores each (8 rows x
    // 128 B = whole lines; the generic epilogue's fragment-wise stores are 16 rows x 64 B, half lines at twice the cost per byte)
    f32x4 t[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {   // all four table loads in flight before the first use
      const int c0 = (ni >> 1) * 32 + (ni & 1) * 16 + wq * 4;
      t[ni] = c0 < ncols ? *(const f32x4*)(trow + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int ni = half * 2 + q;
        const int c0 = half * 32 + q * 16 + wq * 4;
        const f32x4 v = a[ni] + *(const f32x4*)(bias_lds + c0 * 4) + t[ni];
        *(f32x4*)(stg + wrow * 128 + (((q * 4 + wq) ^ (wrow & 7)) << 4)) = v;
        a[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = j * 8 + rrow;
        const u32x4 o = *(const u32x4*)(stg + row * 128 + ((rch ^ (row & 7)) << 4));
        // (always issued: the seam's counted waits assume 32 stores per wave and tile; lanes past N / rows past M are dropped by the range check)
        const bool ok = half * 32 + rch * 4 < ncols;
        __builtin_amdgcn_raw_buffer_store_b128(o, rsC, ok ? goff + (unsigned)(j * 8) * ldc2 + (unsigned)(half * 128) : 0xFFFFFFF0u, 0, 0);
      }
    }
    return;
  }
  f32x4 tb[4];
  if constexpr (KIND == G8_TAB_H16) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int c0 = (ni >> 1) * 32 + (ni & 1) * 16 + wq * 4;
      tb[ni] = c0 < ncols ? *(const f32x4*)(trow + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int c0 = (ni >> 1) * 32 + (ni & 1) * 16 + wq * 4;         // first of this lane's 4 columns inside the wave's 64
    f32x4 v = a[ni];
    if constexpr (KIND == G8_TAB_H16) v += *(const f32x4*)(bias_lds + c0 * 4) + tb[ni];
    if constexpr (KIND == G8_GELU_BF16) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = gelu_fast8<F16>(v[e]);
    }
    if constexpr (KIND == G8_SCALE_BF16) v *= *(const f32x4*)(gam_lds + c0 * 4);
    const int chunk = (ni >> 1) * 4 + (ni & 1) * 2 + (wq >> 1);
    *(u32x2*)(stg + wrow * 128 + ((chunk ^ (wrow & 7)) << 4) + (wq & 1) * 8) = pack4_h_ovfl<F16>(v);   // 2 x v_cvt_pk (RNE)
    if constexpr (KIND == G8_TAB_H16) a[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    else a[ni] = *(const f32x4*)(bias_next + c0 * 4);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = j * 8 + rrow;
    const u32x4 o = *(const u32x4*)(stg + row * 128 + ((rch ^ (row & 7)) << 4));
    if constexpr (LAB & 1) { asm volatile("" ::"v"(o)); continue; }   // lab build only: epilogue without the global stores
    // ALWAYS issued (the seam's counted vmcnt waits assume 16 stores per wave and tile): rows past M fall outside the descriptor's
    // range, lanes whose columns lie past N get an out-of-range offset - the hardware drops both
    __builtin_amdgcn_raw_buffer_store_b128(o, rsC, col_ok ? goff + (unsigned)(j * 8) * ldc2 : 0xFFFFFFF0u, 0, 0);
  }
}

// The same piece WITHOUT the LDS round trip (round 2): lanes (r, q) = (lane & 15, lane >> 4) hold row r and the columns q*4 .. q*4+3 of
// each of the four 16-column blocks.  One v_permlane16_swap per packed dword between the two blocks of a pair (16-lane row 1 <-> row 0,
// row 3 <-> row 2) leaves every lane with 8 CONSECUTIVE columns of its row - lane rows 0, 2, 1, 3 hold columns 0-7, 8-15, 16-23, 24-31 of
// the pair - i.e. one 16-byte store per lane and pair, 16 rows x 64 B per wave-instruction (tools/store_probe.hip: within 6 % of the
// 8 rows x 128 B form), and no ds_write / lgkmcnt / ds_read chain per piece (the staged form spent ~5 800 cycles per tile in it).
//   rsC: buffer descriptor of C with num_records = M * ldc bytes; goff: byte offset in C of (row m0 + wr*128 + mi*16 + (lane & 15),
//   column n0 + wc*64 + (lane >> 4 & 1) * 16 + (lane >> 5) * 8); col_ok0 / col_ok1: the lane's 8 columns of pair 0 / 1 are inside N.
template <int KIND, bool F16, int LAB>
__device__ __forceinline__ void g8_piece_reg(f32x4 (&a)[4], const __amdgpu_buffer_rsrc_t rsC, unsigned goff, bool col_ok0, bool col_ok1,
                                             const char* bias_next, const char* gam_lds, int lane) {
  const int wq = lane >> 4;
  u32x2 pk[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int c0 = (ni >> 1) * 32 + (ni & 1) * 16 + wq * 4;
    f32x4 v = a[ni];                                      // (the accumulators started from the bias: see g8_piece)
    if constexpr (KIND == G8_GELU_BF16) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = gelu_fast8<F16>(v[e]);
    }
    if constexpr (KIND == G8_SCALE_BF16) v *= *(const f32x4*)(gam_lds + c0 * 4);
    pk[ni] = pack4_h_ovfl<F16>(v);
    a[ni] = *(const f32x4*)(bias_next + c0 * 4);
  }
#pragma unroll
  for (int pr = 0; pr < 2; ++pr) {
    const u32x2 s0 = __builtin_amdgcn_permlane16_swap(pk[2 * pr][0], pk[2 * pr + 1][0], false, false);
    const u32x2 s1 = __builtin_amdgcn_permlane16_swap(pk[2 * pr][1], pk[2 * pr + 1][1], false, false);
    const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
    if constexpr (LAB & 1) { asm volatile("" ::"v"(o)); continue; }
    // ALWAYS issued (the tile seam's counted vmcnt waits assume 16 stores per wave and tile): rows past M fall outside the buffer
    // descriptor's range, columns past N get an out-of-range offset - the hardware drops both
    __builtin_amdgcn_raw_buffer_store_b128(o, rsC, (pr ? col_ok1 : col_ok0) ? goff + (unsigned)(pr * 64) : 0xFFFFFFF0u, 0, 0);
  }
}

// One FP8 MFMA of the fp16x2 K-tiles: 128 bytes of a weight row against 128 bytes of an activation row.  The lane's 32 operand bytes are
// the SAME two 16-byte LDS fragments the fp16 MFMAs of a K-tile use (k halves 0 and 1 of lane group g: bytes 16 g.. and 64 + 16 g.. of the
// 128-byte row) - which 32 of the 128 k positions a lane group holds does not matter as long as both operands agree, and the scales
// are per plane, not per 32-element block, so the fragment reads and the LDS image are exactly the fp16 kernel's.
// Operand A = weights (e4m3, cbsz 0, scale byte sw), operand B = activations (e5m2, blgp 1, scale byte sa): tools/fp8_mfma_probe.hip.
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;
__device__ __forceinline__ f32x4 g8_mfma_f8(bf16x8 w0, bf16x8 w1, bf16x8 a0, bf16x8 a1, f32x4 c, int sw, int sa) {
  const i32x8 w = __builtin_shufflevector(__builtin_bit_cast(i32x4, w0), __builtin_bit_cast(i32x4, w1), 0, 1, 2, 3, 4, 5, 6, 7);
  const i32x8 a = __builtin_shufflevector(__builtin_bit_cast(i32x4, a0), __builtin_bit_cast(i32x4, a1), 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w, a, c, 0, 1, 0, sw, 0, sa);
}
// The 16 (fp16) / 8 (FP8) MFMAs of one phase: accumulator quadrant acc[AO .. AO + 3][FO .. FO + 1], weight fragments BQ, activation fragments af
#define G8_MM(AO, FO, BQ)                                                                                                           \
  do {                                                                                                                              \
    if constexpr (f8 && !(LAB & 32768)) {   /* lab, LAB & 32768: the FP8 K-tiles' MFMAs as fp16 ones (timing only: is it the FP8 MFMA?) */    \
      _Pragma("unroll") for (int fi = 0; fi < 4; ++fi)                                                                              \
        _Pragma("unroll") for (int f = 0; f < 2; ++f)                                                                               \
          acc[AO + fi][FO + f] = g8_mfma_f8(BQ[f][0], BQ[f][1], af[fi][0], af[fi][1], acc[AO + fi][FO + f], sca, scb);              \
      break;                                                                                                                        \
    }                                                                                                                               \
    _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                                                                \
      _Pragma("unroll") for (int fi = 0; fi < 4; ++fi)                                                                              \
        _Pragma("unroll") for (int f = 0; f < 2; ++f)                                                                               \
          if constexpr (LAB & 8) asm volatile("" ::"v"(BQ[f][kh]), "v"(af[fi][kh]));                                                \
          else acc[AO + fi][FO + f] = mfma16x16x32_h<F16>(BQ[f][kh], af[fi][kh], acc[AO + fi][FO + f]);                             \
  } while (0)

// Lane id recomputed in place (two VALU, no live range): values derived from the kernel's `lane` and kept across the K loop get
// spilled at 256 VGPRs, and a spill reload is a VMEM load whose wait drains the LDS-DMA stream.
__device__ __forceinline__ int g8_lane_now() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

template <int N> __device__ __forceinline__ void g8_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// LAB (0 in the shipped library; other values are only instantiated under -DEC_G8_LAB by tools/g8_lab.py): ablations for
// locating the bottleneck - 1 no global stores, 2 every tile loads tile 0's operands (L2-resident), 4 no LDS-DMA in the steady
// state, 8 no MFMAs, 32 no epilogue at all, 64 the load stream is drained at the tile seam (the round-1 seam, for A/B),
// 512 (round 4) s_memtime stamps of one wave per wave group at the tile milestones (kernel start | per tile: K loop start, K loop
// end, epilogue end | kernel end), kept in LDS and dumped to p.aux[blockIdx][64] at the end: the per-tile cycle ledger of DESIGN.md;
// 8192 / 16384 (round 4) no per-phase s_setprio flips: a static priority 1 for the younger wave group (MI355X_MICROARCH.md "Two waves
// per SIMD", item 4) / no priorities at all - measured: QKV 67.1-69.3 vs 67.2-69.4 / 66.7-67.4 us, 4096^3 98.6-100.1 vs 98.6-100.1 /
// 98.3-100.9, fc2 73.7-75.4 vs 73.7-74.0 / 73.7-74.5: no difference either way (profiles/r04_g8_setprio_ab.txt), the flips stay;
// 4096 (round 4) the prologue waits for all five half-tiles before the first phase (the round-3 form, for A/B);
// 2048 (round 4) no LDS fragment reads after the first K-tile pair of a workgroup (the MFMAs reuse the registers: what the ds_read
// traffic of the partner group costs the MFMA blocks); 1024 (round 4) every tile STORES to the rows of tile row 0 (the output of a launch aliases onto 256 x N: dirty lines stay in the
// L2s, nothing is written back in the burst): what the seam costs without the fabric write-back.
//
// Tile seam of the 16-bit output kinds (KIND != GENERIC): the epilogue runs at the END of the tile, both wave groups at once
// (all four SIMDs convert and store), and NOTHING is drained: the bias / LayerScale slices come from LDS (one small LDS-DMA per
// wave, issued a whole tile ahead), the next tile's operand stream keeps running 5 half-tiles ahead through the epilogue, and
// the counted waits of the next tile's first three phases are widened by the 16 stores per wave that sit between the half-tiles
// in the in-order VM queue:
//     after issuing stream index q+5 in phase q, index q+2 (issued in phase q-3) must have landed; everything younger may stay in
//     flight: the 6 LDS-DMA pieces of q+3..q+5, the 16 stores if they were issued after phase q-3, the bias pieces after theirs.
// Measured against this (tools/g8_lab.py, same process, QKV shape): the seam with a drain 64-65 us; this epilogue PIPELINED into
// the memory segments of the last / first K-tile's phases (two pieces per phase, partner group in its MFMA block) 74-76 us (fc1 +
// GELU 117 vs 101 us): work moved into one wave's memory segment stretches that barrier interval for the partner's MFMA block too
// (MI355X_MICROARCH.md "Two waves per SIMD", item 3), so it was removed.  Also rejected in round 1: one barrier per phase with the
// groups half a phase apart (8192^3 1300 vs 1355 TFLOP/s) and two 32-MFMA phases per K-tile (+2 % at 8192^3, 0 at K = 768).
template <int KIND, int TAG, bool F16 = false, int LAB = 0, bool X2 = false>
__global__ __launch_bounds__(512) void gemm8_bf16_kernel(GemmP p) {
  static_assert(!X2 || (F16 && (KIND == G8_F32 || KIND == G8_RES_F32 || KIND == G8_GELU_X2)), "fp16x2 operands: fp16 main product, three epilogues");
  static_assert(KIND != G8_GELU_X2 || X2, "the fp16x2 output format is written by the fp16x2 GEMM");
  // Epilogue pieces by lane swaps (no LDS round trip: fc1 + GELU, VALU-bound, 101 -> 95 us) or through the LDS staging slot (whole
  // 128-byte lines per store: the bias / LayerScale kinds are bound by the stores themselves, QKV 64.6 vs 66.4 us); LAB & 128 flips it.
  constexpr bool REGEPI = (KIND == G8_GELU_BF16) != ((LAB & 128) != 0);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool FAST = KIND != G8_GENERIC && !(LAB & 32);   // bias from LDS, epilogue pieces through the staging slot
  constexpr bool NODRAIN = FAST && !(LAB & 64);              // the load stream is not drained at the seam
  constexpr bool GAMMA = KIND == G8_SCALE_BF16 || KIND == G8_RES_F32;
  constexpr int NB = GAMMA ? 2 : 1;                          // LDS-DMA pieces of one bias (+ LayerScale) slice
  constexpr int NST = (KIND == G8_GELU_X3 || KIND == G8_GELU_X2 || g8_f32_out(KIND)) ? 32 : 16;   // global stores of one tile's epilogue per wave
  constexpr bool WRAP = !X2 && (KIND == G8_GENERIC || KIND == G8_TAB_F32 || KIND == G8_F32 || KIND == G8_RES_F32 || KIND == G8_GELU_X3);   // GemmP::kwrap
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  auto stamp = [&](int idx) {           // lab only (LAB & 512): wave 0 / wave 4, lane 0 -> LDS
    if constexpr (LAB & 512) {
      if ((wave & 3) == 0 && g8_lane_now() == 0) ((unsigned long long*)(smem + G8_TRACE))[wr * 32 + idx] = __builtin_amdgcn_s_memtime();
    }
  };
  if constexpr (LAB & 512) {
    if (tid < 64) ((unsigned long long*)(smem + G8_TRACE))[tid] = 0ull;
    __syncthreads();
  }
  stamp(0);

  const int ntm = (p.M + 255) >> 8, ntn = (p.N + 255) >> 8;
  const int ntiles = ntm * ntn;
  const int nk = p.K >> 6;                       // K-tiles per output tile (even: checked on the host)
  const long lda_b = p.lda * 2, ldb_b = p.ldb * 2;
  const unsigned lda8 = (unsigned)lda_b * 8u, ldb8 = (unsigned)ldb_b * 8u;

  // XCD-aware persistent schedule: workgroup b runs on XCD b % 8 (observed, speed only); XCD x walks the contiguous tile
  // range [x*chunk, (x+1)*chunk) (row-major, n fastest) so the tiles in flight on one L2 share operand panels.
  // (round 3: for ANY grid size - the N = 768 shapes have 246 tiles = 246 workgroups, and the former "grid % 8 == 0" condition sent
  // their three column tiles of one A panel to three different XCDs.)  XCD x owns the workgroups b = x, x + 8, ... (nslot of them)
  // and the share of the tile list proportional to that count.
  const int nxcd = gridDim.x >= 8 ? 8 : 1;
  const int gq = gridDim.x / nxcd, gr = gridDim.x % nxcd;
  const int xcd = blockIdx.x % nxcd, slot = blockIdx.x / nxcd, nslot = gq + (xcd < gr ? 1 : 0);
  const int w0 = xcd * gq + min(xcd, gr);                                // workgroups on the XCDs before this one
  const int t_begin = (int)((long)ntiles * w0 / gridDim.x), t_end = (int)((long)ntiles * (w0 + nslot) / gridDim.x);
  const int t_first = t_begin + slot;
  // DYNAMIC schedule (round 3; p.sched != nullptr, the host passes it only for more tiles than workgroups): the first tile of a
  // workgroup is the static one, every further tile is taken from its XCD's counter - tile t_begin + nslot + atomicAdd(sched[xcd], 1) -
  // so the XCD's tiles are still walked in row-major order, but a workgroup that starts late (its CU was held by another stream's
  // kernel when the launch began: ec_forward_pipelined runs the head beside the backbone) simply takes fewer of them instead of making
  // the whole launch late by its delay.  The load stream crosses tile boundaries and the bias slice is staged a tile ahead, so the
  // schedule is known TWO tiles ahead: t (computing), t_nxt, and t_nn, which wave 0 fetches during tile t with a SCALAR atomic
  // (s_atomic_add: the value returns into an SGPR and is counted by lgkmcnt - the vector-memory queue with its counted waits and
  // the 256 VGPRs of the loop are not touched), collects seven phases later where no LDS read is outstanding, and hands to the
  // other waves through an LDS slot.
  // The last workgroup to leave re-arms the counters (sched[8] counts leavers) for the next launch on the stream.
  // (compiled into the bias and bias + GELU kinds only - QKV and fc1, the multi-round shapes of the backbone; the LayerScale kind of
  //  proj / fc2, one tile per workgroup, has no register to spare for it)
  constexpr bool DYN_OK = KIND == G8_BIAS_BF16 || KIND == G8_GELU_BF16 || (X2 && (KIND == G8_F32 || KIND == G8_GELU_X2));   // (fp16x2: QKV, fc1)
  const bool dyn = DYN_OK && p.sched != nullptr;
  // A workgroup is counted as a leaver as soon as it knows that it will not ask again (its last answer was past the range), i.e.
  // during its last tile's K loop, not at its end: the count's round trip is off the launch's tail.  left_s: the scalar atomic's
  // operand (1) / return value (the number of earlier leavers).
  int left_s = 1;
  bool left = false;
  auto sched_rearm = [&](int old) {             // the last leaver zeroes the counters for the next launch on the stream
    if (old == (int)gridDim.x - 1 && tid == 0) {
#pragma unroll
      for (int x = 0; x < 9; ++x) __hip_atomic_store(p.sched + x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  if (t_first >= t_end) {                       // whole workgroup leaves: no barrier has been executed yet
    if (dyn) {
      int old = 0;
      if (tid == 0) old = __hip_atomic_fetch_add(p.sched + 8, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      sched_rearm(__builtin_amdgcn_readfirstlane(old));
    }
    return;
  }
  int t_nxt = t_first + nslot, t_nn = t_first + 2 * nslot;
  int* const sched_lds = (int*)(smem + G8_SCHED);
  int fetch_s = 1;                              // wave 0: operand (1) and return value of the scalar atomic in flight
  int pro_s = 1;                                // second tile of this workgroup: asked for now (scalar atomic, as in the K loop: the value
                                                // returns into an SGPR and is counted by lgkmcnt - it stays out of the vector-memory queue
                                                // whose counted wait ends the prologue), handed over behind that wait
  if (dyn && wave == 0) asm volatile("s_atomic_add %0, %1, 0x0 glc" : "+s"(pro_s) : "s"(p.sched + xcd) : "memory");

  // ---- load stream (LDS-DMA) state -------------------------------------------------------------------------------
  // wave w stages row-group w (16 rows) of every half-tile as two pieces of 8 rows (2 wave-instructions of 1 KiB).
  // lane -> k half lane>>5, row (lane>>2)&7 of the octet, physical 16-B chunk lane&3 which holds logical chunk (lane&3) ^ 2*(octet).
  // The stream uses buffer loads to LDS (buffer_load_dwordx4 ... lds): one 128-bit descriptor per operand in SGPRs, a 32-bit
  // per-lane byte offset of the lane's row + chunk (fixed for a whole output tile) and the K position as the scalar offset -
  // four offset VGPRs instead of four 64-bit row pointers, no per-issue address arithmetic.
  // (host: operand bytes < 4 GiB.  The descriptors end with the operands' last row: the second piece of a clamped edge row group
  // reaches up to 8 rows past it, reads zeros there - range check - and those rows / columns are never stored)
  const unsigned kphys = (WRAP && p.kwrap) ? 128u * (unsigned)p.kwrap : (unsigned)p.K;   // elements of an operand row that exist in memory
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (int)(((unsigned)(p.M - 1) * (unsigned)p.lda + kphys) * 2u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, (int)(((unsigned)(p.N - 1) * (unsigned)p.ldb + kphys) * 2u), 0x00020000);
  const int kwA = (WRAP && p.kwrap) ? 2 * p.kwrap : 0x7fffffff, kwB = (WRAP && p.kwrap) ? p.kwrap : 0x7fffffff;   // first K-tile of the wrapped plane
  unsigned vo0, vo1, vo2, vo3;                       // A-h0, B-h0, B-h1, A-h1: byte offset of this lane's row (+ chunk)
  // (round 3, measured and removed: a per-tile ROTATION of the K walk, so that the workgroups sharing an operand panel are at
  // different K positions, and a start stagger over the slots.  Neither changes FETCH_SIZE / TCC_MISS on any of the four block
  // shapes - workgroups that request the same line in lock-step are merged by the L2 - and rotation costs 3-17 % of the time
  // because it widens the working set; the stagger de-phases the chip-wide store bursts of the tile seams but costs its own delay
  // on the critical workgroups, net +1.5 / +2.2 / +4.6 % at 3 / 6 / 12 k cycles (interleaved medians);
  // profiles/r03_g8_sched_sweep.txt, r03_g8_sched_pmc.csv, r03_g8_stagger_ab.txt.)
  // (round 3, measured and removed: a K SERPENTINE - every second tile of a workgroup walks K backwards, so that a round starts on the
  // K-tiles the previous one ended on - cuts the weight-panel re-reads by 8-9 % (QKV 129 -> 119 MB, fc1 199 -> 182 MB per launch) and
  // changes the time by +1.8 % / +0.3 % (noise); profiles/r03_g8_serpentine_ab.txt.)
  int ls_kt = 0, ls_tile = t_first;
  auto set_rows = [&](int t) {
    const int m0 = (LAB & 2) ? 0 : (t / ntn) << 8, n0 = (LAB & 2) ? 0 : (t % ntn) << 8;
    // everything per-lane is recomputed from the lane id here (once per tile, a dozen VALU): kept live across the K loop these
    // values get spilled, and their reload (scratch is VMEM) would drain the whole load stream at every tile change
    const int l = g8_lane_now();
    // first piece (rows 0..7 of the group): row, k half, chunk  (LAB & 256: the round-1 pieces of 16 rows x 64 B, for A/B)
    const int r = (LAB & 256) ? l >> 2 : (l >> 2) & 7, c = (LAB & 256) ? ((l & 3) ^ ((l >> 5) << 1)) << 4 : ((l >> 5) << 6) + ((l & 3) << 4);
    const int ra = m0 + (wave >> 2) * 128 + (wave & 3) * 16 + r;        // A-h0 row; A-h1 = +64
    const int rb = n0 + (wave >> 1) * 64 + (wave & 1) * 16 + r;         // B-h0 row; B-h1 = +32
    vo0 = (unsigned)min(ra, p.M - 1) * (unsigned)lda_b + (unsigned)c;   // rows past the edge: clamped, computed, never stored
    vo3 = (unsigned)min(ra + 64, p.M - 1) * (unsigned)lda_b + (unsigned)c;
    vo1 = (unsigned)min(rb, p.N - 1) * (unsigned)ldb_b + (unsigned)c;
    vo2 = (unsigned)min(rb + 32, p.N - 1) * (unsigned)ldb_b + (unsigned)c;
  };
  bool lab_steady = false;
  // Past the workgroup's last tile the stream keeps issuing (same rows again, valid addresses) into the ring position the live
  // stream would use - free by the same WAR argument and never read - so the counted waits stay uniform to the end.
  auto issue = [&](const __amdgpu_buffer_rsrc_t rs, unsigned vo, unsigned ld8, int half) {
    if constexpr (LAB & 4) { if (lab_steady) return; }
    char* dst = smem + (ls_kt & 1) * G8_KT + half * G8_HALF + wave * 2048;
    int kpos = ls_kt;                                  // K-tile of the operand's memory image (half 0 / 3: A, 1 / 2: B)
    if constexpr (WRAP) kpos = (half == 0 || half == 3) ? (ls_kt >= kwA ? ls_kt - kwA : ls_kt) : (ls_kt >= kwB ? ls_kt - kwB : ls_kt);
    // (the K position and the second piece's +64 B go into the SCALAR offset: the instruction's immediate offset would be added to
    // the LDS address as well as to the memory address)
    // second piece: rows 8..15 of the group (+ 8 rows; their chunks are stored XOR 2: the swizzle of the reader)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)dst, 16, vo, kpos * 128, 0, 0);
    if constexpr (LAB & 256) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(dst + 1024), 16, vo, kpos * 128 + 64, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(dst + 1024), 16, (vo ^ 32u) + ld8, kpos * 128, 0, 0);
  };
  auto advance = [&]() {
    if (++ls_kt == nk) {
      ls_kt = 0;
      ls_tile = t_nxt;
      if (ls_tile < t_end) set_rows(ls_tile);
    }
  };
  // bias / LayerScale slice of output tile t for this wave's 64 columns -> LDS (parity = tile counter & 1): 64 lanes x 4 B;
  // columns past N read as zero (buffer range check) and are never stored
  auto stage_bias = [&](int t, int parity) {
    if constexpr (FAST) {
      const unsigned vo = (unsigned)(((t % ntn) << 8) + wc * 64 + g8_lane_now()) * 4u;
      const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.N * 4, 0x00020000);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb_, (lptr_t)(smem + G8_BIAS + (parity * 8 + wave) * 256), 4, vo, 0, 0, 0);
      if constexpr (GAMMA) {
        const __amdgpu_buffer_rsrc_t rg_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.gamma), 0, p.N * 4, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rg_, (lptr_t)(smem + G8_GAMMA + (parity * 8 + wave) * 256), 4, vo, 0, 0, 0);
      }
    }
  };
  set_rows(t_first);
  stage_bias(t_first, 0);
  issue(rsA, vo0, lda8, 0); issue(rsB, vo1, ldb8, 1); issue(rsB, vo2, ldb8, 2); issue(rsA, vo3, lda8, 3);
  advance();
  issue(rsA, vo0, lda8, 0);
  // The first phase reads stream indices 0 and 1 (A-h0, B-h0 of the first K-tile) and the bias slice: the oldest NB + 4 of the NB + 10
  // pieces in flight.  Kinds whose seam is not drained wait for exactly those (round 4; the phases' own counted waits - "after issuing
  // index q + 5, index q + 2 has landed" - take over from there): the MFMAs start under the rest of the cold 80 KiB burst instead of
  // behind it (the prologue is 6 % of a QKV launch, 14 % of a proj launch: profiles/r04_g8_cycle_ledger.txt).  The generic kind's first
  // tile has no counted waits (its seam is drained): it waits for everything.
  if constexpr (NODRAIN && !(LAB & 4096)) g8_wait_vm<6>(); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lab_steady = true;
  if (dyn && wave == 0) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(pro_s));
    if (g8_lane_now() == 0) sched_lds[0] = t_begin + nslot + pro_s;
  }
  G8_BAR();
  if (dyn) {                                         // second tile of this workgroup (fetched by thread 0 at the top)
    t_nxt = __builtin_amdgcn_readfirstlane(sched_lds[0]);
    t_nn = t_end;
  }
  if (wr == 1) G8_BAR();                             // stagger: group 1 runs one barrier behind group 0

  // ---- fragment read addresses ------------------------------------------------------------------------------------
  // reader lane: row r = lane&15 of the 16-row sub-tile, logical chunk lane>>4 at physical chunk (lane>>4) ^ 2*(r>=8)
  constexpr int KH = (LAB & 256) ? 1024 : 512;   // byte distance of the two k halves of a fragment
  const int rd_off = ((LAB & 256) ? (lane & 15) << 6 : (((lane & 15) >> 3) << 10) + ((lane & 7) << 6)) + ((((lane >> 4) ^ (((lane & 15) >> 3) << 1))) << 4);
  const char* a_base = smem + rd_off + wr * 8192;                // row-groups 4*wr.. of the A halves
  const char* b_base = smem + rd_off + wc * 4096 + G8_HALF;      // row-groups 2*wc.. of the B halves (B-h0 is slot 1)
  char* const stg = smem + G8_STAGE + wave * 2048;

  // Bias kinds: the accumulators start from the first tile's bias slice (parity 0: landed behind the prologue's vmcnt(0), staged by
  // this wave itself), every later tile's from the slice its predecessor's epilogue loads (g8_piece)
  constexpr bool BIASACC = FAST && (KIND == G8_BIAS_BF16 || KIND == G8_SCALE_BF16 || KIND == G8_GELU_BF16);
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (BIASACC) acc[i][j] = *(const f32x4*)(smem + G8_BIAS + wave * 256 + ((j >> 1) * 32 + (j & 1) * 16 + (lane >> 4) * 4) * 4);
      else acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  bf16x8 af[4][2], b0[2][2], b1[2][2];

  const unsigned ldc2 = (unsigned)p.ldc * (g8_f32_out(KIND) ? 4u : 2u);   // row pitch of C in bytes
  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)((unsigned)p.M * ldc2), 0x00020000);   // (host: < 2 GiB)
  auto pieces = [&](int mi0, int m0, int n0, int par) {   // pieces mi0, mi0 + 1 of tile (m0, n0)
    if constexpr (FAST) {
      const int lane = g8_lane_now();                      // (shadows the kernel's: see g8_lane_now)
      const char* bl = smem + G8_BIAS + (par * 8 + wave) * 256;
      const char* bn = smem + G8_BIAS + ((par ^ 1) * 8 + wave) * 256;     // the next tile's slice (staged a tile ahead)
      const char* gl = smem + G8_GAMMA + (par * 8 + wave) * 256;
      if constexpr (REGEPI) {
        const int cq = ((lane >> 4) & 1) * 16 + (lane >> 5) * 8;          // first of the lane's 8 columns inside a 32-column pair
        const bool ok0 = n0 + wc * 64 + cq < p.N, ok1 = n0 + wc * 64 + 32 + cq < p.N;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const int r0 = m0 + wr * 128 + (mi0 + d) * 16 + (lane & 15);
          const unsigned goff = (unsigned)r0 * ldc2 + (unsigned)(n0 + wc * 64 + cq) * 2u;
          g8_piece_reg<KIND == G8_GENERIC ? G8_BIAS_BF16 : KIND, F16, LAB>(acc[mi0 + d], rsC, goff, ok0, ok1, bn, gl, lane);
        }
      } else {
        const bool col_ok = n0 + wc * 64 + (lane & 7) * 8 < p.N;
        constexpr unsigned esz = g8_f32_out(KIND) ? 4u : 2u;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const int r0 = m0 + wr * 128 + (mi0 + d) * 16;
          const unsigned goff = (unsigned)(r0 + (lane >> 3)) * ldc2 + (unsigned)(n0 + wc * 64) * esz + (unsigned)(lane & 7) * 16u;
          if constexpr (KIND == G8_TAB_H16 || KIND == G8_TAB_F32) {
            const float* trow = p.table + (long)((r0 + (lane & 15)) % p.period) * p.ldt + n0 + wc * 64;
            g8_piece<KIND, F16, LAB>(acc[mi0 + d], rsC, goff, ldc2, col_ok, stg, bl, gl, lane, trow, min(max(p.N - n0 - wc * 64, 0), 64));
          } else if constexpr (KIND == G8_GELU_X2) {
            const unsigned goff8 = (unsigned)(r0 + (lane >> 3)) * ldc2 + (unsigned)p.N * (2u + ((lane >> 2) & 1)) + (unsigned)(n0 + wc * 64) + (unsigned)(lane & 3) * 16u;
            g8_piece<KIND, F16, LAB>(acc[mi0 + d], rsC, goff, ldc2, col_ok, stg, bl, gl, lane, nullptr, 64, nullptr, (unsigned)p.N, goff8,
                                     n0 + wc * 64 + (lane & 3) * 16 < p.N);
          } else if constexpr (KIND == G8_F32 || KIND == G8_RES_F32 || KIND == G8_GELU_X3) {
            g8_piece<KIND, F16, LAB>(acc[mi0 + d], rsC, goff, ldc2, col_ok, stg, bl, gl, lane, nullptr, min(max(p.N - n0 - wc * 64, 0), 64), nullptr,
                                     (unsigned)p.N * 2u);
          } else {
            g8_piece<KIND == G8_GENERIC ? G8_BIAS_BF16 : KIND, F16, LAB>(acc[mi0 + d], rsC, goff, ldc2, col_ok, stg, bl, gl, lane, nullptr, 64, bn);
          }
        }
      }
    }
  };

  if constexpr (LAB & 8192) { if (wr == 1) __builtin_amdgcn_s_setprio(1); }   // lab: STATIC priority for the younger wave group, no per-phase flips
  const int nk16 = X2 ? p.x2 : 0;                      // fp16x2: number of fp16 K-tiles (the other nk - nk16 = nk16 are FP8)
  // E8M0 scale bytes of the weight row's two FP8 planes, replicated into all four bytes of the MFMA's scale register: every lane and
  // byte carries the plane's scale, so the result does not depend on which (lane, byte) the hardware reads for a 32-element block
  const int x2_sa0 = (p.x2_sa & 0xff) * 0x01010101, x2_sa1 = ((p.x2_sa >> 8) & 0xff) * 0x01010101;
  int it = 0;                                          // tile counter of this workgroup (bias parity)
  for (int t = t_first; t < t_end; t = t_nxt, t_nxt = t_nn, t_nn = dyn ? t_end : t_nxt + nslot, ++it) {
    const int m0 = (t / ntn) << 8, n0 = (t % ntn) << 8;
    if (it < 9) stamp(1 + 3 * it);
    // One PAIR of K-tiles (both LDS buffers).  F8 (fp16x2 operands only): the pair's MFMAs are the FP8 ones - a second instantiation of
    // the same body in a loop of its own, so neither loop carries a branch around its MFMA blocks.
    auto kpair = [&](auto f8c, const int kt2) __attribute__((always_inline)) {
      constexpr bool f8 = decltype(f8c)::value;
#pragma unroll
      for (int buf = 0; buf < 2; ++buf) {
        const char* ab = a_base + buf * G8_KT;
        const char* bb = b_base + buf * G8_KT;
        const bool head = !f8 && buf == 0 && kt2 == 0;                 // first K-tile of the tile (never an FP8 one)
        const bool seam = NODRAIN && head && it > 0;                   // ... with the previous tile's stores in the VM queue
        const bool rd = !(LAB & 2048) || (it == 0 && kt2 == 0);        // lab: fragment reads only in the workgroup's first K-tile pair
        // fp16x2 operands: K-tiles [0, nk16) are fp16 MFMAs, [nk16, nk16 + nk16 / 2) FP8 MFMAs of A's lo8 plane against B's hi8 plane,
        // the rest A's hi8 plane against B's lo8 plane (uniform per K-tile: scalar branch, scalar selects)
        const bool pl1 = f8 && kt2 + buf >= nk16 + (nk16 >> 1);
        const int sca = pl1 ? x2_sa1 : x2_sa0, scb = (pl1 ? X2_SCALE_HI8 : X2_SCALE_LO8) * 0x01010101;
        (void)sca; (void)scb;
        // ---------------- phase 0: quadrant (m-half 0, n-half 0)
        if (rd) {
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int kh = 0; kh < 2; ++kh) b0[f][kh] = *(const bf16x8*)(bb + f * 2048 + kh * KH);
#pragma unroll
        for (int fi = 0; fi < 4; ++fi)
#pragma unroll
          for (int kh = 0; kh < 2; ++kh) af[fi][kh] = *(const bf16x8*)(ab + fi * 2048 + kh * KH);
        }
        issue(rsB, vo1, ldb8, 1);
        if constexpr (NODRAIN) {
          if (seam) g8_wait_vm<6 + NST>(); else g8_wait_vm<6>();
        } else {
          if (!head) g8_wait_vm<6>();                    // first K-tile after a drained seam: nothing to wait for
        }
        if (dyn) {
          if (head) {                                    // wave 0 asks for the tile after next
            fetch_s = 1;
            if (wave == 0 && t_nxt < t_end) asm volatile("s_atomic_add %0, %1, 0x0 glc" : "+s"(fetch_s) : "s"(p.sched + xcd) : "memory");
          } else if (buf == 0 && kt2 == 2) {             // ... and everybody picks it up in the third K-tile (written in phase 3 below)
            t_nn = __builtin_amdgcn_readfirstlane(sched_lds[(it + 1) & 1]);
          }
        }
        G8_BAR();
        if constexpr (!(LAB & 24576)) __builtin_amdgcn_s_setprio(1);
        G8_MM(0, 0, b0);
        if constexpr (!(LAB & 24576)) __builtin_amdgcn_s_setprio(0);
        G8_BAR();
        // ---------------- phase 1: quadrant (0, 1)
        if (rd) {
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int kh = 0; kh < 2; ++kh) b1[f][kh] = *(const bf16x8*)(bb + G8_HALF + f * 2048 + kh * KH);
        }
        issue(rsB, vo2, ldb8, 2);
        if constexpr (NODRAIN) {
          if (seam) g8_wait_vm<6 + NST>(); else g8_wait_vm<6>();
        } else {
          if (!head) g8_wait_vm<6>();
        }
        G8_BAR();
        if constexpr (!(LAB & 24576)) __builtin_amdgcn_s_setprio(1);
        G8_MM(0, 2, b1);
        if constexpr (!(LAB & 24576)) __builtin_amdgcn_s_setprio(0);
        G8_BAR();
        // ---------------- phase 2: quadrant (1, 1)
        if (rd) {
#pragma unroll
        for (int fi = 0; fi < 4; ++fi)
#pragma unroll
          for (int kh = 0; kh < 2; ++kh) af[fi][kh] = *(const bf16x8*)(ab + 3 * G8_HALF + fi * 2048 + kh * KH);
        }
        issue(rsA, vo3, lda8, 3);
        if constexpr (FAST) {
          if (head) {                                                              // third phase of a tile
            // the NEXT tile's bias slice, a whole tile ahead (unconditional so that the counts below are uniform: past the last
            // tile the current slice is staged again into the other parity)
            stage_bias(t_nxt < t_end ? t_nxt : t, (it + 1) & 1);
            if constexpr (NODRAIN) { if (seam) g8_wait_vm<6 + NST + NB>(); else g8_wait_vm<6 + NB>(); }
          } else g8_wait_vm<6>();
        } else {
          if (!head) g8_wait_vm<6>();
        }
        G8_BAR();
        if constexpr (!(LAB & 24576)) __builtin_amdgcn_s_setprio(1);
        G8_MM(4, 2, b1);
        if constexpr (!(LAB & 24576)) __builtin_amdgcn_s_setprio(0);
        G8_BAR();
        // ---------------- phase 3: quadrant (1, 0); the load stream moves on to the next K-tile
        advance();
        issue(rsA, vo0, lda8, 0);
        if constexpr (FAST) {
          if (head) g8_wait_vm<6 + NB>(); else g8_wait_vm<6>();    // (the seam's stores are older than the piece this waits for)
        } else {
          g8_wait_vm<6>();
        }
        if (dyn && buf == 1 && kt2 == 0 && wave == 0) {
          // second K-tile of the tile, seven phases behind the request; this phase has no LDS reads of its own, so lgkmcnt(0) only
          // waits for the scalar atomic (if at all)
          asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(fetch_s));
          const int nn = t_nxt < t_end ? t_begin + nslot + fetch_s : t_end;
          if (g8_lane_now() == 0) sched_lds[(it + 1) & 1] = nn;
          if (nn >= t_end && !left) {                    // no further request from this workgroup: count it now
            left = true;
            asm volatile("s_atomic_add %0, %1, 0x20 glc" : "+s"(left_s) : "s"(p.sched) : "memory");
          }
        }
        G8_BAR();
        if constexpr (!(LAB & 24576)) __builtin_amdgcn_s_setprio(1);
        G8_MM(4, 0, b0);
        if constexpr (!(LAB & 24576)) __builtin_amdgcn_s_setprio(0);
        if (buf == 0 || kt2 + 2 < nk) G8_BAR();   // the tile's last barrier is placed around its epilogue
      }
    };
    for (int kt2 = 0; kt2 < (X2 ? nk16 : nk); kt2 += 2) kpair(std::false_type{}, kt2);
    if constexpr (X2) {
      for (int kt2 = nk16; kt2 < nk; kt2 += 2) kpair(std::true_type{}, kt2);
    }
    // ---- epilogue at the end of the tile.  Both groups run it concurrently: group 0 passes the tile's last barrier first.
    if (it < 9) stamp(2 + 3 * it);
    if (wr == 0) G8_BAR();
    if constexpr (F16) {
      // fp16 outputs saturate at +-65504 in the conversion itself (pack4_h_ovfl; ec_common.h).  The mode bit is not confined to
      // conversions (see fp16_ovfl_mode), so it must not flip under MFMAs still in flight: group 1 has no barrier between the tile's
      // last MFMA block (phase 3: acc[4..7][0..1]) and this point - a VALU read of that block's last accumulator waits for it (the matrix
      // pipe retires in order), and the scheduling barriers keep the conversions of the pieces behind the switch
      G8_SB();
      asm volatile("v_mov_b32 %0, %0" : "+v"(acc[7][1][3]));
      G8_SB();
      fp16_ovfl_mode<1>();
      G8_SB();
    }
    if constexpr (KIND == G8_GENERIC) {
      g8_epilogue_generic<F16>(p, acc, m0, n0, wr, wc, lane);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else if constexpr (FAST) {
      if constexpr (!NODRAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // lab only: the round-1 seam
      const int m0s = (LAB & 1024) ? 0 : m0;
      pieces(0, m0s, n0, it & 1); pieces(2, m0s, n0, it & 1); pieces(4, m0s, n0, it & 1); pieces(6, m0s, n0, it & 1);
    }
    if constexpr (F16) { G8_SB(); fp16_ovfl_mode<0>(); G8_SB(); }
    if (it < 9) stamp(3 + 3 * it);
    if (wr == 1) G8_BAR();
  }
  if (wr == 0) G8_BAR();   // balances group 1's extra barrier at the start
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stream's tail pieces must land before the LDS is released
  if constexpr (LAB & 512) {
    stamp(31);
    __syncthreads();
    if (tid < 64 && p.aux) ((unsigned long long*)p.aux)[(long)blockIdx.x * 64 + tid] = ((const unsigned long long*)(smem + G8_TRACE))[tid];
  }
  if (dyn && wave == 0) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(left_s));   // (requested at least ten K-tiles ago)
    sched_rearm(left_s);
  }
}


}  // namespace

// Per-device launch state: the 160 KiB dynamic-LDS attribute is a per-device function attribute and the CU count differs per
// device, so a process that drives several GPUs (one engine per device) gets both for every device it touches.
namespace {
struct G8Dev { bool attr_done = false, x2_attr_done = false; int ncu = 0; float* zeros = nullptr; };   // zeros: the bias of a table kind called without one
G8Dev g8_dev[64];
}  // namespace

// Returns 1 if this kernel handled the problem, 0 if the shape is not eligible (caller falls back), <0 on error.
int gemm8_bf16(const GemmP& p, hipStream_t st) {
  static const int disable = getenv("EC_GEMM8_OFF") ? atoi(getenv("EC_GEMM8_OFF")) : 0;
  if (disable && !p.x2) return 0;   // (an A/B switch for the 16-bit GEMMs; fp16x2 operands have no other kernel)
  if (!p.ab_bf16 || p.batch != 1 || p.act == ACT_TANHGATE) return 0;
  // (fp16x2 operands: this kernel is their ONLY GEMM - any M, so that an image's features do not depend on the batch it rides in)
  if (p.K % 128 != 0 || p.N % 16 != 0 || (p.M < 1024 && !p.x2) || p.N < 256) return 0;
  if ((long)p.M * p.lda * 2 >= (1l << 32) - (1l << 20) || (long)p.N * p.ldb * 2 >= (1l << 32) - (1l << 20)) return 0;   // 32-bit buffer offsets
  typedef void (*kern_t)(GemmP);
  int dev = 0;
  EC_HIP(hipGetDevice(&dev));
  EC_REQUIRE(dev >= 0 && dev < 64, -1, "gemm8: device ordinal out of range");
  G8Dev& ds = g8_dev[dev];
  if (p.x2) {
    // fp16x2 operands (GemmP::x2): QKV (fp32 out), proj / fc2 (LayerScale + in-place fp32 residual), fc1 (GELU, fp16x2 planes out)
    EC_REQUIRE(p.h_f16 && p.x2 % 2 == 0 && p.K == 128 * p.x2 && !p.kwrap && !p.split && p.bias && !p.table && !p.aux, -1,
               "gemm8: fp16x2 operands take K = 2 K_layer (16-bit units) = 128 * x2, K_layer % 128 == 0, fp16 main plane and a bias");
    int k3;
    if (p.c_x2 && p.act == ACT_GELU && !p.gamma && !p.resid && p.ldc >= 2l * p.N && (long)p.M * p.ldc * 2 < (1l << 31)) k3 = 2;
    else if (!p.c_x2 && !p.c_bf16 && !p.c_x3 && p.act == ACT_NONE && p.ldc % 4 == 0 && (long)p.M * p.ldc * 4 < (1l << 31) && !p.resid && !p.gamma) k3 = 0;
    else if (!p.c_x2 && !p.c_bf16 && !p.c_x3 && p.act == ACT_NONE && p.ldc % 4 == 0 && (long)p.M * p.ldc * 4 < (1l << 31) && p.resid == (const float*)p.C && p.ldr == p.ldc && p.gamma) k3 = 1;
    else { set_error("gemm8: fp16x2 operands with an epilogue other than fp32 | LayerScale + in-place residual | GELU + fp16x2 planes"); return -1; }
    static const kern_t x2_table[3][5] = {
        {gemm8_bf16_kernel<G8_F32, 0, true, 0, true>, gemm8_bf16_kernel<G8_F32, 1, true, 0, true>, gemm8_bf16_kernel<G8_F32, 0, true, 0, true>, gemm8_bf16_kernel<G8_F32, 0, true, 0, true>, gemm8_bf16_kernel<G8_F32, 0, true, 0, true>},
        {gemm8_bf16_kernel<G8_RES_F32, 0, true, 0, true>, gemm8_bf16_kernel<G8_RES_F32, 0, true, 0, true>, gemm8_bf16_kernel<G8_RES_F32, 2, true, 0, true>, gemm8_bf16_kernel<G8_RES_F32, 0, true, 0, true>, gemm8_bf16_kernel<G8_RES_F32, 4, true, 0, true>},
        {gemm8_bf16_kernel<G8_GELU_X2, 3, true, 0, true>, gemm8_bf16_kernel<G8_GELU_X2, 3, true, 0, true>, gemm8_bf16_kernel<G8_GELU_X2, 3, true, 0, true>, gemm8_bf16_kernel<G8_GELU_X2, 3, true, 0, true>, gemm8_bf16_kernel<G8_GELU_X2, 3, true, 0, true>}};
    if (!ds.x2_attr_done) {
      for (int k = 0; k < 3; ++k)
        for (int t = 0; t < 5; ++t) EC_HIP(hipFuncSetAttribute((const void*)x2_table[k][t], hipFuncAttributeMaxDynamicSharedMemorySize, G8_LDS));
      if (!ds.ncu) EC_HIP(hipDeviceGetAttribute(&ds.ncu, hipDeviceAttributeMultiprocessorCount, dev));
      ds.x2_attr_done = true;
    }
    const long nt = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
    GemmP q = p;
    if (nt <= ds.ncu || ds.ncu < 8 || p.x2 < 4 || k3 == 1) q.sched = nullptr;   // dynamic tile schedule: multi-round QKV / fc1 only (the hand-over spans three fp16 K-tiles)
#ifdef EC_G8_LAB   // lab library only (tools/x2_lab.py): EC_X2_LAB_AS16=1 runs the FP8 K-tiles' MFMA blocks as fp16 MFMAs on the same bytes - wrong numbers, same stream
    static const int as16 = getenv("EC_X2_LAB_AS16") ? atoi(getenv("EC_X2_LAB_AS16")) : 0;
    if (as16) {
      static const kern_t lab16[3] = {gemm8_bf16_kernel<G8_F32, 1, true, 32768, true>, gemm8_bf16_kernel<G8_RES_F32, 2, true, 32768, true>, gemm8_bf16_kernel<G8_GELU_X2, 3, true, 32768, true>};
      static bool lab_attr = false;
      if (!lab_attr) { for (int k = 0; k < 3; ++k) EC_HIP(hipFuncSetAttribute((const void*)lab16[k], hipFuncAttributeMaxDynamicSharedMemorySize, G8_LDS)); lab_attr = true; }
      hipLaunchKernelGGL(lab16[k3], dim3((unsigned)(nt < ds.ncu ? nt : ds.ncu)), dim3(512), G8_LDS, st, q);
      EC_LAUNCH_CHECK();
      return 1;
    }
#endif
    hipLaunchKernelGGL(x2_table[k3][p.tag], dim3((unsigned)(nt < ds.ncu ? nt : ds.ncu)), dim3(512), G8_LDS, st, q);
    EC_LAUNCH_CHECK();
    return 1;
  }
  // epilogue kind from the options; TAG only names the symbol for rocprof (1 qkv, 2 proj, 3 fc1, 4 fc2)
  int kind = G8_GENERIC;
  if (p.c_bf16 && p.bias && !p.resid && !p.table && (long)p.M * p.ldc * 2 < (1l << 31)) {
    if (p.act == ACT_NONE && !p.gamma) kind = G8_BIAS_BF16;
    else if (p.act == ACT_NONE && p.gamma) kind = G8_SCALE_BF16;
    else if (p.act == ACT_GELU && !p.gamma) kind = G8_GELU_BF16;
  } else if (p.table && !p.resid && !p.gamma && !p.aux && p.act == ACT_NONE && p.period > 0 && p.ldt % 4 == 0 && p.ldc % 4 == 0 &&
             (long)p.M * p.ldc * (p.c_bf16 ? 2 : 4) < (1l << 31)) {
    kind = p.c_bf16 ? G8_TAB_H16 : G8_TAB_F32;   // (the A/B switch back to the generic epilogue, EC_G8_TAB, went in round 5)
  } else if (p.c_x3 && !p.h_f16 && p.bias && p.act == ACT_GELU && !p.gamma && !p.resid && !p.table && (long)p.M * p.ldc * 2 < (1l << 31)) {
    kind = G8_GELU_X3;
  } else if (!p.c_bf16 && !p.c_x3 && p.bias && !p.table && !p.aux && p.act == ACT_NONE && p.ldc % 4 == 0 && (long)p.M * p.ldc * 4 < (1l << 31)) {
    // (measured against the generic epilogue behind a switch that is gone again, cfg2 bf16x3 / bf16x3, interleaved on one box:
    //  2501 / 2501 vs 2316 / 2297 pairs/s; QKV 201 vs 227 us, proj 89 vs 101, fc1 284 vs 343, fc2 238 vs 243; profiles/r05_x3_ab.txt)
    if (!p.resid && !p.gamma) kind = G8_F32;
    else if (p.resid == (const float*)p.C && p.ldr == p.ldc && p.gamma) kind = G8_RES_F32;
  }
  if (p.kwrap && kind >= G8_BIAS_BF16 && kind <= G8_TAB_H16) kind = G8_GENERIC;   // (the K wrap is compiled into the other kinds only)
#define G8_ROW(F) \
      {gemm8_bf16_kernel<0, 0, F>, gemm8_bf16_kernel<0, 1, F>, gemm8_bf16_kernel<0, 2, F>, gemm8_bf16_kernel<0, 3, F>, gemm8_bf16_kernel<0, 4, F>}, \
      {gemm8_bf16_kernel<1, 0, F>, gemm8_bf16_kernel<1, 1, F>, gemm8_bf16_kernel<1, 0, F>, gemm8_bf16_kernel<1, 0, F>, gemm8_bf16_kernel<1, 0, F>}, \
      {gemm8_bf16_kernel<2, 0, F>, gemm8_bf16_kernel<2, 0, F>, gemm8_bf16_kernel<2, 2, F>, gemm8_bf16_kernel<2, 0, F>, gemm8_bf16_kernel<2, 4, F>}, \
      {gemm8_bf16_kernel<3, 0, F>, gemm8_bf16_kernel<3, 0, F>, gemm8_bf16_kernel<3, 0, F>, gemm8_bf16_kernel<3, 3, F>, gemm8_bf16_kernel<3, 0, F>}, \
      {gemm8_bf16_kernel<4, 0, F>, gemm8_bf16_kernel<4, 0, F>, gemm8_bf16_kernel<4, 0, F>, gemm8_bf16_kernel<4, 0, F>, gemm8_bf16_kernel<4, 0, F>}, \
      {gemm8_bf16_kernel<5, 0, F>, gemm8_bf16_kernel<5, 0, F>, gemm8_bf16_kernel<5, 0, F>, gemm8_bf16_kernel<5, 0, F>, gemm8_bf16_kernel<5, 0, F>}, \
      {gemm8_bf16_kernel<6, 0, F>, gemm8_bf16_kernel<6, 1, F>, gemm8_bf16_kernel<6, 0, F>, gemm8_bf16_kernel<6, 0, F>, gemm8_bf16_kernel<6, 0, F>}, \
      {gemm8_bf16_kernel<7, 0, F>, gemm8_bf16_kernel<7, 0, F>, gemm8_bf16_kernel<7, 2, F>, gemm8_bf16_kernel<7, 0, F>, gemm8_bf16_kernel<7, 4, F>}, \
      {gemm8_bf16_kernel<8, 3, false>, gemm8_bf16_kernel<8, 3, false>, gemm8_bf16_kernel<8, 3, false>, gemm8_bf16_kernel<8, 3, false>, gemm8_bf16_kernel<8, 3, false>}
  static const kern_t table[2][9][5] = {{G8_ROW(false)}, {G8_ROW(true)}};
#undef G8_ROW
  if (!ds.attr_done) {
    for (int f = 0; f < 2; ++f)
      for (int k = 0; k < 9; ++k)
        for (int t = 0; t < 5; ++t)
          EC_HIP(hipFuncSetAttribute((const void*)table[f][k][t], hipFuncAttributeMaxDynamicSharedMemorySize, G8_LDS));
    EC_HIP(hipMalloc((void**)&ds.zeros, 16384 * sizeof(float)));
    EC_HIP(hipMemset(ds.zeros, 0, 16384 * sizeof(float)));
    EC_HIP(hipDeviceGetAttribute(&ds.ncu, hipDeviceAttributeMultiprocessorCount, dev));
    ds.attr_done = true;
  }
  const long ntiles = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
  long grid = ds.ncu;
  if (ntiles < grid) grid = ntiles;
  GemmP q = p;
  if ((kind == G8_TAB_H16 || kind == G8_TAB_F32) && !q.bias) {
    if (p.N > 16384) kind = G8_GENERIC; else q.bias = ds.zeros;
  }
  if (ntiles <= grid || grid < 8 || p.K < 256 || (kind != G8_BIAS_BF16 && kind != G8_GELU_BF16)) q.sched = nullptr;   // one tile per workgroup: nothing to deal out (K >= 256: the hand-over spans three K-tiles)
  // (round 5, measured and removed: the multi-round GEMMs of a PIPELINED call on ncu - R workgroups, so that R CUs stay free for the
  //  previous call's head instead of the head holding CUs the persistent workgroups want: R = 8 nothing, R = 16 / 32 -3 % pairs/s, QKV
  //  0.349 -> 0.358 / 0.328 / 0.322 of peak beside the head; profiles/r05_cu_reserve_ab.txt.  The dynamic tile schedule already absorbs
  //  late workgroups; fewer workgroups only lose their tiles' worth of CUs.)
  hipLaunchKernelGGL(table[p.h_f16 ? 1 : 0][kind][p.tag], dim3((unsigned)grid), dim3(512), G8_LDS, st, q);
  EC_LAUNCH_CHECK();
  return 1;
}

#ifdef EC_G8_LAB
// Lab build only (tools/g8_lab.py, libedgecape_hip_lab.so): time an ablated instantiation of the QKV-kind kernel.
extern "C" int ec_lab_gemm8(const void* A, const void* W, const float* bias, void* C, int M, int N, int K, int lab, int iters,
                            void* stream, float* ms) {
  typedef void (*kern_t)(GemmP);
  kern_t k = nullptr;
  switch (lab) {
    case 0: k = gemm8_bf16_kernel<1, 1, false, 0>; break;
    case 1: k = gemm8_bf16_kernel<1, 1, false, 1>; break;
    case 2: k = gemm8_bf16_kernel<1, 1, false, 2>; break;
    case 3: k = gemm8_bf16_kernel<1, 1, false, 3>; break;
    case 4: k = gemm8_bf16_kernel<1, 1, false, 4>; break;
    case 5: k = gemm8_bf16_kernel<1, 1, false, 5>; break;
    case 8: k = gemm8_bf16_kernel<1, 1, false, 8>; break;
    case 9: k = gemm8_bf16_kernel<1, 1, false, 9>; break;
    case 32: k = gemm8_bf16_kernel<1, 1, false, 32>; break;
    case 34: k = gemm8_bf16_kernel<1, 1, false, 34>; break;
    case 36: k = gemm8_bf16_kernel<1, 1, false, 36>; break;
    case 40: k = gemm8_bf16_kernel<1, 1, false, 40>; break;
    case 64: k = gemm8_bf16_kernel<1, 1, false, 64>; break;
    case 256: k = gemm8_bf16_kernel<1, 1, false, 256>; break;   // round-1 DMA pieces (16 rows x 64 B)
    case 356: k = gemm8_bf16_kernel<3, 3, false, 256>; break;
    case 128: k = gemm8_bf16_kernel<1, 1, false, 128>; break;   // epilogue through the LDS staging slot (round-1 form)
    case 129: k = gemm8_bf16_kernel<1, 1, false, 129>; break;
    case 228: k = gemm8_bf16_kernel<3, 3, false, 128>; break;   // fc1 + GELU, staged epilogue
    case 65: k = gemm8_bf16_kernel<1, 1, false, 65>; break;
    case 100: k = gemm8_bf16_kernel<3, 3, false, 0>; break;     // fc1 + GELU, pipelined
    case 164: k = gemm8_bf16_kernel<3, 3, false, 64>; break;    // fc1 + GELU, epilogue at the end of the tile
    case 1001: k = gemm8_bf16_kernel<1, 1, true, 1>; break;     // fp16 qkv kind: no global stores
    case 1008: k = gemm8_bf16_kernel<1, 1, true, 8>; break;     // ... no MFMAs
    case 2024: k = gemm8_bf16_kernel<1, 1, true, 1024>; break;  // ... stores aliased onto tile row 0 (L2-resident)
    case 1004: k = gemm8_bf16_kernel<1, 1, true, 4>; break;     // ... no LDS-DMA in the steady state
    case 9192: k = gemm8_bf16_kernel<1, 1, true, 8192>; break;  // ... static priority for the younger wave group
    case 9384: k = gemm8_bf16_kernel<1, 1, true, 16384>; break; // ... no priorities
    case 5096: k = gemm8_bf16_kernel<1, 1, true, 4096>; break;  // ... the prologue waits for all five half-tiles (round-3 form)
    case 6096: k = gemm8_bf16_kernel<2, 4, true, 4096>; break;  // LayerScale kind, same
    case 7096: k = gemm8_bf16_kernel<3, 3, true, 4096>; break;  // GELU kind, same
    case 3128: k = gemm8_bf16_kernel<3, 3, true, 128>; break;   // fp16 fc1 + GELU through the LDS staging slot (whole-line stores)
    case 3048: k = gemm8_bf16_kernel<1, 1, true, 2048>; break;  // ... no fragment reads in the steady state
    case 3052: k = gemm8_bf16_kernel<1, 1, true, 2052>; break;  // ... neither
    case 1000: k = gemm8_bf16_kernel<1, 1, true, 0>; break;     // the shipped fp16 instantiations: qkv / proj (bias)
    case 2000: k = gemm8_bf16_kernel<2, 4, true, 0>; break;     // fc2 / proj with LayerScale (gamma = the bias vector here)
    case 3000: k = gemm8_bf16_kernel<3, 3, true, 0>; break;     // fc1 + GELU
    default: set_error("ec_lab_gemm8: variant not instantiated"); return -1;
  }
  hipStream_t st = (hipStream_t)stream;
  EC_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, G8_LDS));
  GemmP p;
  p.A = A; p.B = W; p.C = C; p.bias = bias; p.M = M; p.N = N; p.K = K; p.lda = K; p.ldb = K; p.ldc = N; p.ab_bf16 = 1; p.c_bf16 = 1;
  p.gamma = bias;
  int dev = 0, ncu = 0;
  EC_HIP(hipGetDevice(&dev));
  EC_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
  const long ntiles = (long)((M + 255) / 256) * ((N + 255) / 256);
  unsigned grid = (unsigned)(ntiles < ncu ? ntiles : ncu);
  if (getenv("EC_G8_GRID") && atoi(getenv("EC_G8_GRID")) > 0 && (unsigned)atoi(getenv("EC_G8_GRID")) < grid) grid = (unsigned)atoi(getenv("EC_G8_GRID"));
  hipEvent_t e0, e1;
  EC_HIP(hipEventCreate(&e0));
  EC_HIP(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(grid), dim3(512), G8_LDS, st, p);
  EC_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(512), G8_LDS, st, p);
  EC_HIP(hipEventRecord(e1, st));
  EC_HIP(hipEventSynchronize(e1));
  float t = 0.f;
  EC_HIP(hipEventElapsedTime(&t, e0, e1));
  *ms = t / (float)iters;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return 0;
}

// Lab build only (tools/g8_ledger.py): ONE launch of the shipped QKV-kind fp16 kernel with s_memtime stamps (LAB 512) after two warm
// launches; trace: device buffer of grid x 64 uint64 (per workgroup: [wave group][32 stamps], see the LAB comment).
extern "C" int ec_lab_gemm8_trace(const void* A, const void* W, const float* bias, void* C, int M, int N, int K, int kind, void* trace, void* stream,
                                  float* traced_ms) {
  typedef void (*kern_t)(GemmP);
  kern_t k = kind == 3 ? (kern_t)gemm8_bf16_kernel<3, 3, true, 512> : kind == 2 ? (kern_t)gemm8_bf16_kernel<2, 4, true, 512> : (kern_t)gemm8_bf16_kernel<1, 1, true, 512>;
  kern_t k0 = kind == 3 ? (kern_t)gemm8_bf16_kernel<3, 3, true, 0> : kind == 2 ? (kern_t)gemm8_bf16_kernel<2, 4, true, 0> : (kern_t)gemm8_bf16_kernel<1, 1, true, 0>;
  hipStream_t st = (hipStream_t)stream;
  EC_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, G8_LDS_LAB));
  EC_HIP(hipFuncSetAttribute((const void*)k0, hipFuncAttributeMaxDynamicSharedMemorySize, G8_LDS));
  GemmP p;
  p.A = A; p.B = W; p.C = C; p.bias = bias; p.M = M; p.N = N; p.K = K; p.lda = K; p.ldb = K; p.ldc = N; p.ab_bf16 = 1; p.c_bf16 = 1; p.h_f16 = 1;
  p.gamma = bias;
  int dev = 0, ncu = 0;
  EC_HIP(hipGetDevice(&dev));
  EC_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
  const long ntiles = (long)((M + 255) / 256) * ((N + 255) / 256);
  const unsigned grid = (unsigned)(ntiles < ncu ? ntiles : ncu);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k0, dim3(grid), dim3(512), G8_LDS, st, p);   // (warm clocks: the stamps are read against the launch's wall time)
  p.aux = (const float*)trace;
  hipEvent_t e0, e1;
  EC_HIP(hipEventCreate(&e0));
  EC_HIP(hipEventCreate(&e1));
  EC_HIP(hipEventRecord(e0, st));
  hipLaunchKernelGGL(k, dim3(grid), dim3(512), G8_LDS_LAB, st, p);
  EC_HIP(hipEventRecord(e1, st));
  EC_LAUNCH_CHECK();
  EC_HIP(hipEventSynchronize(e1));
  if (traced_ms) EC_HIP(hipEventElapsedTime(traced_ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return 0;
}
#endif

}  // namespace ec
