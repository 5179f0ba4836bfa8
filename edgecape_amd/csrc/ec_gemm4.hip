// 16-bit NT GEMM for the backbone of the EdgeCape hot path on gfx950 (MI355X), one wave per SIMD: the north-star kernel, round 2.
//
//   C[M,N] = epilogue(A[M,K] @ B[N,K]^T)      A = activations (bf16 / fp16, K contiguous), B = nn.Linear weight (same format)
//
// Reference ops: B4 `qkv` Linear (the roofline kernel), B6 `proj`, B7 `fc1`/`fc2` of every DINOv2 block
// (facebookresearch/dinov2 Attention / Mlp, called from EdgeCape/models/detectors/EdgeCape.py:188-189).
//
// Why a second kernel beside ec_gemm8.hip: the 8-wave / 8-phase kernel alternates two wave groups per SIMD between an MFMA segment
// and a memory segment; its K loop runs at 0.59 of the MFMA rate and its epilogue (a fifth of a K = 768 tile) is not hidden at all
// (DESIGN.md §4).  Here a workgroup is FOUR waves, one per SIMD, each owning a 128 x 128 quarter of the 256 x 256 output tile and
// the whole 512-register file of its SIMD:
//   * 256 accumulator registers (16 x v_mfma_f32_32x32x16 tiles), two fragment sets (64) and a 128-register STASH that holds the
//     previous tile's finished, packed 16-bit output;
//   * ONE instruction stream per SIMD: per K-step of 16, eight ds_read_b128 and sixteen MFMAs, with the LDS-DMA pieces of the next
//     K-tile and the stash's global stores dealt out between the MFMAs - the stores of tile t leave under the K loop of tile t+1,
//     spread over the whole tile (the chip's CUs run in lockstep: a burst of 32 MB at every tile seam is what made the old epilogue
//     slow), and nothing but the register-only conversion pass (bias / LayerScale / GELU, pack, lane swaps) sits between two tiles;
//   * one s_barrier per K-tile (64), LDS ring of two K-tiles (2 x 64 KiB), counted vmcnt only.
// LDS image of one operand of a K-tile (256 rows x 128 B): [16-row group][k half] sub-tiles of [16 rows][64 B] = 1 KiB, the 16-byte
// chunk index XORed with (row >> 2) & 3: a 32-row x 16-k fragment read (lane -> row lane & 31, chunk lane >> 5) is conflict-free
// in all four lane groups of ds_read_b128.  LDS-DMA writes lane-linear, so the swizzle is applied to each lane's SOURCE address.
// MFMA roles are swapped (A-operand <- weight rows n, B-operand <- activation rows m): a lane holds ONE output row and, per
// 32 x 32 tile, four runs of four consecutive columns; two v_permlane32_swap per pair of runs give every lane 8 consecutive
// columns = one 16-byte store, 32 rows x 32 B per wave-instruction, eight consecutive instructions covering 32 rows x 256 B.
#include <stdlib.h>

#include "ec_common.h"

#ifndef EC_G8_LAB
// Round-2 status: correct (tools/g8_lab.py checks it against torch, guard rows included) but 8 % SLOWER than the 8-phase kernel on the
// QKV shape (68 vs 63 us; DESIGN.md section 4 has the cycle anatomy: with one wave per SIMD every global store blocks the instruction stream
// for ~125 cycles and every LDS-DMA piece for ~30, which two waves per SIMD hide behind each other) - so it is compiled into the kernel-lab
// library only and the shipped library never dispatches to it.
namespace ec {
int gemm4_h16(const GemmP&, hipStream_t) { return 0; }
}  // namespace ec
#else

namespace ec {
namespace {

typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int G4_OP = 32768;               // one operand of a K-tile: 256 rows x 128 B
constexpr int G4_KT = 2 * G4_OP;           // K-tile: X (activations) | W (weights)
constexpr int G4_BIAS = 2 * G4_KT;         // fp32 bias[N], N <= 4096
constexpr int G4_GAMMA = G4_BIAS + 16384;  // fp32 LayerScale[N]
constexpr int G4_LDS = G4_GAMMA + 16384;   // 160 KiB
constexpr int G4_NST = 32;                 // global stores of one tile per wave (128 x 128 x 2 B / 1 KiB)

enum { G4_BIAS_H = 1, G4_SCALE_H = 2, G4_GELU_H = 3 };   // epilogue kinds (16-bit output): + bias | + bias, * LayerScale | GELU(+ bias)

// GELU for 16-bit outputs (same polynomial as ec_gemm8.hip: x * Phi(x), Phi(x) = 1 / (1 + 2^(x * P(x^2))))
template <bool F16>
__device__ __forceinline__ float g4_gelu(float x) {
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

__device__ __forceinline__ int g4_lane_now() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
template <int N> __device__ __forceinline__ void g4_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// HEAD: the stash's 32 stores leave in the first HEAD K-tiles of the next output tile, dealt out evenly (HEAD odd, < K / 64); those
// K-tile bodies are unrolled so that every stash register index is static.
// LAB (0 in the shipped library): 1 no global stores, 4 no LDS-DMA in the steady state, 8 no MFMAs, 16 no barriers, 32 no conversion
// pass, 64 pieces rotated by wave.
template <int KIND, int TAG, bool F16, int HEAD, int LAB = 0>
__global__ __launch_bounds__(256) void gemm4_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  const int ntm = (p.M + 255) >> 8, ntn = (p.N + 255) >> 8;
  const int ntiles = ntm * ntn;
  const int nk = p.K >> 6;                        // K-tiles per output tile (even, > HEAD: checked on the host)
  const unsigned lda_b = (unsigned)p.lda * 2u, ldb_b = (unsigned)p.ldb * 2u, ldc_b = (unsigned)p.ldc * 2u;

  // XCD-aware persistent schedule (as ec_gemm8.hip): workgroup b runs on XCD b % 8 and walks a contiguous tile range of it.
  const int nxcd = (gridDim.x >= 8 && gridDim.x % 8 == 0) ? 8 : 1;
  const int chunk = (ntiles + nxcd - 1) / nxcd;
  const int xcd = blockIdx.x % nxcd, slot = blockIdx.x / nxcd, nslot = gridDim.x / nxcd;
  const int t_end = min(ntiles, (xcd + 1) * chunk);
  const int t_first = xcd * chunk + slot;
  if (t_first >= t_end) return;

  // ---- bias / LayerScale vectors -> LDS, once per workgroup (read by the conversion pass behind many barriers)
  for (int i = tid * 4; i < p.N; i += 1024) {
    *(f32x4*)(smem + G4_BIAS + i * 4) = *(const f32x4*)(p.bias + i);
    if constexpr (KIND == G4_SCALE_H) *(f32x4*)(smem + G4_GAMMA + i * 4) = *(const f32x4*)(p.gamma + i);
  }

  // ---- load stream (LDS-DMA): wave w stages the 16-row groups 4w .. 4w+3 of both operands, both k halves: 16 pieces of 1 KiB per
  // K-tile.  lane -> row lane >> 2 of the group, physical chunk lane & 3 holding logical chunk (lane & 3) ^ ((row >> 2) & 3).
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, -1, 0x00020000);
  unsigned vx[4], vw[4];
  int ls_kt = 0, ls_tile = t_first;
  auto set_rows = [&](int t) {
    const int m0 = (t / ntn) << 8, n0 = (t % ntn) << 8;
    const int l = g4_lane_now();
    const int r = l >> 2, c = ((l & 3) ^ ((r >> 2) & 3)) << 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      vx[i] = (unsigned)min(m0 + (wave * 4 + i) * 16 + r, p.M - 1) * lda_b + (unsigned)c;   // rows past the edge: clamped, never stored
      vw[i] = (unsigned)min(n0 + (wave * 4 + i) * 16 + r, p.N - 1) * ldb_b + (unsigned)c;
    }
  };
  bool steady = false;
  // piece j of the K-tile the stream stands at, into ring slot sl: operand j >> 3, row group (j >> 1) & 3, k half j & 1
  auto piece = [&](int j, int sl) {
    if constexpr (LAB & 4) { if (steady) return; }
    const int op = j >> 3, i = (j >> 1) & 3, kh = j & 1;
    char* dst = smem + sl * G4_KT + op * G4_OP + ((wave * 4 + i) * 2 + kh) * 1024;
    if (op) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lptr_t)dst, 16, vw[i], ls_kt * 128 + kh * 64, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lptr_t)dst, 16, vx[i], ls_kt * 128 + kh * 64, 0, 0);
  };
  auto advance = [&]() {
    if (++ls_kt == nk) {
      ls_kt = 0;
      ls_tile += nslot;
      if (ls_tile < t_end) set_rows(ls_tile);   // past the last tile the stream re-reads the same rows (never consumed)
    }
  };
  set_rows(t_first);
#pragma unroll
  for (int j = 0; j < 16; ++j) piece(j, 0);
  steady = true;

  // ---- fragment read addresses: lane -> row lane & 31 of a 32-row block, logical chunk 2 * (k step & 1) + (lane >> 5)
  const int fr = lane & 31, frr = fr & 15, fsw = (frr >> 2) & 3;
  const int off_e = (fr >> 4) * 2048 + frr * 64 + ((((lane >> 5)) ^ fsw) << 4);
  const char* const xb = smem + wr * 16384 + off_e;              // X rows wr*128.., even k steps; odd: ^ 32
  const char* const wb = smem + G4_OP + wc * 16384 + off_e;      // W rows wc*128..
  const char* const xbo = smem + wr * 16384 + (off_e ^ 32);
  const char* const wbo = smem + G4_OP + wc * 16384 + (off_e ^ 32);

  f32x16 acc[4][4];        // [ni][mi]: 32 columns x 32 rows
  bf16x8 xf[2][4], wf[2][4];
  u32x4 stash[G4_NST];     // previous tile, packed: store s = (mi, ni, pair) = (s >> 3, (s >> 1) & 3, s & 1)
#pragma unroll
  for (int s = 0; s < G4_NST; ++s) stash[s] = u32x4{0u, 0u, 0u, 0u};
  // stores of the stash: buffer stores, row (lane & 31) of a 32-row block, 16 B at column half (lane >> 5); rows past M fall outside
  // the descriptor's range and are dropped by the hardware, the first tile's (empty) stash is stored out of range on purpose so
  // that the vmcnt arithmetic is the same for every tile
  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)((unsigned)p.M * ldc_b), 0x00020000);
  unsigned st_vo = 0xC0000000u;   // per-lane offset of the stash's stores (out of range until a tile has been converted)
  unsigned st_base = 0;           // byte offset of the stashed tile's (row m0 + wr*128, column n0 + wc*128)
  bool st_cols = false;           // the stashed wave quarter lies inside N
  auto stash_store = [&](int s) {
    if constexpr (LAB & 1) { asm volatile("" ::"v"(stash[s])); return; }
    const int mi = s >> 3, ni = (s >> 1) & 3, pr = s & 1;
    __builtin_amdgcn_raw_buffer_store_b128(stash[s], rsC, st_vo, st_base + (unsigned)(mi * 32) * ldc_b + (unsigned)(ni * 64 + pr * 32), 0);
  };

  // fragment i of K step ks of the K-tile in ring slot sl -> fragment buffer fb; i = 0..3: activation row blocks, 4..7: weight row blocks
  auto rd = [&](int sl, int ks, int fb, int i) {
    const char* b = (i < 4 ? ((ks & 1) ? xbo : xb) : ((ks & 1) ? wbo : wb)) + sl * G4_KT + (ks >> 1) * 1024 + (i & 3) * 4096;
    if (i < 4) xf[fb][i] = *(const bf16x8*)b;
    else wf[fb][i - 4] = *(const bf16x8*)b;
  };
  // the four MFMAs of column block ni (one "chunk": 128 matrix-pipe cycles); j0 >= 0: this chunk also issues pieces j0, j0 + 1 of the next
  // K-tile into slot sl - LAB & 64: behind MFMA number `wave` of the chunk, so that the four waves of the workgroup (which run in lockstep)
  // do not hand their pieces to the CU's one texture-address unit in the same cycle
  auto mf4 = [&](int fb, int ni, bool zero, int j0 = -1, int sl = 0) {
    if (j0 >= 0 && !(LAB & 64)) { piece(j0, sl); piece(j0 + 1, sl); }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      if constexpr (LAB & 8) {
        asm volatile("" ::"v"(wf[fb][ni]), "v"(xf[fb][mi]));
      } else {
        const f32x16 c0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc[ni][mi] = mfma32x32x16_h<F16>(wf[fb][ni], xf[fb][mi], zero ? c0 : acc[ni][mi]);
      }
      if constexpr ((LAB & 64) != 0) {
        if (j0 >= 0 && wave == mi) { piece(j0, sl); piece(j0 + 1, sl); }
      }
    }
  };

  // One K-tile (ring slot SL) = four K steps of four chunks; a chunk is 4 MFMAs plus the memory instructions dealt to it, pinned by
  // sched_barrier so that the matrix pipe never waits behind a burst of them:
  //   steps 0, 1: the fragment reads of the next step (2 per chunk) and the 16 LDS-DMA pieces of the NEXT K-tile (2 per chunk, into the
  //               other slot: every wave finished reading it before the barrier it passed in the previous body);
  //   steps 2, 3: fragment reads of step 3, the NS stash stores from index S0 (at most one per chunk);
  //   before the last chunk: wait for the next K-tile's pieces (this wave's: everything it issued since, the NS stores, may stay in
  //               flight), barrier (everyone's pieces landed), all eight fragment reads of the next K-tile's step 0.
  // On entry the step-0 fragments of this K-tile are in flight into fragment buffer 0.
#define G4_SB() __builtin_amdgcn_sched_barrier(0)
#define G4_BODY(SL, S0, NS, ZERO)                                                                     \
  do {                                                                                                \
    advance();                                                                                        \
    _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                   \
      rd(SL, 1, 1, 2 * c); rd(SL, 1, 1, 2 * c + 1);                                                   \
      mf4(0, c, ZERO, 2 * c, (SL) ^ 1);                                                               \
      G4_SB();                                                                                        \
    }                                                                                                 \
    _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                   \
      rd(SL, 2, 0, 2 * c); rd(SL, 2, 0, 2 * c + 1);                                                   \
      mf4(1, c, false, 8 + 2 * c, (SL) ^ 1);                                                          \
      G4_SB();                                                                                        \
    }                                                                                                 \
    _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                   \
      rd(SL, 3, 1, 2 * c); rd(SL, 3, 1, 2 * c + 1);                                                   \
      _Pragma("unroll") for (int s = 0; s < (NS); ++s) if ((s * 7) / ((NS) ? (NS) : 1) == c) stash_store((S0) + s);   \
      mf4(0, c, false);                                                                               \
      G4_SB();                                                                                        \
    }                                                                                                 \
    _Pragma("unroll") for (int c = 0; c < 3; ++c) {                                                   \
      _Pragma("unroll") for (int s = 0; s < (NS); ++s) if ((s * 7) / ((NS) ? (NS) : 1) == 4 + c) stash_store((S0) + s);   \
      mf4(1, c, false);                                                                               \
      G4_SB();                                                                                        \
    }                                                                                                 \
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NS) : "memory");                              \
    if constexpr (!(LAB & 16)) __builtin_amdgcn_s_barrier();                                          \
    G4_SB();                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) rd((SL) ^ 1, 0, 0, i);                              \
    mf4(1, 3, false);                                                                                 \
    G4_SB();                                                                                          \
  } while (0)

  // head body j stores stash entries [g4_s0(j), g4_s0(j + 1))
#define G4_S0(J) (((J) * G4_NST) / HEAD)
#define G4_NS(J) (G4_S0((J) + 1) - G4_S0(J))
  static_assert((HEAD & 1) && HEAD >= 1 && HEAD <= 17, "stash schedule");

  // first K-tile: landed (this wave's pieces), barrier (everyone's; also orders the bias vectors), its step-0 fragments
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) rd(0, 0, 0, i);

  for (int t = t_first; t < t_end; t += nslot) {
    const int m0 = (t / ntn) << 8, n0 = (t % ntn) << 8;
    // ---- head: static bodies (stash indices), slot = body index & 1
    G4_BODY(0, 0, G4_NS(0), true);
#define G4_HB(J)                                                                                             \
    if constexpr (HEAD > (J)) G4_BODY((J) & 1, G4_S0(J), G4_NS(J), false);
    G4_HB(1) G4_HB(2) G4_HB(3) G4_HB(4) G4_HB(5) G4_HB(6) G4_HB(7) G4_HB(8) G4_HB(9) G4_HB(10) G4_HB(11) G4_HB(12) G4_HB(13)
    G4_HB(14) G4_HB(15) G4_HB(16)
#undef G4_HB
    // ---- tail: no stores
    G4_BODY(1, 0, 0, false);
    for (int kt = HEAD + 1; kt < nk; kt += 2) {   // (HEAD is odd and nk even: pairs of slot 0, slot 1)
      G4_BODY(0, 0, 0, false);
      G4_BODY(1, 0, 0, false);
    }
    // ---- conversion pass: accumulators -> packed 16-bit output in the stash
    if constexpr (LAB & 32) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("" ::"v"(acc[i >> 2][i & 3]));
    } else {
      const int l = g4_lane_now();
      const char* bl = smem + G4_BIAS + (n0 + wc * 128) * 4 + (l >> 5) * 16;
      const char* gl = smem + G4_GAMMA + (n0 + wc * 128) * 4 + (l >> 5) * 16;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          u32x2 pk[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4 v = {acc[ni][mi][4 * g], acc[ni][mi][4 * g + 1], acc[ni][mi][4 * g + 2], acc[ni][mi][4 * g + 3]};
            v += *(const f32x4*)(bl + ni * 128 + g * 32);
            if constexpr (KIND == G4_GELU_H) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = g4_gelu<F16>(v[e]);
            }
            if constexpr (KIND == G4_SCALE_H) v *= *(const f32x4*)(gl + ni * 128 + g * 32);
            pk[g] = pack4_h<F16>(v);
          }
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            const u32x2 s0 = __builtin_amdgcn_permlane32_swap(pk[2 * pr][0], pk[2 * pr + 1][0], false, false);
            const u32x2 s1 = __builtin_amdgcn_permlane32_swap(pk[2 * pr][1], pk[2 * pr + 1][1], false, false);
            stash[(mi * 4 + ni) * 2 + pr] = u32x4{s0[0], s1[0], s0[1], s1[1]};
          }
        }
      st_cols = n0 + wc * 128 < p.N;
      st_vo = st_cols ? (unsigned)(l & 31) * ldc_b + (unsigned)(l >> 5) * 16u : 0xC0000000u;
      st_base = (unsigned)(m0 + wr * 128) * ldc_b + (unsigned)(n0 + wc * 128) * 2u;
    }
  }
#undef G4_BODY
#undef G4_S0
#undef G4_NS
  // ---- the last tile's stash, then everything this wave has in flight (stores and the stream's surplus pieces)
#pragma unroll
  for (int s = 0; s < G4_NST; ++s) stash_store(s);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace

namespace {
struct G4Dev { bool attr_done = false; int ncu = 0; };
G4Dev g4_dev[64];
typedef void (*g4_kern_t)(GemmP);

template <int KIND, int TAG, bool F16>
g4_kern_t g4_pick(int nk) {
  if (nk >= 18) return gemm4_kernel<KIND, TAG, F16, 17>;
  if (nk >= 12) return gemm4_kernel<KIND, TAG, F16, 11>;
  if (nk >= 10) return gemm4_kernel<KIND, TAG, F16, 9>;
  return gemm4_kernel<KIND, TAG, F16, 5>;
}
template <bool F16>
g4_kern_t g4_pick_kind(int kind, int tag, int nk) {
  switch (kind) {
    case G4_BIAS_H: return tag == 1 ? g4_pick<G4_BIAS_H, 1, F16>(nk) : g4_pick<G4_BIAS_H, 0, F16>(nk);
    case G4_SCALE_H: return tag == 2 ? g4_pick<G4_SCALE_H, 2, F16>(nk) : tag == 4 ? g4_pick<G4_SCALE_H, 4, F16>(nk) : g4_pick<G4_SCALE_H, 0, F16>(nk);
    default: return tag == 3 ? g4_pick<G4_GELU_H, 3, F16>(nk) : g4_pick<G4_GELU_H, 0, F16>(nk);
  }
}
}  // namespace

// Returns 1 if this kernel handled the problem, 0 if the shape / epilogue is not eligible (caller falls back), < 0 on error.
int gemm4_h16(const GemmP& p, hipStream_t st) {
  static const int enable = getenv("EC_GEMM4") ? atoi(getenv("EC_GEMM4")) : 0;   // lab library: opt-in through gemm_nt
  if (!enable) return 0;
  if (!p.ab_bf16 || !p.c_bf16 || p.batch != 1 || !p.bias || p.resid || p.table) return 0;
  int kind = 0;
  if (p.act == ACT_NONE && !p.gamma) kind = G4_BIAS_H;
  else if (p.act == ACT_NONE && p.gamma) kind = G4_SCALE_H;
  else if (p.act == ACT_GELU && !p.gamma) kind = G4_GELU_H;
  if (!kind) return 0;
  const int nk = p.K >> 6;
  if (p.K % 128 != 0 || nk < 6 || p.N % 128 != 0 || p.N > 4096 || p.M < 1024 || p.N < 256 || p.ldc % 8 != 0) return 0;
  if ((long)p.M * p.lda * 2 >= (1l << 32) - (1l << 20) || (long)p.N * p.ldb * 2 >= (1l << 32) - (1l << 20)) return 0;   // 32-bit buffer offsets
  if ((long)(p.M + 256) * p.ldc * 2 >= (1l << 30)) return 0;                                                            // store offsets stay below the out-of-range marker
  const g4_kern_t k = p.h_f16 ? g4_pick_kind<true>(kind, p.tag, nk) : g4_pick_kind<false>(kind, p.tag, nk);
  int dev = 0;
  EC_HIP(hipGetDevice(&dev));
  EC_REQUIRE(dev >= 0 && dev < 64, -1, "gemm4: device ordinal out of range");
  G4Dev& ds = g4_dev[dev];
  if (!ds.attr_done) {
    EC_HIP(hipDeviceGetAttribute(&ds.ncu, hipDeviceAttributeMultiprocessorCount, dev));
    ds.attr_done = true;
  }
  // (the attribute is per function and per device; setting it is cheap next to a launch of this size)
  EC_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS));
  const long ntiles = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
  long grid = ds.ncu;
  if (ntiles < grid) grid = ntiles;
  hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(256), G4_LDS, st, p);
  EC_LAUNCH_CHECK();
  return 1;
}

// Lab build only (tools/g8_lab.py): time (ablated) instantiations of the four-wave kernel, QKV-kind epilogue unless noted.
extern "C" int ec_lab_gemm4(const void* A, const void* W, const float* bias, void* C, int M, int N, int K, int lab, int iters,
                            void* stream, float* ms) {
  g4_kern_t k = nullptr;
  const int nk = K >> 6;
  switch (lab) {
    case 0: k = g4_pick<G4_BIAS_H, 1, false>(nk); break;
    case 1: k = gemm4_kernel<G4_BIAS_H, 1, false, 11, 1>; break;
    case 4: k = gemm4_kernel<G4_BIAS_H, 1, false, 11, 4>; break;
    case 5: k = gemm4_kernel<G4_BIAS_H, 1, false, 11, 5>; break;
    case 16: k = gemm4_kernel<G4_BIAS_H, 1, false, 11, 16>; break;
    case 32: k = gemm4_kernel<G4_BIAS_H, 1, false, 11, 32>; break;
    case 48: k = gemm4_kernel<G4_BIAS_H, 1, false, 11, 48>; break;
    case 64: k = gemm4_kernel<G4_BIAS_H, 1, false, 11, 64>; break;
    case 21: k = gemm4_kernel<G4_BIAS_H, 1, false, 11, 21>; break;
    case 53: k = gemm4_kernel<G4_BIAS_H, 1, false, 11, 53>; break;
    case 8: k = gemm4_kernel<G4_BIAS_H, 1, false, 11, 8>; break;
    case 100: k = g4_pick<G4_GELU_H, 3, false>(nk); break;
    case 200: k = g4_pick<G4_SCALE_H, 4, false>(nk); break;
    default: set_error("ec_lab_gemm4: variant not instantiated"); return -1;
  }
  hipStream_t st = (hipStream_t)stream;
  EC_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS));
  GemmP p;
  p.A = A; p.B = W; p.C = C; p.bias = bias; p.gamma = bias; p.M = M; p.N = N; p.K = K; p.lda = K; p.ldb = K; p.ldc = N; p.ab_bf16 = 1; p.c_bf16 = 1;
  int dev = 0, ncu = 0;
  EC_HIP(hipGetDevice(&dev));
  EC_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
  const long ntiles = (long)((M + 255) / 256) * ((N + 255) / 256);
  const unsigned grid = (unsigned)(ntiles < ncu ? ntiles : ncu);
  hipEvent_t e0, e1;
  EC_HIP(hipEventCreate(&e0));
  EC_HIP(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), G4_LDS, st, p);
  EC_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), G4_LDS, st, p);
  EC_HIP(hipEventRecord(e1, st));
  EC_HIP(hipEventSynchronize(e1));
  float t = 0.f;
  EC_HIP(hipEventElapsedTime(&t, e0, e1));
  *ms = t / (float)iters;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return 0;
}

}  // namespace ec
#endif  // EC_G8_LAB
