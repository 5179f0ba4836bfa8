// Row-chain kernel (ec_chain.hip): a sequence of Linear (+ bias / table / activation / residual / LayerNorm) stages applied to
// 32-row slabs, intermediate activations resident in LDS.  Internal.
#pragma once
#include "ec_common.h"

namespace ec {

constexpr int CH_BM = 32;          // rows per workgroup
constexpr int CH_MAX_STAGES = 5;
constexpr int CH_LDS0 = 4096;      // first byte of the activation buffers (the LayerNorm scratch sits in front)

// One stage: Y[32, N] = epilogue(X[32, K] @ W[N, K]^T).
//   X: k columns [0, k1) from the LDS operand buffer at a_off, [k1, K) from the one at b_off (row pitch = columns*4 + 16 bytes;
//      split hi/lo bf16 planes).  An operand buffer is filled either by an earlier stage (s_off) or, right before this stage, from
//      global memory (g_in: fp32 [rows, g_k] -> buffer at g_off).
//   epilogue: v = acc + bias[n] + table[row % period][n]; v = act(v); v += resid[row][n]; v += kept tile; v = LayerNorm(v)
//      (N = 256 only); v += post_table[row % post_period][n]; then any of: global fp32 store, split write into the LDS buffer at s_off, register copy (keep) for a later
//      stage's residual.
struct ChainStage {
  const void* W = nullptr;       // pack_chain_weights image of W [N, K]
  const float* bias = nullptr;
  int N = 0, K = 0;
  int a_off = 0, k1 = 0, b_off = 0;
  const float* g_in = nullptr; long ld_in = 0; int g_k = 0; int g_off = 0;
  int act = ACT_NONE;
  const float* table = nullptr; long ldt = 0; int period = 1;
  const float* resid = nullptr; long ldr = 0;
  int resid_keep = 0;
  const float* ln_w = nullptr; const float* ln_b = nullptr; float eps = 1e-5f;
  const float* post_table = nullptr; long ldpt = 0; int post_period = 1;   // added AFTER the LayerNorm: v += post_table[row % post_period][n]
  // Keypoint-branch tail + reference-point embedding (decoder helper chain; encoder_decoder.py:395-402, 363-371; N = 256, no LayerNorm):
  // with t this stage's output,  b_next[row] = sigmoid(inverse_sigmoid(kp_prev[row]) + t . kp_w[2, 256]^T + kp_b[2])  is stored to
  // kp_next [rows, 2], and with kp_dim_t the sine embedding of b_next ([32, 256]: 128 features of y, then 128 of x) goes to the LDS
  // buffer at s_off INSTEAD of t - the next stage is ref_point_head.
  const float* kp_w = nullptr; const float* kp_b = nullptr; const float* kp_prev = nullptr; float* kp_next = nullptr;
  const float* kp_dim_t = nullptr;
  float* out = nullptr; long ldo = 0;
  int s_off = -1;
  int keep = 0;
  int h1 = 0;                    // W is the single-pass fp16 packing (must agree with ChainP::h1)
};
struct ChainP {
  int rows = 0, n_stages = 0, lds_bytes = 0;
  // split = 2: TWO workgroups per slab.  Stages whose output feeds a later stage (s_off / keep / LayerNorm) are computed by both
  // (only part 0 stores them to global memory); every other stage's 256-column passes are dealt out alternately, so each
  // workgroup streams roughly half of those stages' weights.  Needs out != resid in the shared stages (the other part may still
  // be reading the residual): the callers ping-pong the token state between two buffers.
  int split = 1;
  // h1: single-pass fp16 arithmetic (the head's mixed precision, ec_gemm.hip GM_SPLIT1) in EVERY stage: the hi plane of the weights
  // (pack_chain_weights with h1) and of the LDS activations holds fp16, the lo planes are neither loaded nor written, one
  // v_mfma_f32_16x16x32_f16 per product.  Half the weight bytes - which is what bounds the kernel - and a third of the MFMAs.
  int h1 = 0;
  ChainStage st[CH_MAX_STAGES];
  // Row compaction (round 4): the chain is row-wise, and all MASKED keypoint tokens of a sample are identical rows in the reference
  // (head.py:187 multiplies the pooled support features by mask_s; adjacency rows / columns and key masks of masked tokens are
  // zeroed, skeleton.py:186-189), 62 % of the token rows on average.  With a plan (ec_ops.hip rowplan): slab row i of the launch
  // is token row rowmap[i], i < *n_active - every valid token plus ONE representative masked token per sample; workgroups past
  // *n_active leave at once.
  const int* rowmap = nullptr;
  const int* n_active = nullptr;
  // ... and every global output row of a representative is also written, inside the kernel, to the token rows of its sample's other
  // masked tokens (round 5; round 4: a bcast_rows launch behind every chain): active index i fans out to the rows fan_base[i] + k for
  // every set bit k of the two 64-bit words fan_bits[2 i], fan_bits[2 i + 1] (zero for valid tokens), so every token row of every
  // output is (re)written by every launch, as without compaction.
  const int* fan_base = nullptr;
  const unsigned long long* fan_bits = nullptr;
  int trace_stage = -1;        // TRACE: stage whose K loop / epilogue is stamped finely into trace[64..127]
  unsigned* trace = nullptr;   // EC_CHAIN_TRACE=1 (debug instantiation): s_memtime stamps of one mid-grid workgroup's wave 0
};

int chain_layout_bytes(int k);   // bytes of an operand buffer of k columns (32 rows)
int run_chain(const ChainP& p, hipStream_t st);
// host: W [N, K] fp32 -> fragment-major split packing, N*K*4 bytes (N % 16 == 0, K % 32 == 0)
void pack_chain_weights(const float* W, long N, long K, void* out, bool h1 = false);   // h1: hi plane = fp16(W), lo plane = 0

}  // namespace ec
