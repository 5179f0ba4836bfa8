// Launch wrappers for the non-GEMM kernels (ec_ops.hip). Internal.
#pragma once
#include "ec_common.h"

namespace ec {

// y = LayerNorm(x) * w + b over `cols`; one wave per row.  If drop_period > 0 the rows are grouped in
// periods of that length, the first row of each period (the cls token) is skipped and the output is
// packed (DINOv2 get_intermediate_layers: final norm, drop cls).  y may be fp32 or bf16.
// Optional second output y2 (fp32) = y + table[row % period2] (encoder: `src + pos` of the NEXT layer).
struct LnP {
  const float* x = nullptr; long ldx = 0;
  void* y = nullptr; long ldy = 0; int y_bf16 = 0;   // output format: 0 fp32, 1 bf16, 2 IEEE fp16, 3 bf16 split [hi | lo] (planes cols apart), 6 fp16x2 row
  const float* w = nullptr; const float* b = nullptr;
  int rows = 0, cols = 0; float eps = 1e-5f;
  int drop_period = 0;
  // optional fused residual add (bf16 backbone): x' = x + add (bf16 [rows, ldadd]); x' is written to xsum (may alias x,
  // same stride) and normalised.  Replaces the fp32 residual read-modify-write in the GEMM epilogue.
  const void* add = nullptr; long ldadd = 0;
  int add_fmt = 0;              // 16-bit format of add / add2 (1 bf16, 2 fp16); 0 = the output's format (bf16 for fp32 output)
  const void* add2 = nullptr;   // optional second branch (same stride): x' = (x + add) + add2
  float* xsum = nullptr;        // null with add set: x' is normalised but not written back
  void* y2 = nullptr; long ldy2 = 0;   // optional second output in IEEE fp16 (fp32 main output only): the 16-bit operand of a following GEMM
};
int layernorm(const LnP& p, hipStream_t st);

int add_table(float* x, long ldx, const float* table, long ldt, int period, int rows, int cols, hipStream_t st);
int copy2d(float* dst, long ldd, const float* src, long lds, int rows, int cols, hipStream_t st);
// dst[b][r][c] = src[b][r][c] for b < batch with batch strides
int copy3d(float* dst, long ldd, long sd, const float* src, long lds, long ss, int batch, int rows, int cols, hipStream_t st);
int mean_over(float* dst, const float* src, long stride, int n, long count, hipStream_t st);
// Indexed row transfer (support-side episode cache <-> per-call buffers): for r < n_rows and every outer slice o of every segment,
//   idx_is_dst = false (gather):  dst[o][r]      = src[o][idx[r]]
//   idx_is_dst = true (scatter):  dst[o][idx[r]] = src[o][r]
// rows of row_bytes bytes, outer strides in bytes.  The indices are host values and travel as kernel arguments.
constexpr int XFER_MAX_ROWS = 256;
struct XferSeg { void* dst = nullptr; const void* src = nullptr; long row_bytes = 0; int n_outer = 1; long dst_os = 0, src_os = 0; int align = 1; };
struct XferP {
  XferSeg seg[8]; int n_seg = 0; int idx_is_dst = 0;
  int idx[XFER_MAX_ROWS];
  void add(void* dst, const void* src, long row_bytes, int n_outer = 1, long dst_os = 0, long src_os = 0) {
    XferSeg& S = seg[n_seg++];
    S.dst = dst; S.src = src; S.row_bytes = row_bytes; S.n_outer = n_outer; S.dst_os = dst_os; S.src_os = src_os;
  }
};
int rows_xfer(XferP p, const int* idx_host, int n_rows, bool idx_is_dst, hipStream_t st);
int f32_to_bf16(const float* src, bf16_t* dst, long n, hipStream_t st, int f16 = 0);   // f16: IEEE fp16 instead of bf16
// x [rows, cols] fp32 (row stride ldx) -> fp16x2 rows [cols x fp16 | cols x e5m2 lo8 | cols x e5m2 hi8] (ec_common.h split4_x2), row stride
// ldy16 16-bit units (>= 2 cols); cols % 4 == 0.  The op-level test entry of the fp16x2 GEMM packs its A operand with it.
int pack_x2(const float* x, long ldx, void* y, long ldy16, int rows, int cols, hipStream_t st);
// src [B][L][E] fp32 -> dst [B][E][Lp] bf16 (columns >= L zeroed); test helper for the bf16 attention kernel
int transpose_pad_bf16(const float* src, bf16_t* dst, int B, int L, int E, int Lp, hipStream_t st);
int im2col14(const float* img, void* patches, int out_bf16, int n_img, int H, int W, int gh, int gw, int Kp, hipStream_t st);
int set_cls_rows(float* x, long ldx, const float* cls, const float* pos0, int n_img, int T, int C, hipStream_t st);
int nchw_to_tokens(const float* src, float* dst, int n, int C, int HW, hipStream_t st);
int tokens_to_nchw(const float* src, float* dst, int n, int C, int HW, hipStream_t st);

// on-device input pipeline (ec_preprocess_images / ec_msra_targets); batches of up to 16 images per launch
struct PreprocBatch {
  const unsigned char* src[16];   // RGB uint8 HWC source images (device)
  long pitch[16];                 // bytes per source row
  int hs[16], ws[16];
  float inv[16][6];               // dst -> src affine (row-major 2x3)
  float mean[3], stdv[3];
};
int preprocess_affine(const PreprocBatch& pb, int n, float* out, int H, hipStream_t st);
struct PreprocBatchCv2 {
  const unsigned char* src[16];
  long pitch[16];
  int hs[16], ws[16];
  double minv[16][6];             // dst -> src affine in float64, inverted on the host exactly as cv::warpAffine does
  float mean[3], stdv[3];
};
int preprocess_affine_cv2(const PreprocBatchCv2& pb, int n, float* out, int H, hipStream_t st);
struct MsraP {
  int hm, tmp;        // heatmap side, 3 * sigma
  double stride;      // image_size / heatmap_size (float64 in the reference)
  float g[49];        // (2*tmp+1)^2 gaussian, float32 values computed on the host exactly as the reference does
};
int msra_targets(const float* joints, const float* visible, float* target, float* weight, int n_kpts_total, const MsraP& mp,
                 hipStream_t st);

// head.py:175-184 — bilinear(g->hm) + normalised-heatmap pooling expressed as weights over the g*g cells
int pool_gather(const float* target, const float* mask_s, float inv_shots, const float* F, float* pooled, float beta, int bs, int K,
                int hm, int gh, int gw, int C, hipStream_t st);
int pool_taps(const float* target, const float* mask_s, float inv_shots, int* tap_n, int* tap_i, float* tap_w, int bs, int K, int hm, int gh,
              int gw, hipStream_t st);
int pool_apply(const int* tap_n, const int* tap_i, const float* tap_w, const float* F, float* pooled, float beta, int bs, int K, int hm, int gh,
               int gw, int C, hipStream_t st);
// skeleton.py:171-205 — edges -> binary adjacency, validity vectors, soft-normalised adjacency
int adj_build(const int32_t* edges, const int32_t* offsets, const float* mask_s, float* valid, uint8_t* kmask,
              uint8_t* kmask_fixed, float* binary, float* adj_r1, int bs, int K, hipStream_t st);
// Row compaction plan of the token-row chains (ec_chain.h ChainP::rowmap; round 4).  All masked keypoint tokens of a sample are
// identical rows throughout the head (head.py:187: pooled features * mask_s; skeleton.py:186-189 and encoder_decoder.py key masks: the
// adjacency rows / columns and attention keys of masked tokens are zeroed), so a row-wise kernel only has to compute the valid tokens
// and ONE masked token per sample.  For `ns` samples of K tokens, sample i using the mask row i % bs of mask_s [bs, K]:
//   plan[0] = n_active, plan[1] = n_copy, rowmap [ns*K]: token rows to compute (sample-major, valid tokens in order, then the
//   representative = the first masked token); fan_base [ns*K] / fan_bits [ns*K][2], indexed like rowmap: a representative's rows are
//   also those of its sample's other masked tokens, rows fan_base + k for every set bit k of the two 64-bit words (zero words for a
//   valid token) - the chain kernel writes them itself (ChainP::fan_*).
// A sample WITHOUT a valid token keeps token 0 as a row of its own: its key 0 is un-masked (encoder_decoder.py:359-360, skeleton.py:98-99)
// and sees a different attention bias than the other masked tokens do.
int rowplan(const float* mask_s, int bs, int ns, int K, int* plan, int* rowmap, int* fan_base, unsigned long long* fan_bits, hipStream_t st);
// skeleton.py:70-74 (learn_skeleton=False): adj_out [bs,2,K,K] = stack(diag(valid), adj_r1), adj1 = adj_r1
int adj_gt(const float* adj_r1, const float* valid, float* adj_out, float* adj1, int bs, int K, hipStream_t st);
int rownorm(const float* x, float* y, int rows, int cols, hipStream_t st);
// skeleton.py:134-161 — combine cosine similarity with the prior, soft-normalise, Markov matrix
int adj_combine(const float* P, const float* binary, const float* valid, const float* zc_w, const float* zc_b,
                float* adj_out, float* adj1, float* attn_adj, int bs, int K, hipStream_t st, int gram = 0);
int set_identity(float* dst, int bs, int K, hipStream_t st);
// bias_attn.py:188-191 — MLP(hops+1 -> hops+nhead -> nhead) over the Markov stack
int bias_mlp_layers(const float* attn_adj, const float* const* w1, const float* const* b1, const float* const* w2, const float* const* b2,
                    int n_layers, float* out, long out_stride, int hops1, int hidden, int nhead, int bs, int K, hipStream_t st);
int bias_mlp(const float* attn_adj, const float* w1, const float* b1, const float* w2, const float* b2, float* out,
             int hops1, int hidden, int nhead, int bs, int K, hipStream_t st);
// encoder_decoder.py:76-112 — softmax, soft-argmax, argmax 3x3 window local soft-argmax
int proposals(const float* sim, float* prop_loss, float* prop, int rows, int gh, int gw, hipStream_t st);
// positional_encoding.py:96-122
int sincos_coords(const float* coords, const float* inv_dim_t, float* out, long ldo, int rows, int num_feats, hipStream_t st);
// head.py:216-220 / encoder_decoder.py:395-431 — Linear(d->2) + sigmoid(inverse_sigmoid(prev) + delta)
int kpt_out(const float* h, long ldh, const float* W, const float* b, const float* prev, float* out, int rows, int d,
            hipStream_t st);

}  // namespace ec
