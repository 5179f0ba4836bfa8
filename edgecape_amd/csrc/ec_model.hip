// libedgecape_hip.so — model state, weight loading and the forward_test orchestration (C ABI in
// include/edgecape_hip.h).  The compute is exclusively the gfx950 kernels of ec_gemm / ec_attn / ec_ops;
// there is no CPU fallback: every entry point fails with EC_ERR_NODEVICE / EC_ERR_HIP if no GPU runs it.
//
// Data layout in HBM: all activations are token-major [batch, token, channel] fp32
// (bf16 for GEMM operands in bf16 mode), batch-first — unlike the reference's seq-first [L, bs, c] —
// so every Linear is one NT GEMM over M = batch*tokens rows and attention reads heads as column slices.
#include <math.h>
#include <cmath>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/edgecape_hip.h"
#include "ec_common.h"
#include "ec_chain.h"
#include "ec_ops.h"

namespace ec {

static thread_local std::string g_err;
void set_error(const std::string& s) { g_err = s; }
int hip_fail(hipError_t e, const char* what, const char* file, int line) {
  g_err = std::string("HIP error: ") + hipGetErrorString(e) + " in " + what + " (" + file + ":" + std::to_string(line) + ")";
  return EC_ERR_HIP;
}

struct Tensor {
  std::vector<int64_t> shape;
  std::vector<float> host;
  float* dev = nullptr;
  bf16_t* dev16 = nullptr;
  long numel() const { long n = 1; for (auto s : shape) n *= s; return n; }
};

struct Lin {  // a Linear layer view: W [N,K] (+ optional bf16 copy), bias [N]
  const float* w = nullptr;
  const bf16_t* w16 = nullptr;
  const float* ws = nullptr;   // bf16x3 split-packed copy (head_precision = EC_BF16X3), same byte size as w
  const void* wc = nullptr;    // ... and its fragment-major packing for the row-chain kernel (ec_chain.hip), head only
  const bf16_t* wf16 = nullptr;  // plain IEEE fp16 [N, K] copy (single-pass fp16 layers of the head's mixed precision): the 8-phase 16-bit
                               // GEMM runs the large image-row projections of the skeleton head on it when a 16-bit copy of A exists
  const bf16_t* w16x3 = nullptr; // bf16 [N, 2 K] = [W_hi | W_lo]: K-concatenated bf16x3 operand of the 8-phase GEMM (EC_BF16X3 backbone, GemmP::kwrap)
  const void* w_x2 = nullptr;    // fp16x2 rows [K x fp16 W_hi | K x e4m3 of W * 2^s1 | K x e4m3 of (W - W_hi) * 2^s2], 4 K bytes each (EC_F16X2 backbone)
  int x2_sa = 0;                 // ... E8M0 scale bytes of the two FP8 planes: (127 - s1) | (127 - s2) << 8 (GemmP::x2_sa)
  const float* b = nullptr;
  int N = 0, K = 0;
  bool w16_is_f16 = false;     // w16 holds IEEE fp16 (EC_F16 backbone) instead of bf16
  bool h1 = false;             // ws / wc are the single-pass fp16 packings (head mixed precision: skeleton head + decoder layers)
  const void* wsel(bool split) const { return split ? (const void*)ws : (const void*)w; }
};
struct Norm { const float* w = nullptr; const float* b = nullptr; };

struct BBlock { Norm n1, n2; Lin qkv, proj, fc1, fc2; const float* ls1 = nullptr; const float* ls2 = nullptr; };

struct DecLayer {
  std::vector<float> h_kv_w, h_kv_table;   // host copies of ca_kv weights / positional table (stacked across layers at finalize)
  Lin sa_in, sa_out;                 // self-attention in/out projections (fused [3d, d])
  const float *m_w1 = nullptr, *m_b1 = nullptr, *m_w2 = nullptr, *m_b2 = nullptr;  // markov_structural_mlp
  Lin ca_q, ca_kv, ca_fold;          // cross-attention: Q proj, fused K|V proj of image tokens, out_proj∘choker
  const float* ca_kv_table = nullptr;  // [HW, 2E] positional part of K (+ biases)
  Lin ffn1, ffn2;
  Norm n1, n2, n3, n4;
  Lin i2t_q, i2t_kv, i2t_fold;       // two-way (skeleton head only)
  const float* i2t_q_table = nullptr;  // [HW, E]
};
struct EncLayer { Lin in, out, l1, l2; Norm n1, n2; };
struct KptBranch { Lin l0, l2, l4; const float* w6 = nullptr; const float* b6 = nullptr; };

}  // namespace ec

using namespace ec;

struct ec_model {
  ec_config cfg;
  int gh = 0, gw = 0, HW = 0, T = 0, C = 0, K = 0, d = 0, L = 0, E = 0;   // token grid gh rows x gw columns (H / 14, W / 14: floor), HW = gh * gw
  int H = 0, W = 0;                                                       // input height / width (ec_config::image_size / image_width)
  int Kp = 640;  // padded im2col width (588 -> 640: multiple of 128 bytes for fp32 and bf16)
  bool finalized = false;
  bool gt_skel = false;      // SkeletonPredictor(learn_skeleton=False): the ground-truth adjacency, no skeleton layers, no Markov stack (skeleton.py:70-74)
  bool markov_bias = true;   // the decoder layers' self-attention adds the Markov-bias MLP (attn_bias=True AND a stack exists)
  bool bb16 = false;         // backbone GEMM operands / activations are 16-bit ...
  bool bbf16 = false;        // ... in IEEE fp16 (EC_F16) instead of bf16 (EC_BF16)
  bool bb_split = false;     // EC_BF16X3 backbone: fp32 activations, every MFMA operand split hi+lo bf16 (3 MFMAs per product)
  bool bb_x3 = false;        // ... its block GEMMs in the K-CONCATENATED form on the 8-phase 16-bit kernel (run_backbone); EC_BB_X3=0: A/B
  bool bb_x3_f16 = false;    // ... over IEEE fp16 planes (22 significand bits per operand) instead of bf16 planes (16): the fp16x2 mode's patch
                             // embedding (for the bf16x3 mode itself: measured at scale in round 6, not adopted)
  bool bb_x2 = false;        // EC_F16X2 backbone: the block GEMMs on fp16x2 operands (ec_common.h split4_x2; two MFMA units per product); the
                             // rest as the fp16-planes bf16x3 form (fp32 residual stream, split attention, fp16x3 patch embedding)
  bool head_split = false;   // head GEMMs in bf16x3 (ec_gemm.hip GM_SPLIT)
  bool head_mixed = false;   // ... except the Linear layers of the skeleton head and the decoder layers: single-pass fp16 (GM_SPLIT1)
  bool cur_h1 = false;       // build time: the Lin being made belongs to that set
  bool head_chain = false;   // ... and the row-wise stretches of every head layer as row-chain launches (ec_chain.hip); EC_CHAIN=0: off
  // Row compaction of the token-row chains (round 4; ec_ops.h rowplan): the chains compute the valid keypoint tokens and one
  // representative masked token per sample, whose output rows the kernel also writes to the sample's other masked rows.  EC_COMPACT=0: every row is computed.
  struct RowPlan { int* plan = nullptr; int* rowmap = nullptr; int* fan_base = nullptr; unsigned long long* fan_bits = nullptr; bool split = false; };
  RowPlan plan_dec, plan_skel;   // bs samples (decoder, keypoint branches) / S * bs samples (skeleton head)
  const RowPlan* skel_plan = nullptr;   // the plan the skeleton head of the call being enqueued reads (build_row_plans): plan_dec when one
                                        // plan over the same samples serves both (S == 1, both built from the same mask), else plan_skel
  // Measured (profiles/r04_compact_ab.txt, cfg2, interleaved): +1.5 % pairs/s through ec_forward_pipelined (5384 / 5385 / 5398 -> 5481 /
  // 5464 / 5466: a deferred head costs the backbone beside it CU time), -0.7 % through ec_forward with round 4's copy launch behind every
  // chain.  Round 5 (the fan-out inside the chain kernel, two workgroups per slab under a plan for plain calls; profiles/r05_compact_ab.txt,
  // interleaved): pipelined 5415 / 5413 with compaction in every call vs 5434 / (5192) pipelined-only - the same; ec_forward 4815 / 4858 vs
  // 4899 / 4890 - still -1 %: there the head's LATENCY counts, a chain workgroup takes as long as before (it is bound by its weight
  // stream) and the fan-out adds its tail to each of the 27 chains on the critical lanes.  So: pipelined calls only (compact_mode 1,
  // default); EC_COMPACT=2: every call, EC_COMPACT=0: never.  Results are bit-identical either way.
  // On ViT-S/14 @224, where the head is as long as the backbone, it is worth +4-5 % (11 170 -> 11 610-11 710 pairs/s).
  int compact_mode = 0;
  bool compact = false;          // ... for the call being enqueued
  bool episode_call = false;     // ... which is an episode-cache call: the support lane's samples (new episodes) are not the query lane's, so
                                 // the skeleton head builds a plan of its own (plan_skel) and the query lane builds the decoder's from the gathered masks
  // the support half of the head (pooling + SkeletonPredictor) has no query input: it runs on a side stream, concurrently
  // with input_proj / encoder / proposal generator on the caller's stream (both are small-grid, latency-bound kernels)
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_sk = nullptr, ev_join = nullptr;
  bool overlap = false;
  // the decoder's keypoint-branch / reference-point chains (encoder_decoder.py:371-402, head.py:216-220) hang off the token
  // state of each layer and only rejoin it at the next layer's cross-attention: they run on a second helper stream
  hipStream_t aux = nullptr, side2 = nullptr;   // side2: image lane of the skeleton head (run_head_support)
  hipEvent_t ev_aux[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_sup[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  bool overlap_dec = false;
  // ec_forward_pipelined (round 3): the decoder phase of a pipelined call runs on its own stream `dq` and is NOT joined at the end of
  // the call - it overlaps the next call's backbone; whatever touches the head workspace next (any entry point) first makes its
  // stream wait for ev_dq_done.  dq_active: the call being enqueued is a pipelined one; dq_pending: a decoder may still be running.
  hipStream_t dq = nullptr;
  hipEvent_t ev_dq_start = nullptr, ev_dq_done = nullptr;
  hipEvent_t ev_feat_read = nullptr;   // the support lane's last read of the backbone features (image_project): see run_head
  hipEvent_t ev_feat_read_p = nullptr, ev_feat_read_q = nullptr;   // ... the pooling's (support lane) and input_proj's (query lane)
  bool pipe_full = true;               // the WHOLE head of a pipelined call runs beside the next backbone (run_head); EC_PIPE_FULL=0: only its decoder phase
  bool dq_active = false, dq_pending = false;
  bool dq_recorded = false;            // ev_dq_done has been recorded at least once: ec_pipeline_flush may always wait for it
  bool feat_read_pending = false;      // FULL mode: the ev_feat_read* events have to be waited for before the feature buffer is rewritten
  // EC_TIMELINE=1: timed HIP events at the head's milestones on every stream, printed (us from the head's start) after a
  // device sync at the end of the call - the unprofiled picture of which lane is critical (rocprofv3 makes the head host-bound)
  bool timeline = false;
  bool timeline_defer = false;   // EC_TIMELINE=2: no device sync inside pipelined calls; the marks of all calls are printed by ec_pipeline_flush
  std::vector<std::pair<const char*, hipEvent_t>> tl;
  std::unordered_map<std::string, Tensor> tensors;
  std::vector<void*> owned;  // every hipMalloc'd pointer
  std::unordered_map<std::string, std::pair<const float*, long>> taps;
  bool prof_on = false;            // ec_profile: HIP events around the backbone QKV GEMM launches
  int prof_mode = 0;               // 1: every launch; 2: ONE launch per backbone pass, the block rotating with the pass (see ec_profile)
  size_t prof_pass = 0;            // backbone passes since the profile was armed
  std::vector<hipEvent_t> prof_ev;
  size_t prof_used = 0;

  // backbone
  Lin patch;
  const bf16_t* patch_w16x3 = nullptr;   // fp16 backbone: [C, 3 Kp] = [W_hi | W_hi | W_lo] of the patch embedding (split precision, see run_backbone)
  const float *cls = nullptr, *pos = nullptr;
  Norm bnorm;
  std::vector<BBlock> blocks;
  // head
  Lin input_proj, query_proj, image_project;
  const float *zc_w = nullptr, *zc_b = nullptr;
  const float *pos_img = nullptr, *pos_cat = nullptr, *dim_t = nullptr;
  std::vector<DecLayer> skel, dec;
  std::vector<EncLayer> enc;
  Norm dec_norm;
  Lin rp0, rp1, pg_support, pg_query, pg_dyn0, pg_dyn2;
  Lin dec_kv_all; const float* dec_kv_table = nullptr;   // decoder cross-attention K|V projections of all layers, stacked
  const float* dec_kv_table_L = nullptr;                 // ... the table padded to L rows (period L): the all-rows form of image_kv_all
  bf16_t* e_x16 = nullptr;                                // fp16 copy of the encoder output [bs*L, d] (operand of that form)
  std::vector<KptBranch> kpt;

  // workspace (device)
  int n_img_max = 0;
  float* bb_x = nullptr; void* bb_xn = nullptr; void* bb_qkv = nullptr; void* bb_att = nullptr; void* bb_h = nullptr; void* bb_y = nullptr; void* bb_y2 = nullptr;
  float* feat = nullptr;      // [n_img_max, HW, C] tokens; query first, then shot s at (1+s)*bs
  float* feat_nchw_tmp = nullptr;
  int32_t *d_edges = nullptr, *d_off = nullptr; int edges_cap = 0;
  int32_t *h_edges = nullptr, *h_off = nullptr;   // pinned staging of the skeleton edge lists
  hipEvent_t ev_edges = nullptr;
  float *pooled, *sk, *valid, *binary, *adj_r1, *adj1, *P, *kn, *kp_ref, *attn_adj;
  // dynamic tile schedule of the multi-round backbone GEMMs (QKV, fc1; GemmP::sched, ec_gemm8.hip): used by pipelined calls, whose
  // backbone shares the chip with the previous call's head (+0.75 % there; alone on the chip the static walk is 1.5-2 % faster per
  // launch).  EC_G8_DYN=0: never, =1: every call
  int* g8_sched = nullptr;
  int g8_dyn_mode = 2;
  int32_t *tap_n = nullptr, *tap_i = nullptr; float* tap_w = nullptr;   // compacted pooling taps per (shot, pair, keypoint): pool_taps / pool_apply
  hipEvent_t ev_inputs = nullptr, ev_call = nullptr;   // FULL mode: the caller's heatmaps / masks have been read; the call's inputs are ready
  uint8_t *kmask, *kmask_fixed;
  float *s_mem, *s_x, *s_tmp, *s_qkv, *s_att, *s_qc, *s_kv, *s_y, *s_z, *s_qimg, *s_kvk, *s_attimg, *s_tmpimg;
  bf16_t* s_mem16 = nullptr;   // fp16 copy of the skeleton head's image memory, written by norm4 (head mixed precision)
  float *e_x, *e_qkv, *e_att, *e_tmp, *e_h;
  float *p_fs, *p_fq, *p_g1, *p_fs2, *prop;
  float *d_qin, *d_sc, *d_rp, *d_bias, *d_bias_all, *d_qkv, *d_att, *d_tmp, *d_qc, *d_kv, *d_y, *d_z, *d_hs, *d_pts, *d_k1, *d_k2, *d_k3, *d_k4, *d_hn;
  float *o_sim, *o_adj, *o_init, *o_out;
};

struct ec_support;

namespace ec {

static int tl_mark(ec_model* m, const char* name, hipStream_t st) {
  if (!m->timeline) return 0;
  hipEvent_t e;
  EC_HIP(hipEventCreate(&e));
  EC_HIP(hipEventRecord(e, st));
  m->tl.push_back({name, e});
  return 0;
}
static int tl_dump(ec_model* m, bool at_flush = false) {
  if (!m->timeline || m->tl.empty()) return 0;
  if (m->timeline_defer && m->dq_active && !at_flush) return 0;   // EC_TIMELINE=2: pipelined calls accumulate, ec_pipeline_flush prints
  EC_HIP(hipDeviceSynchronize());
  fprintf(stderr, "[timeline]");
  for (size_t i = 0; i < m->tl.size(); ++i) {
    float ms = 0.f;
    EC_HIP(hipEventElapsedTime(&ms, m->tl[0].second, m->tl[i].second));
    fprintf(stderr, " %s=%.0f", m->tl[i].first, ms * 1e3f);
  }
  fprintf(stderr, "\n");
  for (auto& p : m->tl) (void)hipEventDestroy(p.second);
  m->tl.clear();
  return 0;
}

static int dmalloc(ec_model* m, void** p, size_t bytes) {
  if (bytes == 0) bytes = 16;
  EC_HIP(hipMalloc(p, bytes));
  m->owned.push_back(*p);
  return 0;
}
template <typename T> static int dalloc(ec_model* m, T** p, size_t n) { return dmalloc(m, (void**)p, n * sizeof(T)); }

static int upload(ec_model* m, const std::vector<float>& h, const float** out) {
  float* p = nullptr;
  int rc = dalloc(m, &p, h.size());
  if (rc) return rc;
  EC_HIP(hipMemcpy(p, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
  *out = p;
  return 0;
}
static int upload16(ec_model* m, const std::vector<float>& h, const bf16_t** out) {
  std::vector<bf16_t> t(h.size());
  if (m->bbf16) for (size_t i = 0; i < h.size(); ++i) t[i] = f2half_host(h[i]);
  else for (size_t i = 0; i < h.size(); ++i) t[i] = f2bf(h[i]);
  bf16_t* p = nullptr;
  int rc = dalloc(m, &p, t.size());
  if (rc) return rc;
  EC_HIP(hipMemcpy(p, t.data(), t.size() * sizeof(bf16_t), hipMemcpyHostToDevice));
  *out = p;
  return 0;
}

static int upload_split(ec_model* m, const float* W, long rows, long K, const float** out) {
  EC_REQUIRE(K % 32 == 0, EC_ERR_ARG, "bf16x3 packing needs K % 32 == 0");
  std::vector<float> packed((size_t)rows * K);
  if (m->cur_h1) split_pack_weights_h1(W, rows, K, packed.data());
  else split_pack_weights(W, rows, K, packed.data());
  return upload(m, packed, out);
}
// K-concatenated bf16x3 operand: row n = [W_hi | W_lo] (bf16, 2 K long), walked by the GEMM's load stream as [W_hi | W_hi | W_lo]
// (GemmP::kwrap).  Against activation rows [a_hi | a_lo], walked as [a_hi | a_lo | a_hi], ONE 16-bit GEMM of depth 3 K adds
// a_hi W_hi + a_lo W_hi + a_hi W_lo in its fp32 accumulators - the three products of the bf16x3 mode (the lo x lo term, ~2^-16 of the
// product, is dropped there as well).
static int upload_x3(ec_model* m, const float* W, long rows, long K, const bf16_t** out) {
  std::vector<bf16_t> w3((size_t)rows * 2 * K);
  for (long n = 0; n < rows; ++n) {
    bf16_t* row = &w3[(size_t)n * 2 * K];
    for (long k = 0; k < K; ++k) {
      const float w = W[n * K + k];
      if (m->bb_x3_f16) {   // fp16 planes: the lo part of a small weight is an fp16 subnormal (kept by the matrix pipe)
        const bf16_t h = f2half_host(w);
        row[k] = h; row[K + k] = f2half_host(w - half2f_host(h));
      } else {
        const bf16_t h = f2bf(w);
        row[k] = h; row[K + k] = f2bf(w - bf2f(h));
      }
    }
  }
  bf16_t* p3 = nullptr;
  int rc = dalloc(m, &p3, w3.size());
  if (rc) return rc;
  EC_HIP(hipMemcpy(p3, w3.data(), w3.size() * sizeof(bf16_t), hipMemcpyHostToDevice));
  *out = p3;
  return 0;
}
// host: float -> OCP e4m3fn bit pattern, round to nearest even, saturating at +-448 (the weights' FP8 planes are packed once at ec_finalize)
static uint8_t f2e4m3_host(float f) {
  const uint8_t sgn = std::signbit(f) ? 0x80 : 0;
  float a = std::fabs(f);
  if (std::isnan(a)) return sgn | 0x7f;
  if (a > 448.f) a = 448.f;
  if (a < 0.0009765625f) return sgn;                                  // below 2^-10: half the smallest subnormal (a tie goes to the even 0)
  int e;
  (void)std::frexp(a, &e);
  const int E = e - 1;                                                // a = 1.xxx * 2^E
  if (E < -6) return sgn | (uint8_t)std::nearbyint(std::ldexp(a, 9));   // subnormal: multiples of 2^-9 (8 = the smallest normal's pattern)
  int q = (int)std::nearbyint(std::ldexp(a, 3 - E)), Eb = E + 7;     // 8 .. 16
  if (q == 16) { q = 8; ++Eb; }
  const int code = (Eb << 3) | (q - 8);
  return sgn | (uint8_t)(code > 0x7e ? 0x7e : code);
}
// fp16x2 weight rows (ec_common.h, split4_x2): [W_hi fp16 | e4m3(W * 2^s1) | e4m3((W - W_hi) * 2^s2)], s = the power of two that puts the
// plane's largest magnitude into e4m3's top binade (256 .. 448]
static int pack_x2_weights(const float* W, long rows, long K, std::vector<uint8_t>& w, int* sa) {
  EC_REQUIRE(K % 128 == 0, EC_ERR_ARG, "fp16x2 packing needs K % 128 == 0");
  float amax_w = 0.f, amax_l = 0.f;
  for (long i = 0; i < rows * K; ++i) {
    amax_w = std::max(amax_w, std::fabs(W[i]));
    amax_l = std::max(amax_l, std::fabs(W[i] - half2f_host(f2half_host(W[i]))));
  }
  auto pow2 = [](float amax) { return amax > 0.f && std::isfinite(amax) ? std::min(100, std::max(-100, (int)std::floor(std::log2(448.0 / amax)))) : 0; };
  const int s1 = pow2(amax_w), s2 = pow2(amax_l);
  w.resize((size_t)rows * 4 * K);
  for (long n = 0; n < rows; ++n) {
    uint8_t* row = &w[(size_t)n * 4 * K];
    for (long k = 0; k < K; ++k) {
      const float v = W[n * K + k];
      const bf16_t h = f2half_host(v);
      ((bf16_t*)row)[k] = h;
      row[2 * K + k] = f2e4m3_host(std::ldexp(v, s1));
      row[3 * K + k] = f2e4m3_host(std::ldexp(v - half2f_host(h), s2));
    }
  }
  *sa = (127 - s1) | ((127 - s2) << 8);
  return 0;
}
static int upload_x2(ec_model* m, const float* W, long rows, long K, const void** out, int* sa) {
  std::vector<uint8_t> w;
  int rc = pack_x2_weights(W, rows, K, w, sa);
  if (rc) return rc;
  void* p = nullptr;
  if ((rc = dmalloc(m, &p, w.size()))) return rc;
  EC_HIP(hipMemcpy(p, w.data(), w.size(), hipMemcpyHostToDevice));
  *out = p;
  return 0;
}
// fragment-major split packing for the row-chain kernel; shapes the kernel cannot take simply get no such copy
static int upload_chain(ec_model* m, const float* W, long rows, long K, const void** out) {
  if (rows % 32 != 0 || K % 128 != 0) return 0;
  std::vector<float> packed((size_t)rows * K);
  pack_chain_weights(W, rows, K, packed.data(), m->cur_h1);
  const float* dev = nullptr;
  int rc = upload(m, packed, &dev);
  *out = dev;
  return rc;
}

static const Tensor* find(ec_model* m, const std::string& name) {
  auto it = m->tensors.find(name);
  return it == m->tensors.end() ? nullptr : &it->second;
}

#define GET(var, name)                                                           \
  const Tensor* var = find(m, name);                                             \
  if (!var) { set_error(std::string("missing tensor: ") + (name)); return EC_ERR_STATE; }

static bool name_is_head(const std::string& n) { return n.compare(0, 21, "keypoint_head_module.") == 0; }
static int make_lin(ec_model* m, const std::string& wname, const std::string& bname, Lin* out, bool want16) {
  GET(w, wname);
  out->w = w->dev;
  out->N = (int)w->shape[0];
  out->K = (int)(w->numel() / w->shape[0]);
  if (!bname.empty()) {
    GET(b, bname);
    out->b = b->dev;
  }
  if (want16) {
    int rc = upload16(m, w->host, &out->w16);
    if (rc) return rc;
    out->w16_is_f16 = m->bbf16;
  } else if (m->bb_x2 && !name_is_head(wname)) {   // fp16x2 backbone: the only copy its block GEMMs read
    int rc = upload_x2(m, w->host.data(), out->N, out->K, &out->w_x2, &out->x2_sa);
    if (rc) return rc;
  } else if (m->bb_x3 && !name_is_head(wname)) {   // bf16x3 backbone, K-concatenated form: the only copy its GEMMs read
    int rc = upload_x3(m, w->host.data(), out->N, out->K, &out->w16x3);
    if (rc) return rc;
  } else if ((m->head_split && name_is_head(wname)) || (m->bb_split && !name_is_head(wname))) {
    int rc = upload_split(m, w->host.data(), out->N, out->K, &out->ws);
    if (rc) return rc;
    if (m->head_chain && name_is_head(wname) && (rc = upload_chain(m, w->host.data(), out->N, out->K, &out->wc))) return rc;
    out->h1 = m->cur_h1 && name_is_head(wname);
  }
  return 0;
}
static int make_lin_host(ec_model* m, const std::vector<float>& W, const std::vector<float>& b, int N, int K, Lin* out) {
  out->N = N; out->K = K;
  int rc = upload(m, W, &out->w);
  if (rc) return rc;
  if (m->head_split && (rc = upload_split(m, W.data(), N, K, &out->ws))) return rc;   // make_lin_host is only used by the head
  if (m->head_chain && (rc = upload_chain(m, W.data(), N, K, &out->wc))) return rc;
  out->h1 = m->head_split && m->cur_h1;
  if (out->h1 && N >= 256 && K % 128 == 0) {   // (shapes the 8-phase kernel takes)
    std::vector<bf16_t> t(W.size());
    for (size_t i = 0; i < W.size(); ++i) t[i] = f2half_host(W[i]);
    bf16_t* p16 = nullptr;
    if ((rc = dalloc(m, &p16, t.size()))) return rc;
    EC_HIP(hipMemcpy(p16, t.data(), t.size() * sizeof(bf16_t), hipMemcpyHostToDevice));
    out->wf16 = p16;
  }
  if (!b.empty()) rc = upload(m, b, &out->b);
  return rc;
}
static int make_norm(ec_model* m, const std::string& p, Norm* out) {
  GET(w, p + ".weight");
  GET(b, p + ".bias");
  out->w = w->dev; out->b = b->dev;
  return 0;
}

// out[M,N] = A[M,K] @ B[N,K]^T in double (host, finalize-time folding of constant operands)
static std::vector<float> host_nt(const float* A, long lda, const float* B, long ldb, int M, int N, int K) {
  std::vector<float> o((size_t)M * N);
  for (int i = 0; i < M; ++i)
    for (int j = 0; j < N; ++j) {
      double s = 0;
      for (int k = 0; k < K; ++k) s += (double)A[i * lda + k] * (double)B[j * ldb + k];
      o[(size_t)i * N + j] = (float)s;
    }
  return o;
}

// SinePositionalEncoding.forward on an all-False mask (positional_encoding.py:57-94) -> [HW, 2*nf] token-major
static std::vector<float> sine_table(int gh, int gw, int nf, std::vector<float>* dim_t_out) {
  std::vector<float> dim_t(nf);
  for (int i = 0; i < nf; ++i) dim_t[i] = powf(10000.f, (float)(2 * (i / 2)) / (float)nf);
  const float scale = 2.f * (float)M_PI;
  std::vector<float> t((size_t)gh * gw * 2 * nf);
  for (int y = 0; y < gh; ++y)
    for (int x = 0; x < gw; ++x) {
      const float ye = (float)(y + 1) / ((float)gh + 1e-6f) * scale;   // cumsum over rows / (last row + eps)
      const float xe = (float)(x + 1) / ((float)gw + 1e-6f) * scale;
      float* o = &t[((size_t)y * gw + x) * 2 * nf];
      for (int i = 0; i < nf; ++i) {
        const float ay = ye / dim_t[i], ax = xe / dim_t[i];
        o[i] = (i & 1) ? cosf(ay) : sinf(ay);
        o[nf + i] = (i & 1) ? cosf(ax) : sinf(ax);
      }
    }
  if (dim_t_out) *dim_t_out = dim_t;
  return t;
}

// Build the derived weights of one TransformerDecoderLayer (encoder_decoder.py:527-651).
// biased: a main-decoder layer (q / k / v_proj keys - engine.normalize_state_dict splits a fused in_proj into them -, query = [x | qpe]);
// markov: ... whose self-attention adds the Markov-bias MLP of the hop stack (bias_attn.py:188-191)
static int build_dec_layer(ec_model* m, const std::string& P, bool biased, bool two_way, const std::vector<float>& pos_img,
                           DecLayer* L, bool markov = true) {
  const int d = m->d, E = 2 * m->d, HW = m->HW;
  int rc;
  if (biased) {
    GET(q, P + "self_attn.q_proj.weight"); GET(k, P + "self_attn.k_proj.weight"); GET(v, P + "self_attn.v_proj.weight");
    GET(qb, P + "self_attn.q_proj.bias"); GET(kb, P + "self_attn.k_proj.bias"); GET(vb, P + "self_attn.v_proj.bias");
    std::vector<float> W, b;
    for (const Tensor* t : {q, k, v}) W.insert(W.end(), t->host.begin(), t->host.end());
    for (const Tensor* t : {qb, kb, vb}) b.insert(b.end(), t->host.begin(), t->host.end());
    if ((rc = make_lin_host(m, W, b, 3 * d, d, &L->sa_in))) return rc;
    if (markov) {
      GET(w1, P + "self_attn.markov_structural_mlp.0.weight"); GET(b1, P + "self_attn.markov_structural_mlp.0.bias");
      GET(w2, P + "self_attn.markov_structural_mlp.3.weight"); GET(b2, P + "self_attn.markov_structural_mlp.3.bias");
      L->m_w1 = w1->dev; L->m_b1 = b1->dev; L->m_w2 = w2->dev; L->m_b2 = b2->dev;
    }
  } else {
    if ((rc = make_lin(m, P + "self_attn.in_proj_weight", P + "self_attn.in_proj_bias", &L->sa_in, false))) return rc;
  }
  if ((rc = make_lin(m, P + "self_attn.out_proj.weight", P + "self_attn.out_proj.bias", &L->sa_out, false))) return rc;

  auto cross = [&](const std::string& A, const std::string& choker, bool q_full, Lin* lq, const float** q_table, Lin* lkv,
                   const float** kv_table, bool kv_from_tokens, Lin* fold) -> int {
    GET(wq, A + "q_proj_weight"); GET(wk, A + "k_proj_weight"); GET(wv, A + "v_proj_weight");
    GET(bi, A + "in_proj_bias"); GET(wo, A + "out_proj.weight"); GET(bo, A + "out_proj.bias");
    GET(cw, choker + ".weight"); GET(cb, choker + ".bias");
    EC_REQUIRE(wq->shape[0] == E && wq->shape[1] == E && wv->shape[1] == d, EC_ERR_ARG, "cross-attention weight shapes");
    const float* bq = bi->host.data(); const float* bk = bq + E; const float* bv = bk + E;
    int r;
    // ---- Q side
    if (q_full) {           // main decoder: Q = [x | qpe] @ Wq^T + bq (K = 2d)
      if ((r = make_lin_host(m, wq->host, std::vector<float>(bq, bq + E), E, E, lq))) return r;
    } else if (!q_table) {  // skeleton token->image: positional half of the query is zero
      std::vector<float> Wa((size_t)E * d);
      for (int n = 0; n < E; ++n) memcpy(&Wa[(size_t)n * d], &wq->host[(size_t)n * E], d * sizeof(float));
      if ((r = make_lin_host(m, Wa, std::vector<float>(bq, bq + E), E, d, lq))) return r;
    } else {                // skeleton image->token: query = [mem | pos_img]; fold the constant positional half
      std::vector<float> Wa((size_t)E * d);
      for (int n = 0; n < E; ++n) memcpy(&Wa[(size_t)n * d], &wq->host[(size_t)n * E], d * sizeof(float));
      if ((r = make_lin_host(m, Wa, {}, E, d, lq))) return r;
      std::vector<float> tb = host_nt(pos_img.data(), d, wq->host.data() + d, E, HW, E, d);
      for (int t = 0; t < HW; ++t) for (int n = 0; n < E; ++n) tb[(size_t)t * E + n] += bq[n];
      if ((r = upload(m, tb, q_table))) return r;
    }
    // ---- K|V side: one GEMM with stacked weights [2E, d]
    std::vector<float> Wkv((size_t)2 * E * d);
    for (int n = 0; n < E; ++n) memcpy(&Wkv[(size_t)n * d], &wk->host[(size_t)n * E], d * sizeof(float));
    memcpy(&Wkv[(size_t)E * d], wv->host.data(), (size_t)E * d * sizeof(float));
    if (kv_from_tokens) {   // keys = [x | 0]: plain biases
      std::vector<float> b(bk, bk + 2 * E);
      if ((r = make_lin_host(m, Wkv, b, 2 * E, d, lkv))) return r;
    } else {                // keys = [mem | pos_img]: positional half is a constant [HW, E] table
      if ((r = make_lin_host(m, Wkv, {}, 2 * E, d, lkv))) return r;
      std::vector<float> tk = host_nt(pos_img.data(), d, wk->host.data() + d, E, HW, E, d);
      std::vector<float> tb((size_t)HW * 2 * E);
      for (int t = 0; t < HW; ++t) {
        for (int n = 0; n < E; ++n) tb[(size_t)t * 2 * E + n] = tk[(size_t)t * E + n] + bk[n];
        for (int n = 0; n < E; ++n) tb[(size_t)t * 2 * E + E + n] = bv[n];
      }
      if ((r = upload(m, tb, kv_table))) return r;
      L->h_kv_w = Wkv; L->h_kv_table = tb;
    }
    // ---- out_proj followed by choker, no non-linearity in between (encoder_decoder.py:624-631): fold
    std::vector<float> Wf((size_t)d * E), bf(d);
    for (int i = 0; i < d; ++i) {
      for (int j = 0; j < E; ++j) {
        double s = 0;
        for (int k = 0; k < E; ++k) s += (double)cw->host[(size_t)i * E + k] * (double)wo->host[(size_t)k * E + j];
        Wf[(size_t)i * E + j] = (float)s;
      }
      double s = cb->host[i];
      for (int k = 0; k < E; ++k) s += (double)cw->host[(size_t)i * E + k] * (double)bo->host[k];
      bf[i] = (float)s;
    }
    return make_lin_host(m, Wf, bf, d, E, fold);
  };
  if ((rc = cross(P + "multihead_attn.", P + "choker", biased, &L->ca_q, nullptr, &L->ca_kv, &L->ca_kv_table, false, &L->ca_fold)))
    return rc;
  if ((rc = make_lin(m, P + "ffn1.conv.weight", P + "ffn1.conv.bias", &L->ffn1, false))) return rc;
  if ((rc = make_lin(m, P + "ffn2.weight", P + "ffn2.bias", &L->ffn2, false))) return rc;
  if ((rc = make_norm(m, P + "norm1", &L->n1)) || (rc = make_norm(m, P + "norm2", &L->n2)) || (rc = make_norm(m, P + "norm3", &L->n3)))
    return rc;
  if (two_way) {
    if ((rc = cross(P + "cross_attn_image_to_token.", P + "cross_attn_image_to_token_choker", false, &L->i2t_q, &L->i2t_q_table,
                    &L->i2t_kv, nullptr, true, &L->i2t_fold)))
      return rc;
    if ((rc = make_norm(m, P + "norm4", &L->n4))) return rc;
  }
  return 0;
}

// ---- thin launch helpers -------------------------------------------------------------------------
static int linear(const void* A, long lda, bool a16, const Lin& W, void* C, long ldc, bool c16, int M, int act, hipStream_t st,
                  const float* gamma = nullptr, const float* resid = nullptr, long ldr = 0, const float* table = nullptr,
                  long ldt = 0, int period = 1, const float* aux = nullptr, long ldaux = 0, int tag = 0, int* sched = nullptr) {
  GemmP p;
  p.tag = tag; p.sched = sched;
  p.A = A; p.lda = lda; p.ab_bf16 = a16 ? 1 : 0; p.h_f16 = (a16 && W.w16_is_f16) ? 1 : 0;
  p.split = (!a16 && W.ws) ? (W.h1 ? 2 : 1) : 0;   // head in bf16x3 mode: every head Lin carries a split-packed copy (h1: fp16x1)
  p.B = a16 ? (const void*)W.w16 : W.wsel(p.split); p.ldb = W.K;
  EC_REQUIRE(p.B != nullptr, EC_ERR_STATE, "linear: weight copy for this precision was not built");
  p.C = C; p.ldc = ldc; p.c_bf16 = c16 ? 1 : 0;
  p.bias = W.b; p.gamma = gamma; p.resid = resid; p.ldr = ldr; p.table = table; p.ldt = ldt; p.period = period;
  p.aux = aux; p.ldaux = ldaux;
  p.M = M; p.N = W.N; p.K = W.K; p.act = act;
  return gemm_nt(p, st);
}

// K-concatenated bf16x3 Linear: A = bf16 [M, 2 K] planes [hi | lo] (row stride 2 K), W.w16x3 = [W_hi | W_lo]; C fp32, or (c_x3) the
// split planes of the result, row stride ldc
static int linear_x3(const void* A, const Lin& W, void* C, long ldc, bool c_x3, int M, int act, hipStream_t st, const float* gamma,
                     const float* resid, long ldr, int tag, bool f16) {
  EC_REQUIRE(W.w16x3 != nullptr, EC_ERR_STATE, "linear_x3: K-concatenated weight copy was not built");
  GemmP p;
  p.tag = tag;
  p.A = A; p.lda = 2l * W.K; p.ab_bf16 = 1; p.h_f16 = f16 ? 1 : 0;
  p.B = W.w16x3; p.ldb = 2l * W.K; p.kwrap = W.K / 64;
  p.C = C; p.ldc = ldc; p.c_x3 = c_x3 ? 1 : 0;
  p.bias = W.b; p.gamma = gamma; p.resid = resid; p.ldr = ldr;
  p.M = M; p.N = W.N; p.K = 3 * W.K; p.act = act;
  return gemm_nt(p, st);
}

// fp16x2 Linear (EC_F16X2 backbone): A = fp16x2 rows [hi | lo8 | hi8] of W.K values (row stride 2 K 16-bit units), W.w_x2 likewise; C fp32
// (optionally LayerScale + the in-place residual), or (c_x2) the fp16x2 rows of gelu(.), row stride ldc 16-bit units
static int linear_x2(const void* A, const Lin& W, void* C, long ldc, bool c_x2, int M, int act, hipStream_t st, const float* gamma,
                     const float* resid, long ldr, int tag, int* sched = nullptr) {
  EC_REQUIRE(W.w_x2 != nullptr, EC_ERR_STATE, "linear_x2: fp16x2 weight copy was not built");
  GemmP p;
  p.tag = tag; p.sched = sched;
  p.A = A; p.lda = 2l * W.K; p.ab_bf16 = 1; p.h_f16 = 1;
  p.B = W.w_x2; p.ldb = 2l * W.K; p.x2 = W.K / 64; p.x2_sa = W.x2_sa;
  p.C = C; p.ldc = ldc; p.c_x2 = c_x2 ? 1 : 0;
  p.bias = W.b; p.gamma = gamma; p.resid = resid; p.ldr = ldr;
  p.M = M; p.N = W.N; p.K = 2 * W.K; p.act = act;
  return gemm_nt(p, st);
}

static int ln(const float* x, long ldx, void* y, long ldy, int y16, const Norm& n, int rows, int cols, float eps, hipStream_t st,
              int drop_period = 0, const void* add = nullptr, long ldadd = 0, const void* add2 = nullptr, bool write_x = true,
              int add_fmt = 0) {
  LnP p;
  p.x = x; p.ldx = ldx; p.y = y; p.ldy = ldy; p.y_bf16 = y16; p.add_fmt = add_fmt; p.w = n.w; p.b = n.b; p.rows = rows; p.cols = cols; p.eps = eps;
  p.drop_period = drop_period;
  p.add = add; p.ldadd = ldadd; p.add2 = add2; p.xsum = (add && write_x) ? const_cast<float*>(x) : nullptr;
  return layernorm(p, st);
}

#define RUN(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

// ---------------------------------------------------------------------------------------------
// Backbone: DINOv2 ViT.  n images -> m->feat [n, HW, C] fp32 token-major.
// ---------------------------------------------------------------------------------------------
// The images may come from several tensors of n_each images (query batch + S support batches): they are
// gathered by the im2col step so that the whole ViT runs ONCE over n = n_src * n_each images (one large-M GEMM
// per layer instead of 1+S smaller ones).
// The sources may hold different numbers of images (counts[s]; the streaming episode path: a batch of queries + the support images of
// the episodes that start in this call); a source with no image is skipped.
static int run_backbone(ec_model* m, const float* const* imgs, const int* counts, int n_src, float* feat_out, hipStream_t st) {
  const int C = m->C, T = m->T, HW = m->HW;
  const bool h16 = m->bb16;
  const int hfmt = h16 ? (m->bbf16 ? 2 : 1) : 0;   // 16-bit storage format: 0 fp32, 1 bf16, 2 fp16
  const int nh = m->cfg.num_heads;
  int n = 0;
  for (int s = 0; s < n_src; ++s) n += counts[s];
  EC_REQUIRE(n > 0 && n <= m->n_img_max, EC_ERR_ARG, "backbone: image count outside 1..(1+max_shots)*max_batch");
  const long M = (long)n * T;
  // fp16 backbone: the patch embedding in SPLIT precision (round 3).  The image patches and the patch weights rounded to fp16 carry 38 %
  // of the error variance of the backbone's features (oracle/precision_sites.py) - every later block inherits it through the residual
  // stream - for 0.5 % of the FLOPs: with [hi | lo | hi] patches against [W_hi | W_hi | W_lo] weights the same 8-phase GEMM, K = 3 Kp,
  // computes the products to ~2^-22.  (The bf16 backbone keeps single operands; the fp16 A/B switch EC_PATCH_X3 went in round 5.)
  const bool px3 = m->patch_w16x3 != nullptr;       // fp16 backbone: three fp16 planes; bf16x3 backbone (K-concatenated form): two bf16 planes
  const bool pb3 = px3 && m->bb_x3;
  const int Kpe = pb3 ? 2 * m->Kp : px3 ? 3 * m->Kp : m->Kp;   // row length of the patch rows / weight rows in memory
  for (int s = 0, at = 0; s < n_src; at += counts[s], ++s)
    if (counts[s] > 0)
      RUN(im2col14(imgs[s], (char*)m->bb_h + (size_t)at * T * Kpe * (h16 || pb3 ? 2 : 4), pb3 ? (m->bb_x3_f16 ? 5 : 4) : px3 ? 3 : hfmt, counts[s], m->H, m->W, m->gh, m->gw, m->Kp, st));
  {  // patch embedding: ONE GEMM over all n*T token rows (the zero cls rows produce bias + pos[0], overwritten below);
     // epilogue adds the conv bias and the positional table row m % T
    GemmP p;
    p.A = m->bb_h; p.lda = Kpe; p.ab_bf16 = h16 || pb3; p.h_f16 = m->bbf16 || (pb3 && m->bb_x3_f16);
    p.split = (m->bb_split && !pb3) ? 1 : 0;
    p.B = px3 ? (const void*)m->patch_w16x3 : h16 ? (const void*)m->patch.w16 : m->patch.wsel(m->bb_split); p.ldb = Kpe;
    p.C = m->bb_x; p.ldc = C;
    p.bias = m->patch.b; p.table = m->pos; p.ldt = C; p.period = T;
    p.M = (int)M; p.N = C; p.K = pb3 ? 3 * m->Kp : Kpe;
    if (pb3) p.kwrap = m->Kp / 64;
    RUN(gemm_nt(p, st));
  }
  RUN(set_cls_rows(m->bb_x, C, m->cls, m->pos, n, T, C, st));
  if (!feat_out) m->taps["tokens0"] = {m->bb_x, M * C};
  // bf16 mode: the branch GEMMs (proj, fc2) write y = gamma * (acc + bias) as bf16 and the residual add x += y is fused
  // into the FOLLOWING LayerNorm (same HBM bytes, but the fp32 read-modify-write leaves the GEMM epilogue).
  // fp32 mode: the residual is added in the GEMM epilogue (exact fp32 stream).
  // The attention branch y1 is NOT written into x by norm2 (it only normalises x + y1): the next norm1 adds both pending
  // branches, x <- (x + y1) + y2, the same two fp32 additions in the same order.  22 instead of 24 bytes per element and block.
  const void *pend = nullptr, *pend2 = nullptr;   // branch outputs not yet added to x (attention branch, MLP branch)
  int* const sched = (m->g8_dyn_mode == 1 || (m->g8_dyn_mode == 2 && m->dq_active)) ? m->g8_sched : nullptr;
  for (size_t i = 0; i < m->blocks.size(); ++i) {
    const BBlock& b = m->blocks[i];
    const bool x3 = m->bb_x3, x2 = m->bb_x2;
    RUN(ln(m->bb_x, C, m->bb_xn, x3 ? 2 * C : C, x2 ? 6 : x3 ? 3 : hfmt, b.n1, (int)M, C, 1e-6f, st, 0, pend, C, pend2));
    pend = pend2 = nullptr;
    const bool prof = m->prof_on && (m->prof_mode != 2 || i == m->prof_pass % m->blocks.size()) && m->prof_used + 2 <= m->prof_ev.size();
    if (prof) EC_HIP(hipEventRecord(m->prof_ev[m->prof_used++], st));
    if (x2) {
      RUN(linear_x2(m->bb_xn, b.qkv, m->bb_qkv, 3 * C, false, (int)M, ACT_NONE, st, nullptr, nullptr, 0, 1, sched));
    } else if (x3) {
      RUN(linear_x3(m->bb_xn, b.qkv, m->bb_qkv, 3 * C, false, (int)M, ACT_NONE, st, nullptr, nullptr, 0, 1, false));
    } else {
      GemmP p;
      p.tag = 1;
      p.A = m->bb_xn; p.lda = C; p.ab_bf16 = h16; p.h_f16 = m->bbf16;
      p.split = m->bb_split ? 1 : 0;
      p.B = h16 ? (const void*)b.qkv.w16 : b.qkv.wsel(m->bb_split); p.ldb = C;
      p.C = m->bb_qkv; p.ldc = 3 * C; p.c_bf16 = h16; p.bias = b.qkv.b;
      p.M = (int)M; p.N = 3 * C; p.K = C;
      p.sched = sched;
      RUN(gemm_nt(p, st));
    }
    if (prof) EC_HIP(hipEventRecord(m->prof_ev[m->prof_used++], st));
    AttnP a;
    const size_t es = h16 ? 2 : 4;
    a.Q = m->bb_qkv; a.K = (const char*)m->bb_qkv + (size_t)C * es; a.V = (const char*)m->bb_qkv + (size_t)2 * C * es;
    a.O = m->bb_att;
    a.ldq = a.ldk = a.ldv = 3 * C; a.ldo = C;
    a.sQ = a.sK = a.sV = (long)T * 3 * C; a.sO = (long)T * C;
    a.B = n; a.H = nh; a.Lq = T; a.Lk = T; a.hd = C / nh; a.bf16 = h16; a.f16 = m->bbf16; a.split = m->bb_split ? 1 : 0;
    if (x3) { a.o_x3 = x2 ? 3 : 1; a.ldo = 2 * C; a.sO = (long)T * 2 * C; }
    RUN(attention(a, st));
    if (x2) {
      // fp16x2 (round 6): the same four launches on fp16x2 operands - a_hi W_hi in fp16 MFMAs, both correction terms in one FP8 pass
      RUN(linear_x2(m->bb_att, b.proj, m->bb_x, C, false, (int)M, ACT_NONE, st, b.ls1, m->bb_x, C, 2));
      RUN(ln(m->bb_x, C, m->bb_xn, 2 * C, 6, b.n2, (int)M, C, 1e-6f, st));
      RUN(linear_x2(m->bb_xn, b.fc1, m->bb_h, 8 * C, true, (int)M, ACT_GELU, st, nullptr, nullptr, 0, 3, sched));
      RUN(linear_x2(m->bb_h, b.fc2, m->bb_x, C, false, (int)M, ACT_NONE, st, b.ls2, m->bb_x, C, 4));
    } else if (x3) {
      // K-concatenated bf16x3 (round 5): every block GEMM is ONE 16-bit GEMM of depth 3 K on the 8-phase kernel - activations as bf16
      // [hi | lo] planes written by their producers (LayerNorm, attention, the fc1 epilogue) and walked hi | lo | hi by the load stream,
      // weights [W_hi | W_lo] walked hi | hi | lo (GemmP::kwrap) - with the fp32 epilogue of the exact mode (residual added in place);
      // the same three products per multiply as the split-on-load kernel
      RUN(linear_x3(m->bb_att, b.proj, m->bb_x, C, false, (int)M, ACT_NONE, st, b.ls1, m->bb_x, C, 2, false));
      RUN(ln(m->bb_x, C, m->bb_xn, 2 * C, 3, b.n2, (int)M, C, 1e-6f, st));
      RUN(linear_x3(m->bb_xn, b.fc1, m->bb_h, 8 * C, true, (int)M, ACT_GELU, st, nullptr, nullptr, 0, 3, false));
      RUN(linear_x3(m->bb_h, b.fc2, m->bb_x, C, false, (int)M, ACT_NONE, st, b.ls2, m->bb_x, C, 4, false));
    } else if (h16) {
      RUN(linear(m->bb_att, C, true, b.proj, m->bb_y, C, true, (int)M, ACT_NONE, st, b.ls1, nullptr, 0, nullptr, 0, 1, nullptr, 0, 2));
      RUN(ln(m->bb_x, C, m->bb_xn, C, hfmt, b.n2, (int)M, C, 1e-6f, st, 0, m->bb_y, C, nullptr, false));
      RUN(linear(m->bb_xn, C, true, b.fc1, m->bb_h, 4 * C, true, (int)M, ACT_GELU, st, nullptr, nullptr, 0, nullptr, 0, 1, nullptr, 0, 3, sched));
      RUN(linear(m->bb_h, 4 * C, true, b.fc2, m->bb_y2, C, true, (int)M, ACT_NONE, st, b.ls2, nullptr, 0, nullptr, 0, 1, nullptr, 0, 4));
      pend = m->bb_y; pend2 = m->bb_y2;
    } else {
      RUN(linear(m->bb_att, C, false, b.proj, m->bb_x, C, false, (int)M, ACT_NONE, st, b.ls1, m->bb_x, C, nullptr, 0, 1, nullptr, 0, 2));
      RUN(ln(m->bb_x, C, m->bb_xn, C, false, b.n2, (int)M, C, 1e-6f, st));
      RUN(linear(m->bb_xn, C, false, b.fc1, m->bb_h, 4 * C, false, (int)M, ACT_GELU, st, nullptr, nullptr, 0, nullptr, 0, 1, nullptr, 0, 3));
      RUN(linear(m->bb_h, 4 * C, false, b.fc2, m->bb_x, C, false, (int)M, ACT_NONE, st, b.ls2, m->bb_x, C, nullptr, 0, 1, nullptr, 0, 4));
    }
  }
  if (m->prof_on) ++m->prof_pass;
  // (pipelined calls, FULL mode: the previous call's head may still be reading the feature buffer - see run_head)
  if (m->feat_read_pending && (feat_out == nullptr || feat_out == m->feat)) {
    EC_HIP(hipStreamWaitEvent(st, m->ev_feat_read, 0));
    EC_HIP(hipStreamWaitEvent(st, m->ev_feat_read_p, 0));
    EC_HIP(hipStreamWaitEvent(st, m->ev_feat_read_q, 0));
    m->feat_read_pending = false;
  }
  RUN(ln(m->bb_x, C, feat_out ? feat_out : m->feat, C, 0, m->bnorm, (int)M, C, 1e-6f, st, T, pend, C, pend2, false, hfmt));
  return 0;
}
static int run_backbone(ec_model* m, const float* const* imgs, int n_src, int n_each, float* feat_out, hipStream_t st) {
  std::vector<int> counts(n_src, n_each);
  return run_backbone(m, imgs, counts.data(), n_src, feat_out, st);
}

// One TransformerDecoderLayer (encoder_decoder.py:584-651) on nb = batch entries.
//   x    [nb*K, d] fp32, row stride ldx (the main decoder keeps x as the left half of [x | qpe], ldx = 2d)
//   mem  [nb, HW, d] with batch stride s_mem (row stride d)
//   adjacency / masks are indexed by (batch % bs) so the S shots of the skeleton head share them.
struct LayerIO {
  float* x; long ldx;
  float* mem; long s_mem;
  const float* adj1; const float* valid; const uint8_t* kmask_fixed;
  const float* bias;   // [nb, nhead, K, K] or null
  int nb, bs;
  bool update_mem;
  const float* kv_pre = nullptr;   // if set: K|V of the image tokens were projected beforehand ([nb, HW, ld_kv_pre], K at +0, V at +E)
  long ld_kv_pre = 0;
  long s_kv_pre = 0;               // batch stride of kv_pre in elements (0: HW * ld_kv_pre)
  bool kv16 = false;               // ... and stored as IEEE fp16 (kv_pre then points at fp16 data; ld_kv_pre in fp16 elements): kv16_on()
  // cross-stream hand-offs (decoder helper stream): the first overwrite of x (LN1) waits for wait_x (a helper is still reading the
  // previous layer's x), the cross-attention waits for wait_ca[] (query positional half of x / pre-projected K|V)
  hipEvent_t wait_x = nullptr;
  hipEvent_t wait_ca[2] = {nullptr, nullptr};
  hipEvent_t wait_kv = nullptr;   // pre-projected K|V ready: waited for AFTER the query projection, right before the cross attention
  hipEvent_t wait_sa = nullptr;   // adjacency / attention bias ready: waited for after the self-attention input projection
  hipEvent_t sa_done = nullptr;   // the layer's self-attention was launched beforehand on another stream (run_self_attention): skip it,
                                  // wait for this event before the first chain reads its output
  // row-chain mode (ec_chain.hip): the layer's last chain (ffn2 + norm3) also produces what the NEXT consumers of x need
  bool qkv_ready = false;          // this layer's self-attention input projection was produced by the previous layer's last chain
  const Lin* next_sa_in = nullptr; // next layer's self-attention input projection -> qkv
  bool kvk_in_chain = false;       // two-way layers: K|V of the image->token attention (i2t_kv(x)) -> kvk
  // two workgroups per slab (ChainP::split): the token state ping-pongs between x and x_alt, one hop per chain, because the part
  // that shares a residual stage may still be reading x while the other part stores the new x.  x_final / ldx_final receive the
  // buffer the layer's output ends up in (x itself without x_alt or when the layer does not chain).
  float* x_alt = nullptr; long ldx_alt = 0;
  float** x_final = nullptr; long* ldx_final = nullptr;
  const float* qpe = nullptr; long ld_qpe = 0;   // main decoder: positional half of the cross-attention query (default: x + d)
  const ec_model::RowPlan* plan = nullptr;       // row compaction of the layer's chains (nb * K token rows)
};

// K|V of the image tokens for a layer's token->image cross attention, one batch entry per sample (mem may be a strided view):
// kv[nb, HW, 2E], the positional half of K folded into the epilogue table (encoder_decoder.py:604-617).
// The image K|V of a single-pass fp16 layer (head_precision = EC_MIXED) are STORED as IEEE fp16 (round 3): they come out of a GEMM on
// fp16-rounded operands, so the extra rounding is of the size of the error they already carry, the projection writes half the bytes
// (it is bound by its output: the decoder's stacked K|V were 127 MB per step and the most expensive single kernel of the head,
// 171 us of step time) and the cross attention reads half (attn_split_kernel<64, true>).
// The attentions of the single-pass fp16 layers (head_precision = EC_MIXED: skeleton head, decoder layers) with ONE fp16 MFMA per product
// instead of three bf16 ones (AttnP::one).  The encoder's self-attention, on the proposal argmax's path, stays bf16x3.
// (the A/B switches of round 3, EC_ATTN_ONE / EC_KV16, are gone in round 4: the bf16x3 head keeps the three-MFMA attention and the fp32
//  K|V path alive and tested)
static bool attn_one(const ec_model* m, const DecLayer& L) { return m->head_mixed && m->head_split && L.sa_in.h1; }
static bool kv16_on(const ec_model* m, const Lin& kv) { return m->head_mixed && m->head_split && kv.h1 && m->E / m->cfg.nhead == 64; }

static int project_image_kv(ec_model* m, const DecLayer& L, const float* mem, long s_mem, int nb, float* kv, hipStream_t st,
                            const bf16_t* mem16 = nullptr) {
  const int d = m->d, E = m->E, HW = m->HW;
  const int out16 = kv16_on(m, L.ca_kv) ? 1 : 0;     // (kv is then an fp16 buffer)
  if (mem16 && L.ca_kv.wf16 && L.ca_kv.h1 && s_mem == (long)HW * d && (long)nb * HW >= 1024) {
    // single-pass fp16 layer, contiguous image rows and an fp16 copy of them at hand (norm4 wrote it): one [nb * HW, 2E] problem on the
    // backbone's 8-phase 16-bit GEMM (fp32 output + positional table) - the same products and fp32 accumulation as the fp16x1 path of
    // gemm_nt, which rounds A to fp16 in registers
    GemmP q;
    q.A = mem16; q.lda = d; q.ab_bf16 = 1; q.h_f16 = 1;
    q.B = L.ca_kv.wf16; q.ldb = d;
    q.C = kv; q.ldc = 2 * E; q.c_bf16 = out16;
    q.table = L.ca_kv_table; q.ldt = 2 * E; q.period = HW;
    q.M = nb * HW; q.N = 2 * E; q.K = d;
    return gemm_nt(q, st);
  }
  GemmP p;
  p.A = mem; p.lda = d; p.sA = s_mem;
  p.split = L.ca_kv.ws ? (L.ca_kv.h1 ? 2 : 1) : 0; p.B = L.ca_kv.wsel(p.split); p.ldb = d;
  p.C = kv; p.ldc = 2 * E; p.sC = (long)HW * 2 * E; p.c_bf16 = out16;
  p.table = L.ca_kv_table; p.ldt = 2 * E; p.period = HW;
  p.M = HW; p.N = 2 * E; p.K = d; p.batch = nb;
  return gemm_nt(p, st);
}

// image -> token attention of a two-way layer, NO masks (encoder_decoder.py:638-649), in two pieces: the query projection only
// needs the image memory; the rest needs the layer's final token state x.  `x_read` (optional) is recorded once x has been read.
static int image_update_q(ec_model* m, const DecLayer& L, const float* mem, int nb, float* qimg, hipStream_t st,
                          const bf16_t* mem16 = nullptr) {
  if (mem16 && L.i2t_q.wf16 && L.i2t_q.h1 && (long)nb * m->HW >= 1024) {   // (see project_image_kv)
    GemmP q;
    q.A = mem16; q.lda = m->d; q.ab_bf16 = 1; q.h_f16 = 1;
    q.B = L.i2t_q.wf16; q.ldb = m->d; q.bias = L.i2t_q.b;
    q.C = qimg; q.ldc = m->E;
    q.table = L.i2t_q_table; q.ldt = m->E; q.period = m->HW;
    q.M = nb * m->HW; q.N = m->E; q.K = m->d;
    return gemm_nt(q, st);
  }
  return linear(mem, m->d, false, L.i2t_q, qimg, m->E, false, nb * m->HW, ACT_NONE, st, nullptr, nullptr, 0, L.i2t_q_table, m->E, m->HW);
}
static int image_update(ec_model* m, const DecLayer& L, const float* x, long ldx, float* mem, int nb, const float* qimg, float* kvk,
                        float* attimg, float* tmpimg, hipStream_t st, hipEvent_t x_read, bool kvk_ready = false, bf16_t* mem16 = nullptr) {
  const int d = m->d, E = m->E, K = m->K, HW = m->HW, nh = m->cfg.nhead;
  const int Mi = nb * HW, Mk = nb * K;
  if (!kvk_ready) RUN(linear(x, ldx, false, L.i2t_kv, kvk, 2 * E, false, Mk, ACT_NONE, st));   // (else: the layer's last chain wrote it)
  if (x_read) EC_HIP(hipEventRecord(x_read, st));
  AttnP a;
  a.Q = qimg; a.K = kvk; a.V = kvk + E; a.O = attimg;
  a.ldq = E; a.ldk = a.ldv = 2 * E; a.ldo = E;
  a.sQ = (long)HW * E; a.sK = a.sV = (long)K * 2 * E; a.sO = (long)HW * E;
  a.B = nb; a.H = nh; a.Lq = HW; a.Lk = K; a.hd = E / nh;
  a.split = m->head_split ? 1 : 0;   // head throughput mode: bf16x3 MFMAs
  a.one = attn_one(m, L) ? 1 : 0;
  RUN(attention(a, st));
  RUN(linear(attimg, E, false, L.i2t_fold, tmpimg, d, false, Mi, ACT_NONE, st, nullptr, mem, d));
  LnP q;
  q.x = tmpimg; q.ldx = d; q.y = mem; q.ldy = d; q.y_bf16 = 0; q.w = L.n4.w; q.b = L.n4.b; q.rows = Mi; q.cols = d; q.eps = 1e-5f;
  if (mem16) { q.y2 = mem16; q.ldy2 = d; }   // fp16 copy for the next layer's K|V / image-query projections
  return layernorm(q, st);
}

// ---- row chains of a decoder / two-way layer (ec_chain.hip).  LDS operand buffers are laid out by a bump allocator.
struct ChainBuild {
  ChainP p;
  int top = CH_LDS0;
  const ec_model::RowPlan* plan = nullptr;   // token-row chains: compute the plan's rows only, then fill in the copies (ec_ops.h rowplan)
  ChainBuild() {}
  explicit ChainBuild(const ec_model::RowPlan* pl) : plan(pl && pl->plan ? pl : nullptr) {}
  int buf(int k) { const int o = top; top += chain_layout_bytes(k); return o; }
  ChainStage& add() { return p.st[p.n_stages++]; }
  int run(int rows, hipStream_t st, bool may_split = false) {
    p.rows = rows; p.lds_bytes = top;
    p.h1 = p.st[0].h1;   // one arithmetic per chain (run_chain checks that every stage was packed for it)
    // two workgroups per slab while that still fits one round of the chip and there is a stage to deal out
    // (not under a row plan: two workgroups per slab buy latency with duplicated work - both compute the stages that later stages read.
    //  Measured with compaction on, two / one workgroup per slab / no compaction, pipelined, interleaved (profiles/r04_compact_split_ab.txt):
    //  cfg2 5645 / 5645 / 5545 pairs/s, ViT-S/14 @224 11 610 / 11 710 / 11 170 - one per slab is never worse)
    // (r04: under a plan one workgroup per slab was never worse for PIPELINED calls - 5645 / 5645 pairs/s on cfg2, 11 610 / 11 710 on ViT-S
    //  @224, profiles/r04_compact_split_ab.txt; plain calls, where the head's latency counts, keep two: RowPlan::split)
    p.split = ((!plan || plan->split) && may_split && p.n_stages > 1 && ((rows + CH_BM - 1) / CH_BM) * 2 <= 256) ? 2 : 1;
    if (!plan) return run_chain(p, st);
    p.rowmap = plan->rowmap; p.n_active = plan->plan;
    p.fan_base = plan->fan_base; p.fan_bits = plan->fan_bits;   // (the masked rows that are not computed: written in-kernel)
    return run_chain(p, st);
  }
};
static void chain_lin(ChainStage& S, const Lin& W) { S.W = W.wc; S.bias = W.b; S.N = W.N; S.K = W.K; S.k1 = W.K; S.h1 = W.h1 ? 1 : 0; }
static bool chain_ok(const Lin& W) { return W.wc != nullptr; }

// x <- LayerNorm(x + in @ W^T + b): the residual branch shared by the three chains of a layer.  Returns the LDS buffer with x.
static int chain_resid_ln(ChainBuild& cb, const float* in, long ld_in, const Lin& W, const Norm& n, const float* x, long ldx,
                         float* x_out, long ldx_out, bool to_lds) {
  ChainStage& S = cb.add();
  chain_lin(S, W);
  S.g_in = in; S.ld_in = ld_in; S.g_k = W.K; S.g_off = cb.buf(W.K); S.a_off = S.g_off;
  S.resid = x; S.ldr = ldx; S.ln_w = n.w; S.ln_b = n.b; S.eps = 1e-5f;
  S.out = x_out; S.ldo = ldx_out;
  if (to_lds) S.s_off = cb.buf(W.N);
  return S.s_off;
}

// does this layer run as row chains?  (the callers use the same predicate to hand the next in-proj / i2t K|V to the last chain)
static bool layer_chains(const ec_model* m, const DecLayer& L) {
  const int lds_ffn2 = CH_LDS0 + chain_layout_bytes(L.ffn2.K) + chain_layout_bytes(m->d);   // the widest operand: z [32, F]
  return m->head_chain && chain_ok(L.sa_out) && chain_ok(L.ca_q) && chain_ok(L.ca_fold) && chain_ok(L.ffn1) && chain_ok(L.ffn2) &&
         (L.ca_q.K == m->d || L.ca_q.K == 2 * m->d) && lds_ffn2 <= 160 * 1024;   // (ViT-L skeleton layers, F = 1024: separate launches)
}

// Self attention over the K keypoint tokens of every sample (hd = d/nh = 32; encoder_decoder.py:596-603): qkv [nb*K, 3d] -> att [nb*K, d].
static int run_self_attention(ec_model* m, const LayerIO& io, const float* qkv, float* att, hipStream_t st, bool one = false) {
  const int d = m->d, K = m->K, nh = m->cfg.nhead;
  AttnP a;
  a.Q = qkv; a.K = qkv + d; a.V = qkv + 2 * d; a.O = att;
  a.ldq = a.ldk = a.ldv = 3 * d; a.ldo = d;
  a.sQ = a.sK = a.sV = (long)K * 3 * d; a.sO = (long)K * d;
  a.kmask = io.kmask_fixed; a.mask_start = 0; a.mask_len = K; a.mask_mod = io.bs;
  a.bias = io.bias;
  a.B = io.nb; a.H = nh; a.Lq = K; a.Lk = K; a.hd = d / nh;
  a.split = m->head_split ? 1 : 0;   // head throughput mode: bf16x3 MFMAs
  a.one = one ? 1 : 0;
  return attention(a, st);
}

static int run_dec_layer(ec_model* m, const DecLayer& L, const LayerIO& io, bool biased, bool two_way, float* qkv, float* att,
                         float* tmp, float* qc, float* kv, float* y, float* z, float* qimg, float* kvk, float* attimg,
                         float* tmpimg, int F, hipStream_t st) {
  const int d = m->d, E = m->E, K = m->K, HW = m->HW, nh = m->cfg.nhead;
  const int Mk = io.nb * K;
  const bool chain = layer_chains(m, L);
  // the un-chained path normalises from `tmp` into io.x: a token state left in `tmp` by a chained predecessor would alias its scratch
  EC_REQUIRE(chain || io.x != tmp, EC_ERR_STATE, "decoder layer: token state aliases the layer scratch (chained layer followed by an un-chained one)");
  // current / other buffer of the token state (ping-pong only in chain mode with x_alt)
  float* xc = io.x; long lxc = io.ldx;
  float* xo = (chain && io.x_alt) ? io.x_alt : io.x; long lxo = (chain && io.x_alt) ? io.ldx_alt : io.ldx;
  const bool pp = xo != xc;
  auto hop = [&]() { std::swap(xc, xo); std::swap(lxc, lxo); };
  EC_REQUIRE(chain || (!io.qkv_ready && !io.next_sa_in && !io.kvk_in_chain), EC_ERR_STATE, "chain hand-offs on a layer that does not chain");
  // ---- self attention over the K keypoint tokens (hd = d/nh = 32)
  EC_REQUIRE(!io.sa_done || (chain && io.qkv_ready), EC_ERR_STATE, "pre-launched self attention needs the chained input projection");
  if (!io.sa_done) {
    if (!io.qkv_ready) RUN(linear(io.x, io.ldx, false, L.sa_in, qkv, 3 * d, false, Mk, ACT_NONE, st));
    if (io.wait_sa) EC_HIP(hipStreamWaitEvent(st, io.wait_sa, 0));
    RUN(run_self_attention(m, io, qkv, att, st, attn_one(m, L)));
  }
  if (chain) {
    // x = norm1(x + out_proj(att)); qc = q_proj([x | qpe]) - one launch (encoder_decoder.py:596-611)
    // (with the ping-pong token state the layer's input buffer is first overwritten by the SECOND chain; waiting for the helper lane only
    //  there was measured: no gain - 6.82-6.84 vs 6.81-6.84 ms, decoder layers 212-231 vs 201-222 us - so the wait stays here)
    if (io.wait_x) EC_HIP(hipStreamWaitEvent(st, io.wait_x, 0));
    if (io.sa_done) EC_HIP(hipStreamWaitEvent(st, io.sa_done, 0));
    for (hipEvent_t e : io.wait_ca)
      if (e) EC_HIP(hipStreamWaitEvent(st, e, 0));
    ChainBuild cb(io.plan);
    const int bx = chain_resid_ln(cb, att, d, L.sa_out, L.n1, xc, lxc, xo, lxo, true);
    ChainStage& Q = cb.add();
    chain_lin(Q, L.ca_q);
    Q.a_off = bx; Q.k1 = d;
    if (L.ca_q.K == 2 * d) {   // main decoder: the positional half of the query sits beside x in the token rows
      Q.g_in = io.qpe ? io.qpe : io.x + d; Q.ld_in = io.qpe ? io.ld_qpe : io.ldx;
      Q.g_k = d; Q.g_off = cb.p.st[0].g_off; Q.b_off = Q.g_off;   // (att's buffer is free again)
    }
    Q.out = qc; Q.ldo = E;
    RUN(cb.run(Mk, st, pp));
    hop();
  } else {
    RUN(linear(att, d, false, L.sa_out, tmp, d, false, Mk, ACT_NONE, st, nullptr, io.x, io.ldx));
    if (io.wait_x) EC_HIP(hipStreamWaitEvent(st, io.wait_x, 0));
    RUN(ln(tmp, d, io.x, io.ldx, false, L.n1, Mk, d, 1e-5f, st));
    for (hipEvent_t e : io.wait_ca)
      if (e) EC_HIP(hipStreamWaitEvent(st, e, 0));
    // ---- cross attention tokens -> image (hd = E/nh = 64); Q input is [x | init_pos] (K = 2d) in the main decoder
    RUN(linear(io.x, io.ldx, false, L.ca_q, qc, E, false, Mk, ACT_NONE, st));
  }
  {
    const float* kvp = io.kv_pre;
    long ldkv = io.ld_kv_pre;
    bool k16 = io.kv16;
    if (!kvp) {
      RUN(project_image_kv(m, L, io.mem, io.s_mem, io.nb, kv, st));
      kvp = kv; ldkv = 2 * E; k16 = kv16_on(m, L.ca_kv);
    }
    if (io.wait_kv) EC_HIP(hipStreamWaitEvent(st, io.wait_kv, 0));
    AttnP a;
    a.Q = qc; a.K = kvp; a.V = kvp + E; a.O = att;
    if (k16) { a.kv16 = 1; a.V = (const bf16_t*)(const void*)kvp + E; }   // (fp16 K|V: V starts E fp16 elements into the row)
    a.ldq = E; a.ldk = a.ldv = ldkv; a.ldo = E;
    a.sQ = (long)K * E; a.sK = a.sV = (io.kv_pre && io.s_kv_pre) ? io.s_kv_pre : (long)HW * ldkv; a.sO = (long)K * E;
    a.B = io.nb; a.H = nh; a.Lq = K; a.Lk = HW; a.hd = E / nh;
    a.split = m->head_split ? 1 : 0;   // head throughput mode: bf16x3 MFMAs
    a.one = attn_one(m, L) ? 1 : 0;
    RUN(attention(a, st));
  }
  // ---- GCN feed-forward (encoder_decoder.py:508-524,634-637): y = conv1d(x) -> [.., 2F];
  //      z = relu(valid * y[:, :F] + adj1 @ y[:, F:]);  x = LN3(x + ffn2(z))
  if (chain) {
    ChainBuild cb(io.plan);   // x = norm2(x + choker(out_proj(att))); y = ffn1(x)
    const int bx = chain_resid_ln(cb, att, E, L.ca_fold, L.n2, xc, lxc, xo, lxo, true);
    ChainStage& Y = cb.add();
    chain_lin(Y, L.ffn1);
    Y.a_off = bx;
    Y.out = y; Y.ldo = 2 * F;
    RUN(cb.run(Mk, st, pp));
    hop();
  } else {
    RUN(linear(att, E, false, L.ca_fold, tmp, d, false, Mk, ACT_NONE, st, nullptr, io.x, io.ldx));
    RUN(ln(tmp, d, io.x, io.ldx, false, L.n2, Mk, d, 1e-5f, st));
    RUN(linear(io.x, io.ldx, false, L.ffn1, y, 2 * F, false, Mk, ACT_NONE, st));
  }
  {
    BgemmP p;
    p.A = io.adj1; p.lda = K; p.sA = (long)K * K; p.modA = io.bs;
    p.B = y + F; p.ldb = 2 * F; p.sB = (long)K * 2 * F; p.transB = 0;
    p.C = z; p.ldc = F; p.sC = (long)K * F;
    p.M = K; p.N = F; p.K = K; p.batch = io.nb;
    p.self = y; p.ld_self = 2 * F; p.s_self = (long)K * 2 * F; p.rowscale = io.valid; p.mod_rs = io.bs; p.relu = 1;
    RUN(bgemm_small(p, st));
  }
  if (chain) {
    ChainBuild cb(io.plan);   // x = norm3(x + ffn2(z)) (-> next layer's self-attention in-proj) (-> image->token K|V)
    const bool more = (io.next_sa_in && chain_ok(*io.next_sa_in)) || (io.kvk_in_chain && chain_ok(L.i2t_kv));
    const int bx = chain_resid_ln(cb, z, F, L.ffn2, L.n3, xc, lxc, xo, lxo, more);
    if (io.next_sa_in && chain_ok(*io.next_sa_in)) {
      ChainStage& S = cb.add();
      chain_lin(S, *io.next_sa_in);
      S.a_off = bx;
      S.out = qkv; S.ldo = 3 * d;
    }
    if (io.kvk_in_chain && chain_ok(L.i2t_kv)) {
      ChainStage& S = cb.add();
      chain_lin(S, L.i2t_kv);
      S.a_off = bx;
      S.out = kvk; S.ldo = 2 * E;
    }
    RUN(cb.run(Mk, st, pp));
    hop();
  } else {
    RUN(linear(z, F, false, L.ffn2, tmp, d, false, Mk, ACT_NONE, st, nullptr, io.x, io.ldx));
    RUN(ln(tmp, d, io.x, io.ldx, false, L.n3, Mk, d, 1e-5f, st));
  }
  if (io.x_final) { *io.x_final = xc; *io.ldx_final = lxc; }
  if (two_way && io.update_mem) {
    RUN(image_update_q(m, L, io.mem, io.nb, qimg, st));
    RUN(image_update(m, L, xc, lxc, io.mem, io.nb, qimg, kvk, attimg, tmpimg, st, nullptr));
  }
  return 0;
}

static int kpt_mlp(ec_model* m, const KptBranch& kb, const float* x, long ldx, int rows, const float* prev, float* out,
                   hipStream_t st, float* t1 = nullptr, float* t2 = nullptr, const ec_model::RowPlan* plan = nullptr) {
  const int d = m->d;
  if (m->head_chain && chain_ok(kb.l0) && chain_ok(kb.l2) && chain_ok(kb.l4) && d == 256 && kb.l0.K == d &&
      kb.l0.N == d && kb.l2.K == d && kb.l2.N == d && kb.l4.K == d && kb.l4.N == d && kb.l0.h1 == kb.l2.h1 && kb.l0.h1 == kb.l4.h1) {
    // the three GELU Linear layers and the keypoint tail (kpt_out) as ONE row chain (see the decoder's helper lane)
    ChainBuild cb(plan);
    const int b0 = cb.buf(d), b1 = cb.buf(d);
    ChainStage& S1 = cb.add();
    chain_lin(S1, kb.l0);
    S1.g_in = x; S1.ld_in = ldx; S1.g_k = d; S1.g_off = b0; S1.a_off = b0; S1.act = ACT_GELU; S1.s_off = b1;
    ChainStage& S2 = cb.add();
    chain_lin(S2, kb.l2);
    S2.a_off = b1; S2.act = ACT_GELU; S2.s_off = b0;
    ChainStage& S3 = cb.add();
    chain_lin(S3, kb.l4);
    S3.a_off = b0; S3.act = ACT_GELU;
    S3.kp_w = kb.w6; S3.kp_b = kb.b6; S3.kp_prev = prev; S3.kp_next = out;
    return cb.run(rows, st);
  }
  if (!t1) { t1 = m->d_k1; t2 = m->d_k2; }   // scratch pair; a second pair lets two branches run on two streams
  RUN(linear(x, ldx, false, kb.l0, t1, d, false, rows, ACT_GELU, st));
  RUN(linear(t1, d, false, kb.l2, t2, d, false, rows, ACT_GELU, st));
  RUN(linear(t2, d, false, kb.l4, t1, d, false, rows, ACT_GELU, st));
  return kpt_out(t1, d, kb.w6, kb.b6, prev, out, rows, d, st);
}

// ---------------------------------------------------------------------------------------------
// Head: TwoStageHead.forward (head.py:161-222).  fq: [bs,HW,C] tokens, fs: S pointers [bs,HW,C].
// ---------------------------------------------------------------------------------------------
// Support-side state of one batch (or of a set of cached episodes): everything the query side of the head needs from the
// support images / heatmaps / skeleton.  SkeletonPredictor.forward takes no query input (head.py:196-200).
struct SupportState {
  float* sk = nullptr;           // [n, K, d]   support keypoint tokens after query_proj        head.py:186-188
  float* valid = nullptr;        // [n, K]
  uint8_t* kmask = nullptr;      // [n, K]      1 = padded keypoint                             head.py:189
  uint8_t* kmask_fixed = nullptr;  // [n, K]    with column 0 un-masked for all-padded samples  encoder_decoder.py:359-360
  float* adj1 = nullptr;         // [n, K, K]   predicted, soft-normalised adjacency            skeleton.py:142-150
  float* adj_out = nullptr;      // [n, 2, K, K]
  float* attn_adj = nullptr;     // [hops+1, n, K, K]                                            skeleton.py:152-161
  float* dec_bias = nullptr;     // optional [dec_layers][n, nhead, K, K]: Markov-bias MLP of every decoder layer (bias_attn.py:188-191),
                                 // computed with the support side (it only depends on attn_adj); nullptr -> the decoder computes it
};

// Support half of TwoStageHead.forward (head.py:175-200): pooling + query_proj + SkeletonPredictor.
// part (FULL mode of a pipelined call, see run_head): 0 everything; 1 only what reads the caller's heatmaps / masks (adjacency build,
// pooling tap lists: no backbone output needed - enqueued BEFORE the call's backbone); 2 everything else (the pooling as a gather
// over the tap lists).
// Row-compaction plans of a batch (ec_ops.h rowplan) from its keypoint mask [bs, K] (non-zero = valid): the decoder's over bs samples,
// the skeleton head's over S * bs (shot-major token rows; one shot: the same plan).
static int build_row_plans(ec_model* m, const float* mask, int bs, int S, hipStream_t st, bool dec = true, bool skel = true) {
  if (!m->compact) return 0;
  m->plan_dec.split = m->plan_skel.split = !m->dq_active;   // two workgroups per slab where the head's latency counts (plain calls)
  if (dec) RUN(rowplan(mask, bs, bs, m->K, m->plan_dec.plan, m->plan_dec.rowmap, m->plan_dec.fan_base, m->plan_dec.fan_bits, st));
  if (skel) {
    // (chosen by the RUNTIME S: a model built for max_shots > 1 and called with one shot reads the decoder's plan - round 4 skipped
    //  both branches in that case and the skeleton head ran on a stale plan)
    if (S == 1 && dec) m->skel_plan = &m->plan_dec;
    else {
      RUN(rowplan(mask, bs, S * bs, m->K, m->plan_skel.plan, m->plan_skel.rowmap, m->plan_skel.fan_base, m->plan_skel.fan_bits, st));
      m->skel_plan = &m->plan_skel;
    }
  }
  return 0;
}

// Markov-bias MLP of every decoder layer (bias_attn.py:188-191): depends on attn_adj only, so it rides with the support side (or, with
// the episode cache, behind the gather of the queries' Markov stacks).  out: [dec_layers][bs, nhead, K, K].
static int decoder_bias_all(ec_model* m, const float* attn_adj, float* out, int bs, hipStream_t st) {
  const int K = m->K, hops1 = m->cfg.max_hops + 1;
  const int nl = (int)m->dec.size();
  const size_t per = (size_t)bs * m->cfg.nhead * K * K;
  if (nl <= 4) {   // all layers in one launch when the fused shape applies
    const float *w1[4], *b1[4], *w2[4], *b2[4];
    for (int li = 0; li < nl; ++li) { w1[li] = m->dec[li].m_w1; b1[li] = m->dec[li].m_b1; w2[li] = m->dec[li].m_w2; b2[li] = m->dec[li].m_b2; }
    const int rc = bias_mlp_layers(attn_adj, w1, b1, w2, b2, nl, out, (long)per, hops1, m->cfg.max_hops + m->cfg.nhead, m->cfg.nhead, bs, K, st);
    if (rc < 0) return rc;
    if (rc == 1) return 0;
  }
  for (int li = 0; li < nl; ++li) {
    const DecLayer& Ld = m->dec[li];
    RUN(bias_mlp(attn_adj, Ld.m_w1, Ld.m_b1, Ld.m_w2, Ld.m_b2, out + li * per, hops1, m->cfg.max_hops + m->cfg.nhead, m->cfg.nhead, bs, K, st));
  }
  return 0;
}

typedef std::function<int(hipStream_t)> StreamHook;   // extra work of the episode-cache paths at a fixed point of a lane

static int run_head_support(ec_model* m, const float* const* fs, const float* const* target_s, const float* mask_s, int bs, int S,
                            hipStream_t st, const SupportState& ss, hipEvent_t ev_sk = nullptr, int part = 0,
                            const StreamHook& on_sk = nullptr) {
  const int C = m->C, d = m->d, K = m->K, HW = m->HW, gh = m->gh, gw = m->gw;
  const int Fs = m->cfg.skel_ffn_dim, hops1 = m->cfg.max_hops + 1;
  const int Mk = bs * K, Mi = bs * HW;
  float* adj_out = ss.adj_out;
  float* attn_adj = ss.attn_adj;
  EC_REQUIRE(hops1 <= 5, EC_ERR_ARG, "max_hops > 4 not supported");   // all argument checks happen before the first fork

  if (m->gt_skel && part != 1) {
    // SkeletonPredictor(learn_skeleton=False) (skeleton.py:70-74): adj = normalize_adj of the skeleton edges - the row-normalised binary
    // adjacency adj_build already makes for refine_features (x / (n + 1e-8) and nan_to_num(x / n) are the same fp32 numbers for the
    // integer row sums n of a binary matrix) - no SkeletonPredictor layers, no Markov stack (attn_adj = None)
    for (int s = 0; s < S; ++s) {
      if (part == 2)
        RUN(pool_apply(m->tap_n + (long)s * Mk, m->tap_i + (long)s * Mk * HW, m->tap_w + (long)s * Mk * HW, fs[s], m->pooled,
                       s == 0 ? 0.f : 1.f, bs, K, m->cfg.heatmap_size, gh, gw, C, st));
      else
        RUN(pool_gather(target_s[s], mask_s, 1.f / (float)S, fs[s], m->pooled, s == 0 ? 0.f : 1.f, bs, K, m->cfg.heatmap_size, gh, gw, C, st));
    }
    if (m->ev_feat_read_p) EC_HIP(hipEventRecord(m->ev_feat_read_p, st));
    if (m->ev_feat_read) EC_HIP(hipEventRecord(m->ev_feat_read, st));
    RUN(linear(m->pooled, C, false, m->query_proj, ss.sk, d, false, Mk, ACT_NONE, st));
    m->taps["support_keypoints"] = {ss.sk, (long)Mk * d};
    if (part == 0) {
      RUN(adj_build(m->d_edges, m->d_off, mask_s, ss.valid, ss.kmask, ss.kmask_fixed, m->binary, m->adj_r1, bs, K, st));
      RUN(build_row_plans(m, mask_s, bs, S, st, !m->episode_call, false));
    }
    if (on_sk) RUN(on_sk(st));
    if (ev_sk) EC_HIP(hipEventRecord(ev_sk, st));
    RUN(adj_gt(m->adj_r1, ss.valid, adj_out, ss.adj1, bs, K, st));
    return tl_mark(m, "S.end", st);
  }
  // image lane of the skeleton head (see (3)): forked first so image_project overlaps the pooling chain
  const bool ov2 = m->overlap_dec && m->side2 != nullptr;
  hipStream_t s2 = ov2 ? m->side2 : st;
  hipEvent_t const ev_x = m->ev_sup[0], ev_xr = m->ev_sup[1], ev_kv = m->ev_sup[2], ev_f = m->ev_sup[3], ev_adjb = m->ev_sup[4];
  const int nb = S * bs;
  const int nsk = (int)m->skel.size();
  if (part == 1) {
    RUN(adj_build(m->d_edges, m->d_off, mask_s, ss.valid, ss.kmask, ss.kmask_fixed, m->binary, m->adj_r1, bs, K, st));
    RUN(build_row_plans(m, mask_s, bs, S, st, !m->episode_call, true));
    for (int s = 0; s < S; ++s)
      RUN(pool_taps(target_s[s], mask_s, 1.f / (float)S, m->tap_n + (long)s * Mk, m->tap_i + (long)s * Mk * HW, m->tap_w + (long)s * Mk * HW,
                    bs, K, m->cfg.heatmap_size, gh, gw, st));
    return 0;
  }
  if (ov2) {
    EC_HIP(hipEventRecord(ev_f, st));
    EC_HIP(hipStreamWaitEvent(s2, ev_f, 0));
    if (part == 0) {
      // adjacency from the skeleton edges + key masks (skeleton.py:58-75): needs only the edges and the keypoint mask, so with the helper
      // lane it runs there FIRST, beside the pooling chain, instead of between query_proj and the first layer on the critical lane
      RUN(adj_build(m->d_edges, m->d_off, mask_s, ss.valid, ss.kmask, ss.kmask_fixed, m->binary, m->adj_r1, bs, K, s2));
      RUN(build_row_plans(m, mask_s, bs, S, s2, !m->episode_call, true));
      EC_HIP(hipEventRecord(ev_adjb, s2));
    }
  }
  for (int s = 0; s < S; ++s)
    RUN(linear(fs[s], C, false, m->image_project, m->s_mem + (long)s * Mi * d, d, false, Mi, ACT_NONE, s2));
  if (m->ev_feat_read) EC_HIP(hipEventRecord(m->ev_feat_read, s2));   // (with the pooling on st: the lane's last read of fs)
  if (nsk > 0) {
    // single-pass fp16 layers: an fp16 copy of the projected image memory (norm4 writes the later layers') puts the first layer's
    // K|V and image-query projections on the 8-phase GEMM as well (47 -> 23 us for the K|V)
    bf16_t* mem16 = (m->s_mem16 && m->skel[0].ca_kv.h1 && m->skel[0].ca_kv.wf16) ? m->s_mem16 : nullptr;
    if (mem16) RUN(f32_to_bf16(m->s_mem, mem16, (long)S * Mi * d, s2, 1));
    RUN(project_image_kv(m, m->skel[0], m->s_mem, (long)HW * d, nb, m->s_kv, s2, mem16));
    if (ov2) EC_HIP(hipEventRecord(ev_kv, s2));
    RUN(tl_mark(m, "I.kv0", s2));
    if (nsk > 1) RUN(image_update_q(m, m->skel[0], m->s_mem, nb, m->s_qimg, s2, mem16));
  }

  // (2) support keypoint pooling + query_proj (head.py:175-188)
  // (one fused kernel per shot: tap weights of the 18x18 grid from the heatmap, then the weighted sum of the non-zero cells' feature
  //  rows; the round-1 form - tap-weight matrix + dense batched GEMM - was removed in round 3)
  for (int s = 0; s < S; ++s) {
    if (part == 2)
      RUN(pool_apply(m->tap_n + (long)s * Mk, m->tap_i + (long)s * Mk * HW, m->tap_w + (long)s * Mk * HW, fs[s], m->pooled,
                     s == 0 ? 0.f : 1.f, bs, K, m->cfg.heatmap_size, gh, gw, C, st));
    else
      RUN(pool_gather(target_s[s], mask_s, 1.f / (float)S, fs[s], m->pooled, s == 0 ? 0.f : 1.f, bs, K, m->cfg.heatmap_size, gh, gw, C, st));
  }
  if (m->ev_feat_read_p) EC_HIP(hipEventRecord(m->ev_feat_read_p, st));
  RUN(linear(m->pooled, C, false, m->query_proj, ss.sk, d, false, Mk, ACT_NONE, st));
  m->taps["support_keypoints"] = {ss.sk, (long)Mk * d};
  RUN(tl_mark(m, "S.pooled", st));
  if (part == 2) {}   // (adjacency build: part 1, earlier on this stream)
  else if (ov2) EC_HIP(hipStreamWaitEvent(st, ev_adjb, 0));
  else {
    RUN(adj_build(m->d_edges, m->d_off, mask_s, ss.valid, ss.kmask, ss.kmask_fixed, m->binary, m->adj_r1, bs, K, st));
    RUN(build_row_plans(m, mask_s, bs, S, st, !m->episode_call, true));
  }
  if (on_sk) RUN(on_sk(st));                      // (streaming episodes: the tokens and masks go to their cache slots first)
  if (ev_sk) EC_HIP(hipEventRecord(ev_sk, st));   // support tokens + key masks are ready: the encoder may start

  // (3) skeleton head (skeleton.py:58-161).  Two lanes: the token path of every layer (self-attention, token->image cross
  // attention, GCN feed-forward) stays on st; everything that only touches the image memory - image_project, each layer's K|V
  // and image-query projections, and the image->token update of the previous layer - runs on the helper stream s2, so layer
  // i+1's self-attention block overlaps layer i's image update.  Hand-offs: ev_x (x_{i+1} final -> image update may read it),
  // ev_xr (image update has read x -> LN1 of layer i+1 may overwrite it), ev_kv (K|V of layer i ready -> cross attention).
  for (int s = 0; s < S; ++s) RUN(copy2d(m->s_x + (long)s * Mk * d, d, ss.sk, d, Mk, d, st));
  float* sx = m->s_x;                         // token state: ping-pongs with s_tmp under two-workgroup row chains (LayerIO::x_alt)
  long sx_ld = d;
  for (int i = 0; i < nsk; ++i) {
    LayerIO io;
    io.x = sx; io.ldx = sx_ld; io.mem = m->s_mem; io.s_mem = (long)HW * d;
    io.x_alt = sx == m->s_x ? m->s_tmp : m->s_x; io.ldx_alt = d;
    io.x_final = &sx; io.ldx_final = &sx_ld;
    io.adj1 = m->adj_r1; io.valid = ss.valid; io.kmask_fixed = ss.kmask_fixed; io.bias = nullptr;
    io.nb = nb; io.bs = bs;
    io.plan = m->compact ? m->skel_plan : nullptr;
    io.update_mem = false;                       // done below, on s2
    io.kv_pre = m->s_kv; io.ld_kv_pre = 2 * m->E; io.kv16 = kv16_on(m, m->skel[i].ca_kv);
    if (ov2) {
      io.wait_x = i > 0 ? ev_xr : nullptr;
      io.wait_kv = ev_kv;
    }
    const bool ch = layer_chains(m, m->skel[i]);
    io.qkv_ready = i > 0 && ch && layer_chains(m, m->skel[i - 1]) && chain_ok(m->skel[i].sa_in);
    if (ch && i + 1 < nsk && layer_chains(m, m->skel[i + 1]) && chain_ok(m->skel[i + 1].sa_in)) io.next_sa_in = &m->skel[i + 1].sa_in;
    io.kvk_in_chain = ch && i + 1 < nsk && chain_ok(m->skel[i].i2t_kv);
    RUN(run_dec_layer(m, m->skel[i], io, false, true, m->s_qkv, m->s_att, m->s_tmp, m->s_qc, m->s_kv, m->s_y, m->s_z, m->s_qimg,
                      m->s_kvk, m->s_attimg, m->s_tmpimg, Fs, st));
    RUN(tl_mark(m, i == 0 ? "S.skel0" : i == 1 ? "S.skel1" : "S.skel2", st));
    if (i + 1 < nsk) {   // the last layer's image update is never read (skeleton.py:104-112)
      if (ov2) {
        EC_HIP(hipEventRecord(ev_x, st));
        EC_HIP(hipStreamWaitEvent(s2, ev_x, 0));
      }
      RUN(image_update(m, m->skel[i], sx, d, m->s_mem, nb, m->s_qimg, m->s_kvk, m->s_attimg, m->s_tmpimg, s2, ov2 ? ev_xr : nullptr,
                       io.kvk_in_chain, m->s_mem16));
      RUN(project_image_kv(m, m->skel[i + 1], m->s_mem, (long)HW * d, nb, m->s_kv, s2, m->s_mem16));
      if (ov2) EC_HIP(hipEventRecord(ev_kv, s2));
      RUN(tl_mark(m, i == 0 ? "I.kv1" : "I.kv2", s2));
      if (i + 2 < nsk) RUN(image_update_q(m, m->skel[i + 1], m->s_mem, nb, m->s_qimg, s2, m->s_mem16));
    }
  }
  const float* kp_ref = sx;                  // mean over the shots (skeleton.py:114); one shot: the tokens themselves
  if (S > 1) {
    RUN(mean_over(m->kp_ref, sx, (long)Mk * d, S, (long)Mk * d, st));
    kp_ref = m->kp_ref;
  }
  m->taps["skel_kp_refined"] = {kp_ref, (long)Mk * d};
  {  // Gram matrix of the refined tokens; adj_combine forms the cosine similarity from its diagonal (no row-normalisation launch)
    BgemmP p;
    p.A = kp_ref; p.lda = d; p.sA = (long)K * d;
    p.B = kp_ref; p.ldb = d; p.sB = (long)K * d; p.transB = 1;
    p.C = m->P; p.ldc = K; p.sC = (long)K * K;
    p.M = K; p.N = K; p.K = d; p.batch = bs;
    RUN(bgemm_small(p, st));
  }
  RUN(adj_combine(m->P, m->binary, ss.valid, m->zc_w, m->zc_b, adj_out, ss.adj1, attn_adj, bs, K, st, 1));
  {  // Markov powers: A^2 = A A, A^3 = A^2 A, A^4 = A^2 A^2 (torch.matrix_power's association)
    const long KK = (long)K * K, hop = (long)bs * KK;
    auto mm = [&](const float* a, const float* b, float* c) {
      BgemmP p;
      p.A = a; p.lda = K; p.sA = KK; p.B = b; p.ldb = K; p.sB = KK; p.transB = 0;
      p.C = c; p.ldc = K; p.sC = KK; p.M = K; p.N = K; p.K = K; p.batch = bs;
      return bgemm_small(p, st);
    };
    float* A1 = attn_adj + hop;
    if (hops1 > 2) RUN(mm(A1, A1, attn_adj + 2 * hop));
    if (hops1 > 4) {   // A^3 = A^2 A and A^4 = A^2 A^2 in ONE launch: batch z < bs -> (A^2[z], A[z]) -> hop 3, z >= bs -> (A^2, A^2) -> hop 4
      BgemmP p;        // (hops 1,2 and hops 3,4 are adjacent in the stack, so B and C are plain strided batches of 2*bs)
      p.A = attn_adj + 2 * hop; p.lda = K; p.sA = KK; p.modA = bs;
      p.B = A1; p.ldb = K; p.sB = KK; p.transB = 0;
      p.C = attn_adj + 3 * hop; p.ldc = K; p.sC = KK; p.M = K; p.N = K; p.K = K; p.batch = 2 * bs;
      RUN(bgemm_small(p, st));
    } else if (hops1 > 3) {
      RUN(mm(attn_adj + 2 * hop, A1, attn_adj + 3 * hop));
    }
  }
  if (ss.dec_bias) RUN(decoder_bias_all(m, attn_adj, ss.dec_bias, bs, st));
  RUN(tl_mark(m, "S.end", st));
  return 0;
}

// Query half of TwoStageHead.forward (head.py:169-173, 202-222): input_proj, encoder, proposal generator, decoder, kpt branches.
// after_sk / before_dec (episode cache): run on the query lane once it has waited for wait_sk / wait_adj - they gather the cached
// support tokens + masks, and the adjacency stack (+ the decoder's Markov bias), of every query's episode into `ss`.
static int run_head_query(ec_model* m, const float* fq, int bs, hipStream_t st, const ec_outputs* out, const SupportState& ss,
                          hipEvent_t wait_sk = nullptr, hipEvent_t wait_adj = nullptr, const StreamHook& after_sk = nullptr,
                          const StreamHook& before_dec = nullptr) {
  const int C = m->C, d = m->d, E = m->E, K = m->K, HW = m->HW, L = m->L, nh = m->cfg.nhead;
  const int Fd = m->cfg.ffn_dim, hops1 = m->cfg.max_hops + 1;
  const int Mk = bs * K;
  float* sim = out->similarity_map_dev;
  float* attn_adj = ss.attn_adj;
  float* pts = out->out_points_dev ? out->out_points_dev : m->d_pts;
  const int Me = bs * L;
  float* mem = m->e_x;                       // image tokens of sample b: rows b*L .. b*L+HW-1
  float* kp = m->e_x + (long)HW * d;         // keypoint tokens: rows b*L+HW ..
  const long s_tok = (long)L * d;
  // ---- helper stream (see the stream plan at (6)); the decoder's image K|V projection only needs the encoder output, so it
  // starts beside the proposal generator
  const bool ovd = m->overlap_dec;
  hipStream_t ax = ovd ? m->aux : st;
  auto fork = [&](hipEvent_t e) -> int {   // ax continues after everything enqueued on st so far
    if (!ovd) return 0;
    EC_HIP(hipEventRecord(e, st));
    EC_HIP(hipStreamWaitEvent(ax, e, 0));
    return 0;
  };
  auto mark = [&](hipEvent_t e) -> int {   // a point on ax that st will wait for
    if (ovd) EC_HIP(hipEventRecord(e, ax));
    return 0;
  };
  hipEvent_t const ev_fork = m->ev_aux[0], ev_x = m->ev_aux[1], ev_qpe = m->ev_aux[2], ev_kv = m->ev_aux[3], ev_done = m->ev_aux[4];
  hipEvent_t const ev_sa = m->ev_aux[5];
  const int nL = (int)m->dec.size();
  const bool deferred = m->dq_active && m->dq != nullptr;
  // fp16 K|V (kv16_on): the projection runs on the backbone's 8-phase GEMM over ALL bs*L encoder rows - image and keypoint rows alike,
  // one contiguous [bs*L, d] fp16 operand (a small conversion pass) instead of 32 batch entries of 324 fp32 rows; the keypoint rows'
  // results (24 % of the rows) are never read.  The 2-barrier kernel streamed 590 MB of fp32 / split-packed operands through the LDS
  // for this K = 256 problem (2304 tiles of 128 x 128: 171 us of step time, the head's most expensive kernel); 636 tiles of 256 x 256
  // on 16-bit operands move 163 MB.
  const bool kv_wide = kv16_on(m, m->dec_kv_all) && m->dec_kv_all.wf16 && m->e_x16 && m->dec_kv_table_L && (long)bs * L >= 1024 &&
                       L <= 2 * HW;   // (the fp16 [bs*L, ..] result must fit the buffer sized for fp32 [bs*HW, ..])
  auto image_kv_all = [&]() -> int {
    // the decoder never updates the image memory (two_way_attn=False, encoder_decoder.py:638): project K|V of the image
    // tokens for ALL decoder layers in one GEMM (stacked weights [nL*2E, d], stacked positional tables [HW, nL*2E])
    if (kv_wide) {
      RUN(f32_to_bf16(m->e_x, m->e_x16, (long)Me * d, ax, 1));
      GemmP q;
      q.A = m->e_x16; q.lda = d; q.ab_bf16 = 1; q.h_f16 = 1;
      q.B = m->dec_kv_all.wf16; q.ldb = d;
      q.C = m->d_kv; q.ldc = (long)nL * 2 * E; q.c_bf16 = 1;
      q.table = m->dec_kv_table_L; q.ldt = (long)nL * 2 * E; q.period = L;
      q.M = Me; q.N = nL * 2 * E; q.K = d;
      RUN(gemm_nt(q, ax));
      RUN(mark(ev_kv));
      return tl_mark(m, "A.kv", ax);
    }
    GemmP p;
    p.A = mem; p.lda = d; p.sA = s_tok;
    p.split = m->dec_kv_all.ws ? (m->dec_kv_all.h1 ? 2 : 1) : 0; p.B = m->dec_kv_all.wsel(p.split); p.ldb = d;
    p.C = m->d_kv; p.ldc = (long)nL * 2 * E; p.sC = (long)HW * nL * 2 * E; p.c_bf16 = kv16_on(m, m->dec_kv_all) ? 1 : 0;
    p.table = m->dec_kv_table; p.ldt = (long)nL * 2 * E; p.period = HW;
    p.M = HW; p.N = nL * 2 * E; p.K = d; p.batch = bs;
    RUN(gemm_nt(p, ax));
    RUN(mark(ev_kv));
    return tl_mark(m, "A.kv", ax);
  };
  auto ref_point_embed = [&](const float* bi) -> int {   // qpe = ref_point_head(sine(b))  (encoder_decoder.py:363-371)
    RUN(sincos_coords(bi, m->dim_t, m->d_sc, d, Mk, d / 2, ax));
    RUN(linear(m->d_sc, d, false, m->rp0, m->d_rp, d, false, Mk, ACT_GELU, ax));
    RUN(linear(m->d_rp, d, false, m->rp1, m->d_qin + d, 2 * d, false, Mk, ACT_NONE, ax));
    return 0;
  };

  // (1) input_proj on the query features, written straight into the encoder token buffer [bs, L, d] (rows 0..HW-1)
  {
    GemmP p;
    p.A = fq; p.lda = C; p.sA = (long)HW * C;
    p.split = m->input_proj.ws ? 1 : 0; p.B = m->input_proj.wsel(p.split); p.ldb = C; p.bias = m->input_proj.b;
    p.C = m->e_x; p.ldc = d; p.sC = (long)L * d;
    p.M = HW; p.N = d; p.K = C; p.batch = bs;
    RUN(gemm_nt(p, st));
  }
  if (m->dq_active && m->ev_feat_read_q) EC_HIP(hipEventRecord(m->ev_feat_read_q, st));
  RUN(tl_mark(m, "Q.inproj", st));
  if (wait_sk) EC_HIP(hipStreamWaitEvent(st, wait_sk, 0));
  if (after_sk) RUN(after_sk(st));
  RUN(copy3d(m->e_x + (long)HW * d, d, (long)L * d, ss.sk, d, (long)K * d, bs, K, d, st));

  // (4) encoder (encoder_decoder.py:276-310, 461-483) over [bs, L = HW + K, d]
  bool enc_qkv_ready = false;   // the previous layer's row chain already produced src + pos and this layer's in-proj
  for (size_t i = 0; i < m->enc.size(); ++i) {
    const EncLayer& e = m->enc[i];
    if (!enc_qkv_ready) {
      RUN(add_table(m->e_x, d, m->pos_cat, d, L, Me, d, st));   // src = src + pos, every layer, feeds q,k,v
      RUN(linear(m->e_x, d, false, e.in, m->e_qkv, 3 * d, false, Me, ACT_NONE, st));
    }
    enc_qkv_ready = false;
    AttnP a;
    a.Q = m->e_qkv; a.K = m->e_qkv + d; a.V = m->e_qkv + 2 * d; a.O = m->e_att;
    a.ldq = a.ldk = a.ldv = 3 * d; a.ldo = d;
    a.sQ = a.sK = a.sV = (long)L * 3 * d; a.sO = (long)L * d;
    a.kmask = ss.kmask; a.mask_start = HW; a.mask_len = K; a.mask_mod = 0;
    a.B = bs; a.H = nh; a.Lq = L; a.Lk = L; a.hd = d / nh;
    a.split = m->head_split ? 1 : 0;   // head throughput mode: bf16x3 MFMAs
    // (round 3, measured and not adopted: this attention in single-pass fp16 like the skeleton head's and the decoder's - +0.6 % pairs/s,
    //  23 instead of 22 argmax flips of 20 293 on the conformance set; it sits on the proposal argmax's path and stays bf16x3)
    RUN(attention(a, st));
    // The row-wise rest of the layer (encoder_decoder.py:470-483) as ONE row chain: x1 = norm1(x + out_proj(att)); y = relu(linear1(x1));
    // x = norm2(x1 + linear2(y)) (+ pos -> the next layer's in-proj).  x1 stays in registers (keep) and LDS, y [32, F] in LDS; att is
    // staged into y's buffer (dead by then).  7 launches per layer become 2 and the encoder alone gets 25-30 % faster.  Round 2 measured
    // it 0.5-1 % SLOWER for the step, because its 424 CU-filling workgroups starved the support lane beside it, then the critical one;
    // under ec_forward_pipelined (round 3) the query lane is what the caller's stream waits for - the support lane and the decoder run
    // beside the next backbone - so the encoder's time counts and the support lane's does not.  (Re-measured in round 5 before the A/B
    // switch EC_ENC_CHAIN went: within 1 % either way, profiles/r05_enc_chain_ab.txt; the fp32 head runs the separate launches.)
    const bool enc_chain = m->head_chain && chain_ok(e.out) && chain_ok(e.l1) && chain_ok(e.l2) && e.out.K == d &&
                           e.l1.K == d && e.l2.N == d && CH_LDS0 + chain_layout_bytes(Fd) + chain_layout_bytes(d) <= 160 * 1024;
    if (enc_chain) {
      ChainBuild cb;
      const int by = cb.buf(Fd), bx = cb.buf(d);
      ChainStage& A = cb.add();
      chain_lin(A, e.out);
      A.g_in = m->e_att; A.ld_in = d; A.g_k = d; A.g_off = by; A.a_off = by;
      A.resid = m->e_x; A.ldr = d; A.ln_w = e.n1.w; A.ln_b = e.n1.b; A.eps = 1e-5f;
      A.s_off = bx; A.keep = 1;
      ChainStage& B = cb.add();
      chain_lin(B, e.l1);
      B.a_off = bx; B.act = ACT_RELU; B.s_off = by;
      ChainStage& Cc = cb.add();
      chain_lin(Cc, e.l2);
      Cc.a_off = by; Cc.resid_keep = 1; Cc.ln_w = e.n2.w; Cc.ln_b = e.n2.b; Cc.eps = 1e-5f;
      Cc.out = m->e_x; Cc.ldo = d;
      if (i + 1 < m->enc.size() && chain_ok(m->enc[i + 1].in) && m->enc[i + 1].in.K == d) {
        // ... and the head of the next layer: src = x + pos (stored, and the operand of) its self-attention in-proj
        Cc.post_table = m->pos_cat; Cc.ldpt = d; Cc.post_period = L;
        Cc.s_off = bx;
        ChainStage& D = cb.add();
        chain_lin(D, m->enc[i + 1].in);
        D.a_off = bx;
        D.out = m->e_qkv; D.ldo = 3 * d;
        enc_qkv_ready = true;
      }
      RUN(cb.run(Me, st));
    } else {
      RUN(linear(m->e_att, d, false, e.out, m->e_tmp, d, false, Me, ACT_NONE, st, nullptr, m->e_x, d));
      RUN(ln(m->e_tmp, d, m->e_x, d, false, e.n1, Me, d, 1e-5f, st));
      RUN(linear(m->e_x, d, false, e.l1, m->e_h, Fd, false, Me, ACT_RELU, st));
      RUN(linear(m->e_h, Fd, false, e.l2, m->e_tmp, d, false, Me, ACT_NONE, st, nullptr, m->e_x, d));
      RUN(ln(m->e_tmp, d, m->e_x, d, false, e.n2, Me, d, 1e-5f, st));
    }
    RUN(tl_mark(m, i == 0 ? "Q.enc0" : i == 1 ? "Q.enc1" : "Q.enc2", st));
  }
  m->taps["enc"] = {m->e_x, (long)Me * d};
  RUN(fork(ev_fork));
  RUN(image_kv_all());

  // (5) proposal generator (encoder_decoder.py:49-112)
  {
    GemmP p;
    p.A = kp; p.lda = d; p.sA = s_tok; p.split = m->pg_support.ws ? 1 : 0; p.B = m->pg_support.wsel(p.split); p.ldb = d; p.bias = m->pg_support.b;
    p.C = m->p_fs; p.ldc = d; p.sC = (long)K * d; p.M = K; p.N = d; p.K = d; p.batch = bs;
    RUN(gemm_nt(p, st));
    GemmP q;
    q.A = mem; q.lda = d; q.sA = s_tok; q.split = m->pg_query.ws ? 1 : 0; q.B = m->pg_query.wsel(q.split); q.ldb = d; q.bias = m->pg_query.b;
    q.C = m->p_fq; q.ldc = d; q.sC = (long)HW * d; q.M = HW; q.N = d; q.K = d; q.batch = bs;
    RUN(gemm_nt(q, st));
  }
  RUN(linear(m->p_fs, d, false, m->pg_dyn0, m->p_g1, m->pg_dyn0.N, false, Mk, ACT_RELU, st));
  RUN(linear(m->p_g1, m->pg_dyn0.N, false, m->pg_dyn2, m->p_fs2, d, false, Mk, ACT_TANHGATE, st, nullptr, nullptr, 0, nullptr, 0, 1,
             m->p_fs, d));
  {
    BgemmP p;  // similarity[b, k, hw] = fs'[b,k,:] . fq[b,hw,:]
    p.A = m->p_fs2; p.lda = d; p.sA = (long)K * d;
    p.B = m->p_fq; p.ldb = d; p.sB = (long)HW * d; p.transB = 1;
    p.C = sim; p.ldc = HW; p.sC = (long)K * HW; p.M = K; p.N = HW; p.K = d; p.batch = bs;
    RUN(bgemm_small(p, st));
  }
  RUN(proposals(sim, out->initial_proposals_dev, pts, Mk, m->gh, m->gw, st));   // pts[0] = decoder proposals b_0
  RUN(tl_mark(m, "Q.prop", st));

  // (6) decoder (encoder_decoder.py:330-425): x lives as the left half of d_qin = [x | qpe].
  // Stream plan (ax = helper stream, == st when overlap is off).  With x_l the token state entering layer l and b_l its
  // reference points:  layer l needs x_l at once but qpe_l = ref_point_head(sine(b_l)) and the image K|V only at its
  // cross-attention, and b_{l+1} = update(b_l, kpt_branch[l](x_{l+1})) hangs off the layer's output.  So after layer l the helper
  // runs   dec_norm(x_{l+1}) -> hs[l];  kpt_branch[l](x_{l+1}) -> b_{l+1};  sine -> ref_point_head -> qpe_{l+1};
  //        kpt_branch[l](hs[l]) -> output_kpts[l]   (head.py:216-220)
  // while st runs layer l+1's self-attention block; st waits for "x_{l+1} no longer read" before LN1 overwrites x and for
  // qpe_{l+1} before the cross-attention query projection.
  RUN(fork(ev_fork));                      // proposals (pts[0]) and the encoder output are final
  RUN(ref_point_embed(pts));
  RUN(mark(ev_qpe));
  if (deferred) {   // pipelined call: the decoder phase on its own stream (see ec_model::dq; FULL mode: st is that stream already)
    EC_HIP(hipEventRecord(m->ev_dq_start, st));
    EC_HIP(hipStreamWaitEvent(m->dq, m->ev_dq_start, 0));
    st = m->dq;
    if (!ovd) ax = st;   // (no helper stream: the helper-lane work follows the decoder onto its stream)
  }
  RUN(copy3d(m->d_qin, 2 * d, (long)K * 2 * d, kp, d, s_tok, bs, K, d, st));
  // adjacency / Markov stack from the support side: first needed by layer 0's self-attention kernel (bias, key mask) - its input
  // projection runs before the wait.  Without the precomputed bias stack the bias MLP itself reads attn_adj: wait here.
  if (before_dec) {   // (the hook fills ss.adj1 / attn_adj / dec_bias on THIS stream: nothing left to wait for further down)
    if (wait_adj) EC_HIP(hipStreamWaitEvent(st, wait_adj, 0));
    RUN(before_dec(st));
    wait_adj = nullptr;
  }
  if (wait_adj && !ss.dec_bias) EC_HIP(hipStreamWaitEvent(st, wait_adj, 0));
  RUN(tl_mark(m, "Q.adjwait", st));
  float* dx = m->d_qin;                       // token state: left half of d_qin, ping-ponging with d_tmp under two-workgroup chains
  long dx_ld = 2 * d;
  // Round 3: the CRITICAL chain of a layer boundary stays on ONE stream.  Layer l+1's first chain needs qpe_{l+1}, i.e. the
  // five-stage helper chain on x_{l+1} (~45 us) - that, not the token self-attention, is what the boundary waits for; with the helper
  // chain on the helper stream every boundary paid two cross-stream hand-offs (~12 + ~16 us of signal latency, r03_step_trace.txt).
  // Now the helper chain runs on st right behind the layer, and what has slack moves to ax instead: layer l+1's self-attention
  // (its q|k|v were produced by layer l's last chain), dec_norm -> hs[l] and the output keypoint branch.
  bool sa_prelaunched = false;                // this layer's self-attention already runs on ax (ev_sa)
  const ec_model::RowPlan* dplan = m->compact ? &m->plan_dec : nullptr;   // row compaction of the decoder's token-row chains
  for (int li = 0; li < nL; ++li) {
    const DecLayer& Ld = m->dec[li];
    float* bi = pts + (long)li * Mk * 2;
    float* lbias = m->d_bias;
    if (!m->markov_bias) lbias = nullptr;   // (attn_bias=False / no Markov stack: plain self-attention with the key-padding mask)
    else if (ss.dec_bias) lbias = ss.dec_bias + li * (size_t)bs * nh * K * K;
    else RUN(bias_mlp(attn_adj, Ld.m_w1, Ld.m_b1, Ld.m_w2, Ld.m_b2, m->d_bias, hops1, m->cfg.max_hops + nh, nh, bs, K, st));
    LayerIO io;
    io.x = dx; io.ldx = dx_ld; io.mem = mem; io.s_mem = s_tok;
    if (dx == m->d_qin) { io.x_alt = m->d_tmp; io.ldx_alt = d; } else { io.x_alt = m->d_qin; io.ldx_alt = 2 * d; }
    io.x_final = &dx; io.ldx_final = &dx_ld;
    io.qpe = m->d_qin + d; io.ld_qpe = 2 * d;
    io.adj1 = ss.adj1; io.valid = ss.valid; io.kmask_fixed = ss.kmask_fixed; io.bias = lbias;
    io.nb = bs; io.bs = bs; io.update_mem = false;
    io.plan = dplan;
    io.kv_pre = m->d_kv + (long)li * 2 * E; io.ld_kv_pre = (long)nL * 2 * E;
    if (kv16_on(m, m->dec_kv_all)) {   // fp16 K|V: the layer's slice starts li * 2E fp16 elements into the row
      io.kv16 = true;
      io.kv_pre = (const float*)(const void*)((const bf16_t*)(const void*)m->d_kv + (long)li * 2 * E);
      if (kv_wide) io.s_kv_pre = (long)L * nL * 2 * E;   // (all bs*L rows were projected: sample b's image rows start at row b*L)
    }
    if (li == 0 && ss.dec_bias) io.wait_sa = wait_adj;
    if (ovd) {
      io.wait_x = li > 0 ? ev_x : nullptr;
      io.wait_ca[0] = sa_prelaunched ? nullptr : ev_qpe;   // (pre-launched: qpe was written on st itself)
      io.wait_kv = li == 0 ? ev_kv : nullptr;
      io.sa_done = sa_prelaunched ? ev_sa : nullptr;
    }
    const bool ch = layer_chains(m, Ld);
    io.qkv_ready = li > 0 && ch && layer_chains(m, m->dec[li - 1]) && chain_ok(Ld.sa_in);
    if (ch && li + 1 < nL && layer_chains(m, m->dec[li + 1]) && chain_ok(m->dec[li + 1].sa_in)) io.next_sa_in = &m->dec[li + 1].sa_in;
    RUN(run_dec_layer(m, Ld, io, true, false, m->d_qkv, m->d_att, m->d_tmp, m->d_qc, m->d_kv, m->d_y, m->d_z, nullptr, nullptr,
                      nullptr, nullptr, Fd, st));
    RUN(tl_mark(m, li == 0 ? "Q.dec0" : li == 1 ? "Q.dec1" : "Q.dec2", st));
    // ---- helper chain of layer li
    RUN(fork(ev_fork));
    float* hs = m->d_hs + (long)li * Mk * d;
    float* bnext = pts + (long)(li + 1) * Mk * 2;
    const bool last_split = ovd && li + 1 == nL;   // after the last layer nothing is left to hide under: the two keypoint
                                                  // branches (on x and on dec_norm(x)) run side by side on st and ax
    const KptBranch& kb = m->kpt[li];
    // b_{l+1} = sigmoid(inverse_sigmoid(b_l) + kpt_branch[l](x))   (un-normed x, :395-402), then qpe_{l+1} = ref_point_head(sine(b_{l+1})).
    // Layer l+1 waits for qpe_{l+1} right after its self-attention kernel (~25 us), so these seven dependent launches (~70 us) were on
    // the decoder's critical path: as ONE row chain (three GELU stages, the keypoint tail + sine embedding inside the kernel, two
    // ref_point_head stages) the helper lane is back under the attention.  (The fp32 head runs the separate launches.)
    const bool kpt_chain = li + 1 < nL && m->head_chain && chain_ok(kb.l0) && chain_ok(kb.l2) && chain_ok(kb.l4) &&
                           chain_ok(m->rp0) && chain_ok(m->rp1) && kb.l0.K == d && kb.l0.N == d && kb.l2.K == d && kb.l2.N == d &&
                           kb.l4.K == d && kb.l4.N == d && m->rp0.K == d && m->rp0.N == d && m->rp1.K == d && d == 256 &&
                           kb.l0.h1 == m->rp0.h1 && kb.l0.h1 == m->rp1.h1 && kb.l0.h1 == kb.l2.h1 && kb.l0.h1 == kb.l4.h1;
    // next layer's self-attention on the helper stream, right behind this layer's last chain (which produced its q|k|v)
    sa_prelaunched = ovd && kpt_chain && ss.dec_bias && ch && layer_chains(m, m->dec[li + 1]) && chain_ok(m->dec[li + 1].sa_in);
    if (sa_prelaunched) {
      LayerIO nio = io;
      nio.bias = ss.dec_bias + (size_t)(li + 1) * bs * nh * K * K;
      RUN(run_self_attention(m, nio, m->d_qkv, m->d_att, ax, attn_one(m, m->dec[li + 1])));
      EC_HIP(hipEventRecord(ev_sa, ax));
    }
    if (kpt_chain) {
      ChainBuild cb(dplan);
      const int b0 = cb.buf(d), b1 = cb.buf(d);
      ChainStage& S1 = cb.add();
      chain_lin(S1, kb.l0);
      S1.g_in = dx; S1.ld_in = dx_ld; S1.g_k = d; S1.g_off = b0; S1.a_off = b0; S1.act = ACT_GELU; S1.s_off = b1;
      ChainStage& S2 = cb.add();
      chain_lin(S2, kb.l2);
      S2.a_off = b1; S2.act = ACT_GELU; S2.s_off = b0;
      ChainStage& S3 = cb.add();
      chain_lin(S3, kb.l4);
      S3.a_off = b0; S3.act = ACT_GELU; S3.s_off = b1;
      S3.kp_w = kb.w6; S3.kp_b = kb.b6; S3.kp_prev = bi; S3.kp_next = bnext; S3.kp_dim_t = m->dim_t;
      ChainStage& S4 = cb.add();
      chain_lin(S4, m->rp0);
      S4.a_off = b1; S4.act = ACT_GELU; S4.s_off = b0;
      ChainStage& S5 = cb.add();
      chain_lin(S5, m->rp1);
      S5.a_off = b0; S5.out = m->d_qin + d; S5.ldo = 2 * d;
      RUN(cb.run(Mk, sa_prelaunched ? st : ax));
      if (!sa_prelaunched) RUN(mark(ev_qpe));
      RUN(ln(dx, dx_ld, hs, d, false, m->dec_norm, Mk, d, 1e-5f, ax));
      RUN(mark(ev_x));                     // x_{l+1} has been read (chain and dec_norm): layer l+1 may overwrite it
    } else if (li + 1 == nL) {
      // last layer: b_L = update(b_{L-1}, kpt_branch(x)) on the helper lane, dec_norm + kpt_branch(hs) beside it (one row chain each)
      RUN(ln(dx, dx_ld, hs, d, false, m->dec_norm, Mk, d, 1e-5f, last_split ? st : ax));
      RUN(kpt_mlp(m, kb, dx, dx_ld, Mk, bi, bnext, ax, nullptr, nullptr, dplan));
      RUN(mark(ev_x));
    } else {
      RUN(ln(dx, dx_ld, hs, d, false, m->dec_norm, Mk, d, 1e-5f, last_split ? st : ax));
      RUN(linear(dx, dx_ld, false, kb.l0, m->d_k1, d, false, Mk, ACT_GELU, ax));
      RUN(mark(ev_x));                       // x_{l+1} has been read: layer l+1 may overwrite it
      RUN(linear(m->d_k1, d, false, kb.l2, m->d_k2, d, false, Mk, ACT_GELU, ax));
      RUN(linear(m->d_k2, d, false, kb.l4, m->d_k1, d, false, Mk, ACT_GELU, ax));
      RUN(kpt_out(m->d_k1, d, kb.w6, kb.b6, bi, bnext, Mk, d, ax));
      if (li + 1 < nL) {
        RUN(ref_point_embed(bnext));
        RUN(mark(ev_qpe));
      }
    }
    // (7) head output of this level (head.py:216-220): kpt_branch[l](hs[l]) on top of out_points[l] = b_l
    if (last_split) RUN(kpt_mlp(m, kb, hs, d, Mk, bi, out->output_kpts_dev + (long)li * Mk * 2, st, m->d_k3, m->d_k4, dplan));
    else RUN(kpt_mlp(m, kb, hs, d, Mk, bi, out->output_kpts_dev + (long)li * Mk * 2, ax, nullptr, nullptr, dplan));
  }
  RUN(tl_mark(m, "A.end", ax));
  if (ovd) {
    EC_HIP(hipEventRecord(ev_done, ax));
    EC_HIP(hipStreamWaitEvent(st, ev_done, 0));
  }
  RUN(tl_mark(m, "Q.end", st));
  if (deferred) {
    EC_HIP(hipEventRecord(m->ev_dq_done, m->dq));
    m->dq_pending = true;
    m->dq_recorded = true;
  }
  m->taps["hs"] = {m->d_hs, (long)m->dec.size() * Mk * d};
  return 0;
}

}  // namespace ec
struct ec_support {
  ec_model* m = nullptr;
  int cap = 0;
  ec::SupportState ss;    // `cap` slots: one cached episode each (attn_adj: [hops+1][cap][K][K])
  ec::SupportState stg;   // staging of the episodes a call encodes, in run_head_support's dense layout; scattered into their slots
  std::vector<char> filled;
  std::vector<void*> owned;
};
namespace ec {

static SupportState workspace_support(ec_model* m, const ec_outputs* out) {
  SupportState ss;
  ss.sk = m->sk; ss.valid = m->valid; ss.kmask = m->kmask; ss.kmask_fixed = m->kmask_fixed; ss.adj1 = m->adj1;
  ss.adj_out = out->adj_dev;
  ss.attn_adj = out->attn_adj_dev ? out->attn_adj_dev : m->attn_adj;
  ss.dec_bias = m->markov_bias ? m->d_bias_all : nullptr;
  return ss;
}

// Error path of the multi-stream head: work already forked onto the helper streams may still read caller-owned buffers
// (features, heatmaps, masks) and write the outputs.  The header promises that nothing of a failed call is still running when it
// returns, so drain the helper streams before reporting the error.
static int join_on_error(ec_model* m, int rc) {
  if (rc == 0) return 0;
  for (hipStream_t s : {m->side, m->side2, m->aux, m->dq})
    if (s) (void)hipStreamSynchronize(s);
  return rc;
}

// TwoStageHead.forward (head.py:161-222).  fq: [bs,HW,C] tokens, fs: S pointers [bs,HW,C].
static int run_head(ec_model* m, const float* fq, const float* const* fs, const float* const* target_s, const float* mask_s,
                    int bs, int S, hipStream_t st, const ec_outputs* out) {
  const SupportState ss = workspace_support(m, out);
  RUN(tl_mark(m, "head", st));
  if (m->overlap) {
    EC_HIP(hipEventRecord(m->ev_fork, st));
    EC_HIP(hipStreamWaitEvent(m->side, m->ev_fork, 0));
    if (m->dq_active && m->dq && m->pipe_full) {   // (m->overlap holds: forward_impl's `full`)
      // Pipelined call, FULL mode (round 3): the whole head of call i - both lanes of phase 1 and the decoder - leaves the caller's
      // stream, which goes straight on to the next backbone.  The query lane runs on the decoder stream from its first kernel; a head
      // waits for the previous call's head (it owns the head workspace) on ITS streams, not on the caller's.  What the caller's stream
      // still waits for: the lanes' reads of the caller's inputs (heatmaps / masks: pooling and adj_build, in front of ev_sk), and -
      // in front of the next backbone's last LayerNorm, run_backbone - their reads of the feature buffer.
      EC_HIP(hipStreamWaitEvent(m->dq, m->ev_fork, 0));
      if (m->dq_pending) EC_HIP(hipStreamWaitEvent(m->dq, m->ev_dq_done, 0));   // (the support lane waited for it in run_head_pre)
      RUN(join_on_error(m, run_head_support(m, fs, target_s, mask_s, bs, S, m->side, ss, m->ev_sk, 2)));
      EC_HIP(hipEventRecord(m->ev_join, m->side));
      RUN(join_on_error(m, run_head_query(m, fq, bs, m->dq, out, ss, m->ev_sk, m->ev_join)));
      EC_HIP(hipStreamWaitEvent(st, m->ev_inputs, 0));   // recorded beside the backbone, long ago: the caller may reuse its inputs
      m->feat_read_pending = true;
      return tl_dump(m);
    }
    RUN(join_on_error(m, run_head_support(m, fs, target_s, mask_s, bs, S, m->side, ss, m->ev_sk)));
    EC_HIP(hipEventRecord(m->ev_join, m->side));
    RUN(join_on_error(m, run_head_query(m, fq, bs, st, out, ss, m->ev_sk, m->ev_join)));   // st joins the side stream before the decoder
    // A pipelined call leaves the support lane and the decoder running when `st` moves on to the next backbone, which ends by
    // overwriting the feature buffer: `st` must at least have seen the lanes' reads of the features and of the caller's inputs
    // (heatmaps / masks: pooling and adj_build, both in front of ev_sk, which the query lane waited for; features: image_project).
    if (m->dq_active) EC_HIP(hipStreamWaitEvent(st, m->ev_feat_read, 0));
    return tl_dump(m);
  }
  RUN(join_on_error(m, run_head_support(m, fs, target_s, mask_s, bs, S, st, ss)));
  RUN(join_on_error(m, run_head_query(m, fq, bs, st, out, ss)));
  return tl_dump(m);
}

// FULL mode of a pipelined call, before its backbone is enqueued: what reads the caller's heatmaps / masks runs on the support lane's
// stream beside the backbone - behind the previous call's head, which owns the head workspace until its decoder is done.
static int run_head_pre(ec_model* m, const float* const* target_s, const float* mask_s, int bs, int S, hipStream_t st, const ec_outputs* out) {
  const SupportState ss = workspace_support(m, out);
  EC_HIP(hipEventRecord(m->ev_call, st));              // the caller's inputs (and the edge upload) are ready in `st`'s order
  EC_HIP(hipStreamWaitEvent(m->side, m->ev_call, 0));
  if (m->dq_pending) EC_HIP(hipStreamWaitEvent(m->side, m->ev_dq_done, 0));
  RUN(join_on_error(m, run_head_support(m, nullptr, target_s, mask_s, bs, S, m->side, ss, nullptr, 1)));
  EC_HIP(hipEventRecord(m->ev_inputs, m->side));
  return 0;
}

// A pipelined call's decoder may still own the head workspace (and write its caller's outputs): every entry point that touches the
// head makes its stream wait for it first.  Cheap when nothing is pending.
static int wait_pending_decoder(ec_model* m, hipStream_t st) {
  if (m->dq && m->dq_pending) {
    EC_HIP(hipStreamWaitEvent(st, m->ev_dq_done, 0));
    m->dq_pending = false;
    m->feat_read_pending = false;   // (the decoder waited for the whole support lane)
  }
  return 0;
}

static int upload_edges(ec_model* m, const int32_t* edges, const int32_t* off, int bs, hipStream_t st) {
  EC_REQUIRE(off && off[0] == 0, EC_ERR_ARG, "edge_offsets must start at 0");
  const int ne = off[bs];
  for (int b = 0; b < bs; ++b) EC_REQUIRE(off[b + 1] >= off[b], EC_ERR_ARG, "edge_offsets must be non-decreasing");
  for (int e = 0; e < 2 * ne; ++e)
    EC_REQUIRE(edges[e] >= 0 && edges[e] < m->K, EC_ERR_ARG, "skeleton edge index out of range [0, K)");  // reference: IndexError
  if (ne > m->edges_cap) {   // grow: the old buffer may still be read by work enqueued earlier on this stream
    const int cap = ne * 2 + 64;
    int32_t *nd = nullptr, *nh = nullptr;
    EC_HIP(hipMalloc((void**)&nd, (size_t)cap * 2 * sizeof(int32_t)));
    if (hipHostMalloc((void**)&nh, (size_t)cap * 2 * sizeof(int32_t), hipHostMallocDefault) != hipSuccess) {
      (void)hipFree(nd);
      return hip_fail(hipErrorOutOfMemory, "hipHostMalloc(edge staging)", __FILE__, __LINE__);
    }
    if (m->d_edges) {
      EC_HIP(hipStreamSynchronize(st));
      (void)hipFree(m->d_edges);
      (void)hipHostFree(m->h_edges);
    }
    m->d_edges = nd; m->h_edges = nh; m->edges_cap = cap;
  }
  // staged through pinned host buffers owned by the model: the copies are truly asynchronous and the caller's arrays may be
  // released as soon as the call returns.  The previous step's copy must have been consumed before the staging is rewritten.
  if (m->ev_edges) EC_HIP(hipEventSynchronize(m->ev_edges));
  else EC_HIP(hipEventCreateWithFlags(&m->ev_edges, hipEventDisableTiming));
  if (ne > 0) {
    memcpy(m->h_edges, edges, (size_t)ne * 2 * sizeof(int32_t));
    EC_HIP(hipMemcpyAsync(m->d_edges, m->h_edges, (size_t)ne * 2 * sizeof(int32_t), hipMemcpyHostToDevice, st));
  }
  memcpy(m->h_off, off, (size_t)(bs + 1) * sizeof(int32_t));
  EC_HIP(hipMemcpyAsync(m->d_off, m->h_off, (size_t)(bs + 1) * sizeof(int32_t), hipMemcpyHostToDevice, st));
  EC_HIP(hipEventRecord(m->ev_edges, st));
  return 0;
}

}  // namespace ec

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

const char* ec_last_error(void) { return g_err.c_str(); }
int ec_version(void) { return EC_ABI_VERSION; }
int ec_abi_sizes(int* config_bytes, int* outputs_bytes) {
  if (config_bytes) *config_bytes = (int)sizeof(ec_config);
  if (outputs_bytes) *outputs_bytes = (int)sizeof(ec_outputs);
  return EC_OK;
}

int ec_create(const ec_config* cfg, ec_handle* out) {
  EC_REQUIRE(cfg && out, EC_ERR_ARG, "null argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    set_error("no HIP device visible: libedgecape_hip has no CPU fallback");
    return EC_ERR_NODEVICE;
  }
  EC_REQUIRE(cfg->patch == 14, EC_ERR_ARG, "patch size must be 14");
  EC_REQUIRE(cfg->embed_dim % 64 == 0 && cfg->embed_dim / cfg->num_heads == 64, EC_ERR_ARG, "backbone head dim must be 64");
  EC_REQUIRE(cfg->d_model == 256 && cfg->nhead == 8, EC_ERR_ARG, "head d_model/nhead must be 256/8");
  // K is dynamic in the reference (target_s[0].shape[1]; 100 in the test configs, the number of clicked points in the demos)
  EC_REQUIRE(cfg->num_kpts > 0 && cfg->num_kpts <= 256, EC_ERR_ARG, "num_kpts must be within 1..256");   // (adjacency kernels: one K x K byte map in LDS, four columns per lane)
  EC_REQUIRE(cfg->max_hops == 4, EC_ERR_ARG, "max_hops must be 4");
  // layer counts: the workspace and the launch plans are sized from them; the helper-stream plan of the decoder carries per-layer
  // timeline names for up to 8 layers.  skel_layers >= 1: the skeleton head's image lane is joined through its first layer.
  EC_REQUIRE(cfg->dec_layers >= 1 && cfg->dec_layers <= 8, EC_ERR_ARG, "num_decoder_layers must be within 1..8");
  EC_REQUIRE(cfg->enc_layers >= 0 && cfg->enc_layers <= 8, EC_ERR_ARG, "num_encoder_layers must be within 0..8");
  EC_REQUIRE(cfg->skel_layers >= 1 && cfg->skel_layers <= 8, EC_ERR_ARG, "skeleton_predictor depth must be within 1..8");
  EC_REQUIRE(cfg->head_precision == EC_F32 || cfg->head_precision == EC_BF16X3 || cfg->head_precision == EC_MIXED, EC_ERR_ARG,
             "head_precision: EC_F32 (exact), EC_BF16X3 (split-bf16 MFMA, fp32-class accuracy) or EC_MIXED (bf16x3 + single-pass fp16)");
  EC_REQUIRE(cfg->max_batch > 0 && cfg->max_shots > 0, EC_ERR_ARG, "max_batch / max_shots must be positive");
  ec_model* m = new ec_model();
  m->cfg = *cfg;
  // inputs are image_size (height) x image_width pixels; image_width = 0: square.  The reference takes any img.shape[-2:]
  // (EdgeCape.py:143); the DINOv2 patch embedding floors both
  m->H = cfg->image_size; m->W = cfg->image_width > 0 ? cfg->image_width : cfg->image_size;
  m->gh = m->H / cfg->patch; m->gw = m->W / cfg->patch;
  m->HW = m->gh * m->gw; m->T = m->HW + 1; m->C = cfg->embed_dim; m->K = cfg->num_kpts; m->d = cfg->d_model;
  m->L = m->HW + m->K; m->E = 2 * m->d;
  EC_REQUIRE((cfg->backbone_precision >= EC_F32 && cfg->backbone_precision <= EC_F16) || cfg->backbone_precision == EC_F16X2, EC_ERR_ARG,
             "backbone_precision: EC_F32 (exact), EC_BF16X3 (split bf16, fp32-class), EC_F16X2 (fp16 + FP8 corrections), EC_BF16 or EC_F16 (16-bit MFMA operands)");
  m->bb_x2 = cfg->backbone_precision == EC_F16X2;
  EC_REQUIRE(!m->bb_x2 || cfg->embed_dim % 128 == 0, EC_ERR_ARG, "backbone_precision EC_F16X2 needs embed_dim % 128 == 0");
  m->bb_split = cfg->backbone_precision == EC_BF16X3 || m->bb_x2;
  m->bb_x3 = m->bb_split && cfg->embed_dim % 128 == 0 && (m->bb_x2 || !(getenv("EC_BB_X3") && atoi(getenv("EC_BB_X3")) == 0));
  // fp16 planes for the bf16x3 mode itself (three fp16 MFMAs per product) were measured at scale in round 6 and NOT adopted
  // (profiles/r06_conformance_fp16planes_*.json: 0 / 0 / 2 / 1 argmax flips on cfg1 / 2 / 4 / 5 against 0 / 1 / 1 / 1 with bf16 planes, -2.4 %
  // pairs/s); the plane format lives on as the fp16x2 mode's patch embedding
  m->bb_x3_f16 = m->bb_x2;
  m->bb16 = cfg->backbone_precision == EC_BF16 || cfg->backbone_precision == EC_F16;
  m->bbf16 = cfg->backbone_precision == EC_F16;
  m->head_split = cfg->head_precision == EC_BF16X3 || cfg->head_precision == EC_MIXED;
  m->head_mixed = cfg->head_precision == EC_MIXED;
  m->head_chain = m->head_split && !(getenv("EC_CHAIN") && atoi(getenv("EC_CHAIN")) == 0);
  EC_REQUIRE(m->gh >= 2 && m->gh <= 32 && m->gw >= 2 && m->gw <= 32, EC_ERR_ARG, "token grid must be within 2..32 in both directions");
  *out = m;
  return EC_OK;
}

int ec_destroy(ec_handle m) {
  if (!m) return EC_OK;
  for (void* p : m->owned) (void)hipFree(p);
  if (m->d_edges) (void)hipFree(m->d_edges);
  if (m->h_edges) (void)hipHostFree(m->h_edges);
  if (m->h_off) (void)hipHostFree(m->h_off);
  if (m->ev_edges) (void)hipEventDestroy(m->ev_edges);
  for (hipEvent_t e : m->prof_ev) (void)hipEventDestroy(e);
  if (m->dq) { (void)hipStreamSynchronize(m->dq); (void)hipStreamDestroy(m->dq); }
  for (hipEvent_t e : {m->ev_dq_start, m->ev_dq_done, m->ev_feat_read, m->ev_feat_read_p, m->ev_feat_read_q, m->ev_inputs, m->ev_call}) if (e) (void)hipEventDestroy(e);
  if (m->side) { (void)hipStreamSynchronize(m->side); (void)hipStreamDestroy(m->side); }
  for (hipEvent_t e : {m->ev_fork, m->ev_sk, m->ev_join}) if (e) (void)hipEventDestroy(e);
  if (m->aux) { (void)hipStreamSynchronize(m->aux); (void)hipStreamDestroy(m->aux); }
  if (m->side2) { (void)hipStreamSynchronize(m->side2); (void)hipStreamDestroy(m->side2); }
  for (hipEvent_t e : m->ev_aux) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : m->ev_sup) if (e) (void)hipEventDestroy(e);
  delete m;
  return EC_OK;
}

int ec_load_tensor(ec_handle m, const char* name, const void* host, const int64_t* shape, int ndim, int dtype) {
  EC_REQUIRE(m && name && host && shape && ndim >= 0 && ndim <= 6, EC_ERR_ARG, "bad argument");
  EC_REQUIRE(!m->finalized, EC_ERR_STATE, "model already finalized");
  Tensor t;
  t.shape.assign(shape, shape + ndim);
  const long n = t.numel();
  t.host.resize(n);
  switch (dtype) {
    case EC_DT_F32: memcpy(t.host.data(), host, n * sizeof(float)); break;
    case EC_DT_F64: for (long i = 0; i < n; ++i) t.host[i] = (float)((const double*)host)[i]; break;
    case EC_DT_BF16: for (long i = 0; i < n; ++i) t.host[i] = bf2f(((const bf16_t*)host)[i]); break;
    case EC_DT_F16: for (long i = 0; i < n; ++i) t.host[i] = (float)((const _Float16*)host)[i]; break;
    default: set_error("unknown dtype"); return EC_ERR_ARG;
  }
  RUN(dalloc(m, &t.dev, (size_t)n));
  EC_HIP(hipMemcpy(t.dev, t.host.data(), n * sizeof(float), hipMemcpyHostToDevice));
  m->tensors[name] = std::move(t);
  return EC_OK;
}

int ec_set_pos_embed(ec_handle m, const float* table, int64_t rows, int64_t cols) {
  EC_REQUIRE(m && table, EC_ERR_ARG, "bad argument");
  EC_REQUIRE(rows == m->T && cols == m->C, EC_ERR_ARG, "pos table must be [1 + gh*gw, C]");
  int64_t shape[2] = {rows, cols};
  return ec_load_tensor(m, "@pos_table", table, shape, 2, EC_DT_F32);
}

int ec_finalize(ec_handle m) {
  EC_REQUIRE(m && !m->finalized, EC_ERR_STATE, "bad handle or already finalized");
  const int C = m->C, d = m->d, K = m->K, HW = m->HW, T = m->T, L = m->L, E = m->E;
  const std::string bp = "encoder_query.", hp = "keypoint_head_module.";
  int rc;
  // ---------------- backbone
  {
    GET(pw, bp + "patch_embed.proj.weight"); GET(pb, bp + "patch_embed.proj.bias");
    EC_REQUIRE(pw->shape[0] == C && pw->numel() == (long)C * 588, EC_ERR_ARG, "patch_embed weight shape");
    std::vector<float> Wp((size_t)C * m->Kp, 0.f);
    for (int n = 0; n < C; ++n) memcpy(&Wp[(size_t)n * m->Kp], &pw->host[(size_t)n * 588], 588 * sizeof(float));
    m->patch.N = C; m->patch.K = m->Kp; m->patch.b = pb->dev;
    if ((rc = upload(m, Wp, &m->patch.w))) return rc;
    if (m->bb16 && (rc = upload16(m, Wp, &m->patch.w16))) return rc;
    if (m->bbf16) {
      std::vector<bf16_t> w3((size_t)C * 3 * m->Kp);
      for (int n = 0; n < C; ++n)
        for (int k = 0; k < m->Kp; ++k) {
          const float w = Wp[(size_t)n * m->Kp + k];
          const bf16_t h = f2half_host(w);
          const bf16_t l = f2half_host(w - half2f_host(h));
          bf16_t* row = &w3[(size_t)n * 3 * m->Kp];
          row[k] = h; row[m->Kp + k] = h; row[2 * m->Kp + k] = l;
        }
      bf16_t* p3 = nullptr;
      if ((rc = dalloc(m, &p3, w3.size()))) return rc;
      EC_HIP(hipMemcpy(p3, w3.data(), w3.size() * sizeof(bf16_t), hipMemcpyHostToDevice));
      m->patch_w16x3 = p3;
    }
    if (m->bb_x3) { if ((rc = upload_x3(m, Wp.data(), C, m->Kp, &m->patch_w16x3))) return rc; }   // (Kp = 640: ten K-tiles per plane)
    else if (m->bb_split && (rc = upload_split(m, Wp.data(), C, m->Kp, &m->patch.ws))) return rc;
    GET(cls, bp + "cls_token"); GET(pos, "@pos_table");
    m->cls = cls->dev; m->pos = pos->dev;
    if ((rc = make_norm(m, bp + "norm", &m->bnorm))) return rc;
    m->blocks.resize(m->cfg.depth);
    for (int i = 0; i < m->cfg.depth; ++i) {
      const std::string p = bp + "blocks." + std::to_string(i) + ".";
      BBlock& b = m->blocks[i];
      if ((rc = make_norm(m, p + "norm1", &b.n1)) || (rc = make_norm(m, p + "norm2", &b.n2))) return rc;
      if ((rc = make_lin(m, p + "attn.qkv.weight", p + "attn.qkv.bias", &b.qkv, m->bb16))) return rc;
      if ((rc = make_lin(m, p + "attn.proj.weight", p + "attn.proj.bias", &b.proj, m->bb16))) return rc;
      if ((rc = make_lin(m, p + "mlp.fc1.weight", p + "mlp.fc1.bias", &b.fc1, m->bb16))) return rc;
      if ((rc = make_lin(m, p + "mlp.fc2.weight", p + "mlp.fc2.bias", &b.fc2, m->bb16))) return rc;
      EC_REQUIRE(b.qkv.N == 3 * C && b.qkv.K == C && b.fc1.N == 4 * C && b.fc2.K == 4 * C, EC_ERR_ARG, "backbone block shapes");
      GET(l1, p + "ls1.gamma"); GET(l2, p + "ls2.gamma");
      b.ls1 = l1->dev; b.ls2 = l2->dev;
    }
  }
  // ---------------- head
  std::vector<float> dim_t;
  std::vector<float> pos_img = sine_table(m->gh, m->gw, d / 2, &dim_t);
  if ((rc = upload(m, pos_img, &m->pos_img)) || (rc = upload(m, dim_t, &m->dim_t))) return rc;
  {
    std::vector<float> pc((size_t)L * d, 0.f);
    memcpy(pc.data(), pos_img.data(), pos_img.size() * sizeof(float));
    if ((rc = upload(m, pc, &m->pos_cat))) return rc;
  }
  if ((rc = make_lin(m, hp + "input_proj.weight", hp + "input_proj.bias", &m->input_proj, false))) return rc;
  if ((rc = make_lin(m, hp + "query_proj.weight", hp + "query_proj.bias", &m->query_proj, false))) return rc;
  m->gt_skel = m->cfg.gt_skeleton != 0;
  m->markov_bias = !m->gt_skel && m->cfg.no_attn_bias == 0;
  EC_REQUIRE(m->input_proj.K == C && m->query_proj.K == C, EC_ERR_ARG, "head in_channels must equal backbone width");
  if (!m->gt_skel) {   // (learn_skeleton=False never runs refine_features / predict_skeleton: their weights need not be there)
    m->cur_h1 = m->head_mixed;   // the skeleton head's image projection, its two-way layers and the decoder layers: single-pass fp16
    rc = make_lin(m, hp + "skeleton_head.image_project.weight", hp + "skeleton_head.image_project.bias", &m->image_project, false);
    m->cur_h1 = false;
    if (rc) return rc;
    EC_REQUIRE(m->image_project.K == C && m->cfg.skel_ffn_dim == C, EC_ERR_ARG,
               "skeleton_head.dim_feedforward must equal the backbone width (skeleton.py:40,92)");
    GET(zw, hp + "skeleton_head.zero_conv.weight"); GET(zb, hp + "skeleton_head.zero_conv.bias");
    m->zc_w = zw->dev; m->zc_b = zb->dev;
  }
  m->cur_h1 = m->head_mixed;
  struct H1Off { ec_model* m; ~H1Off() { m->cur_h1 = false; } } h1_off{m};   // (every early return below leaves the flag cleared)
  m->skel.resize(m->gt_skel ? 0 : m->cfg.skel_layers);
  for (int i = 0; i < (int)m->skel.size(); ++i)
    if ((rc = build_dec_layer(m, hp + "skeleton_head.skeleton_predictor." + std::to_string(i) + ".", false, true, pos_img, &m->skel[i])))
      return rc;
  m->dec.resize(m->cfg.dec_layers);
  for (int i = 0; i < m->cfg.dec_layers; ++i)
    if ((rc = build_dec_layer(m, hp + "transformer.decoder.layers." + std::to_string(i) + ".", true, false, pos_img, &m->dec[i], m->markov_bias)))
      return rc;
  {  // stack the decoder layers' K|V projections: W [nL*2E, d], table [HW, nL*2E]
    const int nL = m->cfg.dec_layers;
    std::vector<float> W((size_t)nL * 2 * E * d), tb((size_t)HW * nL * 2 * E);
    for (int l = 0; l < nL; ++l) {
      memcpy(&W[(size_t)l * 2 * E * d], m->dec[l].h_kv_w.data(), (size_t)2 * E * d * sizeof(float));
      for (int t = 0; t < HW; ++t)
        memcpy(&tb[((size_t)t * nL + l) * 2 * E], &m->dec[l].h_kv_table[(size_t)t * 2 * E], (size_t)2 * E * sizeof(float));
    }
    if ((rc = make_lin_host(m, W, {}, nL * 2 * E, d, &m->dec_kv_all))) return rc;
    if ((rc = upload(m, tb, &m->dec_kv_table))) return rc;
    {
      std::vector<float> tl((size_t)L * nL * 2 * E, 0.f);
      memcpy(tl.data(), tb.data(), tb.size() * sizeof(float));
      if ((rc = upload(m, tl, &m->dec_kv_table_L))) return rc;
    }
    for (auto& l : m->dec) { std::vector<float>().swap(l.h_kv_w); std::vector<float>().swap(l.h_kv_table); }
    for (auto& l : m->skel) { std::vector<float>().swap(l.h_kv_w); std::vector<float>().swap(l.h_kv_table); }
  }
  m->cur_h1 = false;
  for (auto& l : m->skel) EC_REQUIRE(l.ffn1.N == 2 * m->cfg.skel_ffn_dim, EC_ERR_ARG, "skeleton GCN width mismatch");
  for (auto& l : m->dec) EC_REQUIRE(l.ffn1.N == 2 * m->cfg.ffn_dim, EC_ERR_ARG, "decoder GCN width mismatch");
  m->enc.resize(m->cfg.enc_layers);
  for (int i = 0; i < m->cfg.enc_layers; ++i) {
    const std::string p = hp + "transformer.encoder.layers." + std::to_string(i) + ".";
    EncLayer& e = m->enc[i];
    if ((rc = make_lin(m, p + "self_attn.in_proj_weight", p + "self_attn.in_proj_bias", &e.in, false))) return rc;
    if ((rc = make_lin(m, p + "self_attn.out_proj.weight", p + "self_attn.out_proj.bias", &e.out, false))) return rc;
    if ((rc = make_lin(m, p + "linear1.weight", p + "linear1.bias", &e.l1, false))) return rc;
    if ((rc = make_lin(m, p + "linear2.weight", p + "linear2.bias", &e.l2, false))) return rc;
    if ((rc = make_norm(m, p + "norm1", &e.n1)) || (rc = make_norm(m, p + "norm2", &e.n2))) return rc;
  }
  if ((rc = make_norm(m, hp + "transformer.decoder.norm", &m->dec_norm))) return rc;
  // (round 3, measured and not adopted: the decoder's helper chain - ref_point_head, keypoint branches - in single-pass fp16 under
  // the mixed head: step +0.3 % = noise, median |d kpt| 3.0e-6 -> 4.4e-6; these small MLPs stay bf16x3)
  if ((rc = make_lin(m, hp + "transformer.decoder.ref_point_head.layers.0.weight", hp + "transformer.decoder.ref_point_head.layers.0.bias", &m->rp0, false))) return rc;
  if ((rc = make_lin(m, hp + "transformer.decoder.ref_point_head.layers.1.weight", hp + "transformer.decoder.ref_point_head.layers.1.bias", &m->rp1, false))) return rc;
  const std::string pg = hp + "transformer.proposal_generator.";
  if ((rc = make_lin(m, pg + "support_proj.weight", pg + "support_proj.bias", &m->pg_support, false))) return rc;
  if ((rc = make_lin(m, pg + "query_proj.weight", pg + "query_proj.bias", &m->pg_query, false))) return rc;
  if ((rc = make_lin(m, pg + "dynamic_proj.0.weight", pg + "dynamic_proj.0.bias", &m->pg_dyn0, false))) return rc;
  if ((rc = make_lin(m, pg + "dynamic_proj.2.weight", pg + "dynamic_proj.2.bias", &m->pg_dyn2, false))) return rc;
  m->kpt.resize(m->cfg.dec_layers);
  for (int i = 0; i < m->cfg.dec_layers; ++i) {
    const std::string p = hp + "kpt_branch." + std::to_string(i) + ".mlp.";
    KptBranch& kb = m->kpt[i];
    if ((rc = make_lin(m, p + "0.weight", p + "0.bias", &kb.l0, false))) return rc;
    if ((rc = make_lin(m, p + "2.weight", p + "2.bias", &kb.l2, false))) return rc;
    if ((rc = make_lin(m, p + "4.weight", p + "4.bias", &kb.l4, false))) return rc;
    GET(w6, p + "6.weight"); GET(b6, p + "6.bias");
    kb.w6 = w6->dev; kb.b6 = b6->dev;
  }
  // host copies are no longer needed
  for (auto& kv : m->tensors) std::vector<float>().swap(kv.second.host);

  // ---------------- workspace
  const int bs = m->cfg.max_batch, S = m->cfg.max_shots;
  const int n = (1 + S) * bs;
  m->n_img_max = n;
  const size_t es = m->bb16 ? 2 : 4;
  const size_t MT = (size_t)n * T;
  if ((rc = dalloc(m, &m->bb_x, MT * C))) return rc;
  const size_t es3 = m->bb_x3 ? 4 : es;   // K-concatenated bf16x3: two bf16 planes per activation value
  if ((rc = dmalloc(m, &m->bb_xn, MT * C * es3))) return rc;
  if ((rc = dmalloc(m, &m->bb_qkv, MT * 3 * C * es))) return rc;
  if ((rc = dmalloc(m, &m->bb_att, MT * C * es3))) return rc;
  if (m->bb16 && (rc = dmalloc(m, &m->bb_y, MT * C * 2))) return rc;
  if (m->bb16 && (rc = dmalloc(m, &m->bb_y2, MT * C * 2))) return rc;
  if ((rc = dmalloc(m, &m->bb_h, std::max(MT * 4 * C * es3, MT * 3 * m->Kp * es)))) return rc;
  if ((rc = dalloc(m, &m->feat, (size_t)n * HW * C))) return rc;
  if ((rc = dalloc(m, &m->feat_nchw_tmp, (size_t)n * HW * C))) return rc;
  if ((rc = dalloc(m, &m->d_off, (size_t)bs + 1))) return rc;
  if (getenv("EC_G8_DYN")) m->g8_dyn_mode = atoi(getenv("EC_G8_DYN"));
  if (m->g8_dyn_mode != 0) {
    if ((rc = dalloc(m, &m->g8_sched, 16))) return rc;
    EC_HIP(hipMemset(m->g8_sched, 0, 16 * sizeof(int)));
  }
  EC_HIP(hipHostMalloc((void**)&m->h_off, ((size_t)bs + 1) * sizeof(int32_t), hipHostMallocDefault));
  const size_t Mk = (size_t)bs * K, Mi = (size_t)bs * HW, KK = (size_t)K * K;
  const int Fs = m->cfg.skel_ffn_dim, Fd = m->cfg.ffn_dim;
#define WS(ptr, count) if ((rc = dalloc(m, &m->ptr, (size_t)(count)))) return rc
  WS(tap_n, (size_t)S * Mk); WS(tap_i, (size_t)S * Mk * HW); WS(tap_w, (size_t)S * Mk * HW);
  WS(pooled, Mk * C); WS(sk, Mk * d); WS(valid, Mk); WS(kmask, Mk); WS(kmask_fixed, Mk);
  WS(binary, bs * KK); WS(adj_r1, bs * KK); WS(adj1, bs * KK); WS(P, bs * KK); WS(kn, Mk * d); WS(kp_ref, Mk * d);
  WS(attn_adj, 5 * bs * KK);
  WS(s_mem, S * Mi * d); WS(s_x, S * Mk * d); WS(s_tmp, S * Mk * d); WS(s_qkv, S * Mk * 3 * d); WS(s_att, S * Mk * E);
  WS(s_qc, S * Mk * E); WS(s_kv, S * Mi * 2 * E); WS(s_y, S * Mk * 2 * Fs); WS(s_z, S * Mk * Fs); WS(s_qimg, S * Mi * E);
  WS(s_kvk, S * Mk * 2 * E); WS(s_attimg, S * Mi * E); WS(s_tmpimg, S * Mi * d);
  if (m->head_mixed) WS(s_mem16, S * Mi * d);
  const size_t Me = (size_t)bs * L;
  if (m->head_mixed) WS(e_x16, Me * d);
  WS(e_x, Me * d); WS(e_qkv, Me * 3 * d); WS(e_att, Me * d); WS(e_tmp, Me * d); WS(e_h, Me * Fd);
  WS(p_fs, Mk * d); WS(p_fq, Mi * d); WS(p_g1, Mk * 128); WS(p_fs2, Mk * d);
  WS(d_qin, Mk * 2 * d); WS(d_sc, Mk * d); WS(d_rp, Mk * d); WS(d_bias, (size_t)bs * m->cfg.nhead * KK); WS(d_bias_all, (size_t)m->cfg.dec_layers * bs * m->cfg.nhead * KK); WS(d_qkv, Mk * 3 * d);
  WS(d_att, Mk * E); WS(d_tmp, Mk * d); WS(d_qc, Mk * E); WS(d_kv, Mi * 2 * E * m->cfg.dec_layers); WS(d_y, Mk * 2 * Fd); WS(d_z, Mk * Fd);
  WS(d_hs, (size_t)m->cfg.dec_layers * Mk * d); WS(d_pts, (size_t)(m->cfg.dec_layers + 1) * Mk * 2); WS(d_k1, Mk * d); WS(d_k2, Mk * d); WS(d_k3, Mk * d); WS(d_k4, Mk * d);
#undef WS
  m->compact_mode = (m->head_chain && K <= 128) ? (getenv("EC_COMPACT") ? atoi(getenv("EC_COMPACT")) : 1) : 0;
  if (m->compact_mode) {
    for (int which = 0; which < 2; ++which) {
      ec_model::RowPlan& pl = which ? m->plan_skel : m->plan_dec;
      const size_t rows = (which ? (size_t)S : 1) * Mk;
      if ((rc = dalloc(m, &pl.plan, 4)) || (rc = dalloc(m, &pl.rowmap, rows)) || (rc = dalloc(m, &pl.fan_base, rows)) || (rc = dalloc(m, &pl.fan_bits, 2 * rows))) return rc;
      EC_HIP(hipMemset(pl.plan, 0, 4 * sizeof(int)));
    }
  }
  EC_REQUIRE(m->pg_dyn0.N <= 128, EC_ERR_ARG, "dynamic_proj_dim must be <= 128");
  {
    const char* ov = getenv("EC_OVERLAP");
    m->overlap = !(ov && atoi(ov) == 0);
    m->timeline = getenv("EC_TIMELINE") != nullptr;
    m->timeline_defer = m->timeline && atoi(getenv("EC_TIMELINE")) == 2;
    m->pipe_full = !(getenv("EC_PIPE_FULL") && atoi(getenv("EC_PIPE_FULL")) == 0);
    // the decoder stream of the pipelined entry point (ec_forward_pipelined)
    // (round 3, measured: helper / decoder streams created with a non-default priority - least OR most urgent - wreck the pipelined
    //  step: 6.4 -> 9.6 / 11.2 ms, QKV 0.38 -> 0.34 / 0.22 of peak; profiles/r03_lane_priority_probe.txt.  Default priority only.)
    EC_HIP(hipStreamCreateWithFlags(&m->dq, hipStreamNonBlocking));
    EC_HIP(hipEventCreateWithFlags(&m->ev_dq_start, hipEventDisableTiming));
    EC_HIP(hipEventCreateWithFlags(&m->ev_dq_done, hipEventDisableTiming));
    EC_HIP(hipEventCreateWithFlags(&m->ev_feat_read, hipEventDisableTiming));
    EC_HIP(hipEventCreateWithFlags(&m->ev_feat_read_p, hipEventDisableTiming));
    EC_HIP(hipEventCreateWithFlags(&m->ev_feat_read_q, hipEventDisableTiming));
    EC_HIP(hipEventCreateWithFlags(&m->ev_inputs, hipEventDisableTiming));
    EC_HIP(hipEventCreateWithFlags(&m->ev_call, hipEventDisableTiming));
    if (m->overlap) {
      // (a high stream priority for the support lane, the longer one, measured nothing - the lanes hold one kernel in flight each)
      EC_HIP(hipStreamCreateWithFlags(&m->side, hipStreamNonBlocking));
      EC_HIP(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
      EC_HIP(hipEventCreateWithFlags(&m->ev_sk, hipEventDisableTiming));
      EC_HIP(hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming));
      m->overlap_dec = !(ov && atoi(ov) == 1);   // EC_OVERLAP=1: support-side overlap only
      if (m->overlap_dec) {
        EC_HIP(hipStreamCreateWithFlags(&m->aux, hipStreamNonBlocking));
        EC_HIP(hipStreamCreateWithFlags(&m->side2, hipStreamNonBlocking));
        for (hipEvent_t& e : m->ev_aux) EC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (hipEvent_t& e : m->ev_sup) EC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      }
    }
  }
  EC_HIP(hipDeviceSynchronize());
  m->finalized = true;
  return EC_OK;
}

int ec_backbone(ec_handle m, const float* img, int n_img, float* feat, int layout, void* stream) {
  EC_REQUIRE(m && m->finalized, EC_ERR_STATE, "model not finalized");
  EC_REQUIRE(img && feat && n_img > 0 && n_img <= m->n_img_max, EC_ERR_ARG, "bad image batch (n_img <= (1+max_shots)*max_batch)");
  hipStream_t st = (hipStream_t)stream;
  if (layout == EC_LAYOUT_TOKENS) return run_backbone(m, &img, 1, n_img, feat, st);
  RUN(run_backbone(m, &img, 1, n_img, m->feat_nchw_tmp, st));
  return tokens_to_nchw(m->feat_nchw_tmp, feat, n_img, m->C, m->HW, st);
}

static int check_head_args(ec_handle m, int bs, int S, const ec_outputs* out) {
  EC_REQUIRE(m && m->finalized, EC_ERR_STATE, "model not finalized");
  EC_REQUIRE(bs > 0 && bs <= m->cfg.max_batch && S > 0 && S <= m->cfg.max_shots, EC_ERR_ARG, "bs / S exceed the configured maxima");
  EC_REQUIRE(out && out->output_kpts_dev && out->initial_proposals_dev && out->similarity_map_dev && out->adj_dev, EC_ERR_ARG,
             "missing output buffer");
  return 0;
}

int ec_head(ec_handle m, const float* fq, const float* const* fs, int layout, const float* const* target_s, const float* mask_s,
            const int32_t* edges, const int32_t* off, int bs, int S, void* stream, const ec_outputs* out) {
  RUN(check_head_args(m, bs, S, out));
  EC_REQUIRE(fq && fs && target_s && mask_s, EC_ERR_ARG, "null input");
  hipStream_t st = (hipStream_t)stream;
  struct CScope { ec_model* m; ~CScope() { m->compact = false; } } cscope{m};
  m->compact = m->compact_mode == 2;
  RUN(wait_pending_decoder(m, st));
  RUN(upload_edges(m, edges, off, bs, st));
  const size_t per = (size_t)bs * m->HW * m->C;
  std::vector<const float*> fsp(S);
  const float* fqp = fq;
  if (layout == EC_LAYOUT_NCHW) {
    RUN(nchw_to_tokens(fq, m->feat, bs, m->C, m->HW, st));
    fqp = m->feat;
    for (int s = 0; s < S; ++s) {
      RUN(nchw_to_tokens(fs[s], m->feat + (1 + s) * per, bs, m->C, m->HW, st));
      fsp[s] = m->feat + (1 + s) * per;
    }
  } else {
    for (int s = 0; s < S; ++s) fsp[s] = fs[s];
  }
  return run_head(m, fqp, fsp.data(), target_s, mask_s, bs, S, st, out);
}

static int forward_impl(ec_handle m, const float* img_q, const float* const* img_s, const float* const* target_s, const float* mask_s,
                        const int32_t* edges, const int32_t* off, int bs, int S, void* stream, const ec_outputs* out, bool pipelined) {
  RUN(check_head_args(m, bs, S, out));
  EC_REQUIRE(img_q && img_s && target_s && mask_s, EC_ERR_ARG, "null input");
  hipStream_t st = (hipStream_t)stream;
  struct Scope { ec_model* m; ~Scope() { m->dq_active = false; m->compact = false; } } scope{m};
  m->dq_active = pipelined;
  m->compact = m->compact_mode == 2 || (m->compact_mode == 1 && pipelined);   // row compaction of the token chains: see ec_model::compact_mode
  const bool full = pipelined && m->pipe_full && m->overlap && m->dq;
  // (FULL mode: the edge lists are only read by the adjacency build on the support lane's stream; uploading them there keeps the
  //  copy engine's hand-overs - ~50 us between two kernels - out of the caller's stream)
  RUN(upload_edges(m, edges, off, bs, full ? m->side : st));
  const size_t per = (size_t)bs * m->HW * m->C;
  // EdgeCape.extract_features (EdgeCape.py:186-191): the same backbone on the query and on every support image
  std::vector<const float*> srcs(1 + S), fsp(S);
  srcs[0] = img_q;
  for (int s = 0; s < S; ++s) {
    srcs[1 + s] = img_s[s];
    fsp[s] = m->feat + (1 + s) * per;
  }
  // (round 4, measured and not adopted: the same for plain calls - the support lane's input-only kernels on the side stream beside
  //  the call's own backbone: 4781 / 4803 / 4806 vs 4795 / 4790 / 4806 pairs/s through ec_forward, no difference; profiles/r04_head_pre_ab.txt)
  if (full) RUN(run_head_pre(m, target_s, mask_s, bs, S, st, out));
  if (m->timeline_defer) RUN(tl_mark(m, "BB", st));
  RUN(run_backbone(m, srcs.data(), 1 + S, bs, m->feat, st));   // (beside the previous pipelined call's decoder, if one is pending)
  m->taps["feature_q"] = {m->feat, (long)per};
  if (m->timeline_defer) RUN(tl_mark(m, "BBend", st));
  // ... which owns the head workspace until it is done (FULL mode: the next head waits for it on its own streams, run_head)
  if (!full) RUN(wait_pending_decoder(m, st));
  return run_head(m, m->feat, fsp.data(), target_s, mask_s, bs, S, st, out);
}

int ec_forward(ec_handle m, const float* img_q, const float* const* img_s, const float* const* target_s, const float* mask_s,
               const int32_t* edges, const int32_t* off, int bs, int S, void* stream, const ec_outputs* out) {
  return forward_impl(m, img_q, img_s, target_s, mask_s, edges, off, bs, S, stream, out, false);
}

// Pipelined forward (header): as ec_forward, but the decoder phase is left running on the library's decoder stream.
int ec_forward_pipelined(ec_handle m, const float* img_q, const float* const* img_s, const float* const* target_s, const float* mask_s,
                         const int32_t* edges, const int32_t* off, int bs, int S, void* stream, const ec_outputs* out) {
  return forward_impl(m, img_q, img_s, target_s, mask_s, edges, off, bs, S, stream, out, true);
}

int ec_pipeline_flush(ec_handle m, void* stream) {
  EC_REQUIRE(m && m->finalized, EC_ERR_STATE, "model not finalized");
  // ALWAYS waits for the most recent pipelined call's head, also after a plain entry point has cleared dq_pending on ITS stream
  // (wait_pending_decoder): `stream` may be a different one that has never been ordered behind that head.  A wait for a completed
  // event costs nothing on the device.  (dq_pending stays: see header)
  if (m->dq && m->dq_recorded) EC_HIP(hipStreamWaitEvent((hipStream_t)stream, m->ev_dq_done, 0));
  if (m->timeline_defer) return tl_dump(m, true);
  return EC_OK;
}

// ---- support-side episode cache -------------------------------------------------------------
// The reference evaluates one support set against 15 queries (test_dataset.py:93-97) and recomputes the support backbone
// features, pooled support tokens and the whole SkeletonPredictor for every pair; none of that depends on the query
// (head.py:196-200).  ec_support_encode runs it once per episode, ec_forward_cached runs only the query side.
int ec_support_create(ec_handle m, int max_episodes, ec_support_t* out) {
  EC_REQUIRE(m && m->finalized && out, EC_ERR_STATE, "model not finalized");
  EC_REQUIRE(max_episodes > 0, EC_ERR_ARG, "max_episodes must be positive");
  ec_support* c = new ec_support();
  c->m = m; c->cap = max_episodes;
  c->filled.assign(max_episodes, 0);
  const size_t K = m->K, d = m->d, KK = K * K, hops1 = m->cfg.max_hops + 1;
  auto al = [&](void** p, size_t bytes) -> int {
    EC_HIP(hipMalloc(p, bytes));
    c->owned.push_back(*p);
    EC_HIP(hipMemset(*p, 0, bytes));   // (a slot's Markov stack is never written by a gt_skeleton model, yet travels with the slot)
    return 0;
  };
  int rc = 0;
  for (int which = 0; which < 2 && !rc; ++which) {
    SupportState& t = which ? c->stg : c->ss;
    const size_t n = which ? (size_t)std::min(max_episodes, m->cfg.max_batch) : (size_t)max_episodes;   // (a call encodes <= max_batch episodes)
    (rc = al((void**)&t.sk, n * K * d * 4)) || (rc = al((void**)&t.valid, n * K * 4)) || (rc = al((void**)&t.kmask, n * K)) ||
        (rc = al((void**)&t.kmask_fixed, n * K)) || (rc = al((void**)&t.adj1, n * KK * 4)) || (rc = al((void**)&t.adj_out, n * 2 * KK * 4)) ||
        (rc = al((void**)&t.attn_adj, hops1 * n * KK * 4));
  }
  if (rc) {
    for (void* p : c->owned) (void)hipFree(p);
    delete c;
    return rc;
  }
  *out = c;
  return EC_OK;
}

int ec_support_destroy(ec_support_t c) {
  if (!c) return EC_OK;
  for (void* p : c->owned) (void)hipFree(p);
  delete c;
  return EC_OK;
}

// One call of the episode cache, in its general form (ec_forward_episodes): n_new episodes are encoded - their support images ride in
// the SAME backbone pass as the bs query images, so no small-M pass exists - and stored in their cache slots; the query side of the
// head then runs for the bs queries against the slots slot_q[b] (which may have been filled by this very call).  bs = 0: encode only
// (ec_support_encode); n_new = 0: queries only (ec_forward_cached).
static int episodes_impl(ec_handle m, ec_support_t c, const float* const* img_s, const float* const* target_s, const float* mask_s,
                         const int32_t* edges, const int32_t* off, const int32_t* slots, int n_new, int S, const float* img_q,
                         const int32_t* slot_q, int bs, void* stream, const ec_outputs* out, bool pipelined) {
  EC_REQUIRE(m && m->finalized && c && c->m == m, EC_ERR_STATE, "bad handle");
  EC_REQUIRE(n_new >= 0 && bs >= 0 && n_new + bs > 0, EC_ERR_ARG, "nothing to do: no new episode and no query");
  if (n_new > 0) {
    EC_REQUIRE(img_s && target_s && mask_s && slots && off, EC_ERR_ARG, "null input");
    EC_REQUIRE(n_new <= c->cap && n_new <= m->cfg.max_batch && S > 0 && S <= m->cfg.max_shots, EC_ERR_ARG, "n_episodes / S exceed the configured maxima");
    for (int i = 0; i < n_new; ++i) {
      EC_REQUIRE(slots[i] >= 0 && slots[i] < c->cap, EC_ERR_ARG, "episode slot out of range");
      for (int j = 0; j < i; ++j) EC_REQUIRE(slots[j] != slots[i], EC_ERR_ARG, "two new episodes share a cache slot");
    }
  } else {
    S = 0;
  }
  if (bs > 0) {
    RUN(check_head_args(m, bs, 1, out));
    EC_REQUIRE(img_q && slot_q, EC_ERR_ARG, "null input");
    for (int b = 0; b < bs; ++b) {
      EC_REQUIRE(slot_q[b] >= 0 && slot_q[b] < c->cap, EC_ERR_ARG, "episode index out of range");
      bool ok = c->filled[slot_q[b]] != 0;
      for (int i = 0; i < n_new && !ok; ++i) ok = slots[i] == slot_q[b];
      EC_REQUIRE(ok, EC_ERR_STATE, "support cache slot is empty: encode the episode first (ec_support_encode / ec_forward_episodes)");
      // (a slot that was filled earlier AND is re-encoded by this call serves its NEW episode: the scatter precedes the gather)
    }
  }
  EC_REQUIRE(bs + S * n_new <= m->n_img_max, EC_ERR_ARG, "queries + support images of the new episodes exceed (1 + max_shots) * max_batch");
  hipStream_t st = (hipStream_t)stream;
  pipelined = pipelined && bs > 0;
  struct Scope { ec_model* m; ~Scope() { m->dq_active = false; m->compact = false; m->episode_call = false; } } scope{m};
  m->dq_active = pipelined;
  m->episode_call = true;
  m->compact = m->compact_mode == 2 || (m->compact_mode == 1 && pipelined);
  const bool full = pipelined && m->pipe_full && m->overlap && m->dq;
  const int C = m->C, K = m->K, HW = m->HW, hops1 = m->cfg.max_hops + 1;
  const long KK = (long)K * K;
  const size_t per_img = (size_t)HW * C;

  const float* srcs[1 + 16];
  int counts[1 + 16];
  EC_REQUIRE(S <= 16, EC_ERR_ARG, "more than 16 shots");
  srcs[0] = img_q; counts[0] = bs;
  std::vector<const float*> fsp(S);
  for (int s = 0; s < S; ++s) {
    srcs[1 + s] = img_s[s]; counts[1 + s] = n_new;
    fsp[s] = m->feat + ((size_t)bs + (size_t)s * n_new) * per_img;
  }

  // ---- cache traffic (ec_ops.h rows_xfer; the slot numbers travel as kernel arguments)
  const SupportState &cs = c->ss, &sg = c->stg;
  SupportState ws;
  if (bs > 0) {
    ws = workspace_support(m, out);
    // (the decoder's Markov bias is recomputed from the gathered stacks - before_dec below - rather than cached: 8 x the stack's bytes)
  }
  const int32_t *slots_v = slots, *slot_q_v = slot_q;
  auto scatter_a = [&](hipStream_t s2) -> int {   // support tokens + masks of the new episodes -> their slots
    XferP x;
    x.add(cs.sk, sg.sk, (long)K * m->d * 4); x.add(cs.valid, sg.valid, (long)K * 4);
    x.add(cs.kmask, sg.kmask, K); x.add(cs.kmask_fixed, sg.kmask_fixed, K);
    return rows_xfer(x, slots_v, n_new, true, s2);
  };
  auto scatter_b = [&](hipStream_t s2) -> int {   // adjacency + Markov stack
    XferP x;
    x.add(cs.adj1, sg.adj1, KK * 4); x.add(cs.adj_out, sg.adj_out, 2 * KK * 4);
    x.add(cs.attn_adj, sg.attn_adj, KK * 4, hops1, (long)c->cap * KK * 4, (long)n_new * KK * 4);
    return rows_xfer(x, slots_v, n_new, true, s2);
  };
  auto gather_a = [&](hipStream_t s2) -> int {    // every query's episode -> the head workspace
    XferP x;
    x.add(ws.sk, cs.sk, (long)K * m->d * 4); x.add(ws.valid, cs.valid, (long)K * 4);
    x.add(ws.kmask, cs.kmask, K); x.add(ws.kmask_fixed, cs.kmask_fixed, K);
    RUN(rows_xfer(x, slot_q_v, bs, false, s2));
    return build_row_plans(m, ws.valid, bs, 1, s2, true, false);   // the queries' masks: valid[b] = 1 / 0 from their episodes
  };
  auto gather_b = [&](hipStream_t s2) -> int {
    XferP x;
    x.add(ws.adj1, cs.adj1, KK * 4); x.add(ws.adj_out, cs.adj_out, 2 * KK * 4);
    x.add(ws.attn_adj, cs.attn_adj, KK * 4, hops1, (long)bs * KK * 4, (long)c->cap * KK * 4);
    RUN(rows_xfer(x, slot_q_v, bs, false, s2));
    return m->markov_bias ? decoder_bias_all(m, ws.attn_adj, ws.dec_bias, bs, s2) : 0;
  };
  auto mark_filled = [&]() { for (int i = 0; i < n_new; ++i) c->filled[slots[i]] = 1; };

  if (full) {
    // Same lanes as a FULL pipelined ec_forward (run_head / run_head_pre): only the backbone on the caller's stream; what reads the
    // caller's heatmaps / masks beside it on the support lane; the whole head - support lane of the new episodes, query lane of the bs
    // queries - on the library's streams beside the NEXT call's backbone.
    if (n_new > 0) RUN(upload_edges(m, edges, off, n_new, m->side));
    EC_HIP(hipEventRecord(m->ev_call, st));
    EC_HIP(hipStreamWaitEvent(m->side, m->ev_call, 0));
    if (m->dq_pending) EC_HIP(hipStreamWaitEvent(m->side, m->ev_dq_done, 0));   // (also: the previous call's gathers have read the slots)
    if (n_new > 0) RUN(join_on_error(m, run_head_support(m, nullptr, target_s, mask_s, n_new, S, m->side, sg, nullptr, 1)));
    EC_HIP(hipEventRecord(m->ev_inputs, m->side));
    if (m->timeline_defer) RUN(tl_mark(m, "BB", st));
    RUN(run_backbone(m, srcs, counts, 1 + S, m->feat, st));
    if (m->timeline_defer) RUN(tl_mark(m, "BBend", st));
    RUN(tl_mark(m, "head", st));
    EC_HIP(hipEventRecord(m->ev_fork, st));
    EC_HIP(hipStreamWaitEvent(m->side, m->ev_fork, 0));
    EC_HIP(hipStreamWaitEvent(m->dq, m->ev_fork, 0));
    if (m->dq_pending) EC_HIP(hipStreamWaitEvent(m->dq, m->ev_dq_done, 0));
    if (n_new > 0) {
      RUN(join_on_error(m, run_head_support(m, fsp.data(), target_s, mask_s, n_new, S, m->side, sg, m->ev_sk, 2, scatter_a)));
      RUN(join_on_error(m, scatter_b(m->side)));
      EC_HIP(hipEventRecord(m->ev_join, m->side));
    }
    RUN(join_on_error(m, run_head_query(m, m->feat, bs, m->dq, out, ws, n_new > 0 ? m->ev_sk : nullptr, n_new > 0 ? m->ev_join : nullptr,
                                        gather_a, gather_b)));
    EC_HIP(hipStreamWaitEvent(st, m->ev_inputs, 0));
    m->feat_read_pending = true;
    mark_filled();                                   // (only once every enqueue of the call has succeeded: ADVICE r5)
    return tl_dump(m);
  }

  RUN(wait_pending_decoder(m, st));
  if (n_new > 0) RUN(upload_edges(m, edges, off, n_new, st));
  RUN(run_backbone(m, srcs, counts, 1 + S, m->feat, st));
  if (bs > 0) m->taps["feature_q"] = {m->feat, (long)((size_t)bs * per_img)};
  if (bs == 0) RUN(tl_mark(m, "support-only", st));
  else RUN(tl_mark(m, "head", st));
  if (m->overlap && n_new > 0 && bs > 0) {
    EC_HIP(hipEventRecord(m->ev_fork, st));
    EC_HIP(hipStreamWaitEvent(m->side, m->ev_fork, 0));
    RUN(join_on_error(m, run_head_support(m, fsp.data(), target_s, mask_s, n_new, S, m->side, sg, m->ev_sk, 0, scatter_a)));
    RUN(join_on_error(m, scatter_b(m->side)));
    EC_HIP(hipEventRecord(m->ev_join, m->side));
    RUN(join_on_error(m, run_head_query(m, m->feat, bs, st, out, ws, m->ev_sk, m->ev_join, gather_a, gather_b)));
    if (m->dq_active) EC_HIP(hipStreamWaitEvent(st, m->ev_feat_read, 0));   // (see run_head)
    mark_filled();
    return tl_dump(m);
  }
  if (n_new > 0) {
    RUN(join_on_error(m, run_head_support(m, fsp.data(), target_s, mask_s, n_new, S, st, sg, nullptr, 0, scatter_a)));
    RUN(join_on_error(m, scatter_b(st)));
  }
  if (bs > 0) RUN(join_on_error(m, run_head_query(m, m->feat, bs, st, out, ws, nullptr, nullptr, gather_a, gather_b)));
  mark_filled();
  return tl_dump(m);
}

int ec_support_encode(ec_handle m, ec_support_t c, const float* const* img_s, const float* const* target_s, const float* mask_s,
                      const int32_t* edges, const int32_t* off, int n_episodes, int S, void* stream) {
  EC_REQUIRE(m && m->finalized && c && c->m == m, EC_ERR_STATE, "bad handle");
  EC_REQUIRE(n_episodes > 0 && n_episodes <= c->cap, EC_ERR_ARG, "n_episodes / S exceed the configured maxima");
  std::vector<int32_t> slots(n_episodes);
  for (int i = 0; i < n_episodes; ++i) slots[i] = i;
  const std::vector<char> before = c->filled;
  std::fill(c->filled.begin(), c->filled.end(), 0);   // the cache then holds exactly these episodes, slots 0 .. n_episodes - 1
  const int rc = episodes_impl(m, c, img_s, target_s, mask_s, edges, off, slots.data(), n_episodes, S, nullptr, nullptr, 0, stream, nullptr, false);
  if (rc) {
    // a scatter into slots 0 .. n - 1 may already have been enqueued when a later step failed: the episodes that were there are gone,
    // the others are what they were (restoring every flag claimed that partly overwritten slots were still valid)
    c->filled = before;
    for (int i = 0; i < n_episodes; ++i) c->filled[i] = 0;
  }
  return rc;
}

int ec_forward_cached(ec_handle m, ec_support_t c, const float* img_q, const int32_t* episode_of_query, int bs, void* stream,
                      const ec_outputs* out) {
  EC_REQUIRE(bs > 0, EC_ERR_ARG, "ec_forward_cached: bs must be positive");
  return episodes_impl(m, c, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, img_q, episode_of_query, bs, stream, out, false);
}

int ec_forward_episodes(ec_handle m, ec_support_t c, const float* const* img_s, const float* const* target_s, const float* mask_s,
                        const int32_t* edges, const int32_t* off, const int32_t* slots, int n_new, int S, const float* img_q,
                        const int32_t* slot_of_query, int bs, void* stream, const ec_outputs* out, int pipelined) {
  const int rc = episodes_impl(m, c, img_s, target_s, mask_s, edges, off, slots, n_new, S, img_q, slot_of_query, bs, stream, out, pipelined != 0);
  if (rc && c && slots)   // the slots this call targeted may hold partly written state: they are empty until encoded again
    for (int i = 0; i < n_new; ++i)
      if (slots[i] >= 0 && slots[i] < c->cap) c->filled[slots[i]] = 0;
  return rc;
}

// ---- on-device input pipeline ---------------------------------------------------------------
int ec_preprocess_images(const uint8_t* const* src_dev, const int32_t* src_hw, const int64_t* src_pitch, const float* inv_affine,
                         int n, int out_size, const float* mean, const float* stdv, float* out_dev, void* stream) {
  EC_REQUIRE(src_dev && src_hw && inv_affine && mean && stdv && out_dev && n > 0 && out_size > 0, EC_ERR_ARG, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  for (int i0 = 0; i0 < n; i0 += 16) {
    PreprocBatch pb;
    const int nb = std::min(16, n - i0);
    for (int i = 0; i < nb; ++i) {
      EC_REQUIRE(src_dev[i0 + i] && src_hw[2 * (i0 + i)] > 0 && src_hw[2 * (i0 + i) + 1] > 0, EC_ERR_ARG, "bad source image");
      pb.src[i] = src_dev[i0 + i];
      pb.hs[i] = src_hw[2 * (i0 + i)]; pb.ws[i] = src_hw[2 * (i0 + i) + 1];
      pb.pitch[i] = src_pitch ? src_pitch[i0 + i] : (long)pb.ws[i] * 3;
      memcpy(pb.inv[i], inv_affine + 6 * (size_t)(i0 + i), 6 * sizeof(float));
    }
    for (int c = 0; c < 3; ++c) { pb.mean[c] = mean[c]; pb.stdv[c] = stdv[c]; }
    RUN(preprocess_affine(pb, nb, out_dev + (size_t)i0 * 3 * out_size * out_size, out_size, st));
  }
  return EC_OK;
}

// cv::warpAffine's inversion of the src -> dst matrix (no WARP_INVERSE_MAP), float64, every operation rounded on its own
static void cv2_invert_affine(const double* M, double* out) {
#pragma clang fp contract(off)
  double D = M[0] * M[4] - M[1] * M[3];
  D = D != 0 ? 1. / D : 0;
  const double A11 = M[4] * D, A22 = M[0] * D;
  const double m0 = A11, m1 = M[1] * (-D), m3 = M[3] * (-D), m4 = A22;
  const double b1 = -m0 * M[2] - m1 * M[5];
  const double b2 = -m3 * M[2] - m4 * M[5];
  out[0] = m0; out[1] = m1; out[2] = b1; out[3] = m3; out[4] = m4; out[5] = b2;
}

int ec_preprocess_images_cv2(const uint8_t* const* src_dev, const int32_t* src_hw, const int64_t* src_pitch, const double* fwd_affine,
                             int n, int out_size, const float* mean, const float* stdv, float* out_dev, void* stream) {
  EC_REQUIRE(src_dev && src_hw && fwd_affine && mean && stdv && out_dev && n > 0 && out_size > 0, EC_ERR_ARG, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  for (int i0 = 0; i0 < n; i0 += 16) {
    PreprocBatchCv2 pb;
    const int nb = std::min(16, n - i0);
    for (int i = 0; i < nb; ++i) {
      EC_REQUIRE(src_dev[i0 + i] && src_hw[2 * (i0 + i)] > 0 && src_hw[2 * (i0 + i) + 1] > 0, EC_ERR_ARG, "bad source image");
      pb.src[i] = src_dev[i0 + i];
      pb.hs[i] = src_hw[2 * (i0 + i)]; pb.ws[i] = src_hw[2 * (i0 + i) + 1];
      pb.pitch[i] = src_pitch ? src_pitch[i0 + i] : (long)pb.ws[i] * 3;
      cv2_invert_affine(fwd_affine + 6 * (size_t)(i0 + i), pb.minv[i]);
    }
    for (int c = 0; c < 3; ++c) { pb.mean[c] = mean[c]; pb.stdv[c] = stdv[c]; }
    RUN(preprocess_affine_cv2(pb, nb, out_dev + (size_t)i0 * 3 * out_size * out_size, out_size, st));
  }
  return EC_OK;
}

int ec_msra_targets(const float* joints_dev, const float* visible_dev, int n, int K, int image_size, int heatmap_size, int sigma,
                    const float* gauss_host, float* target_dev, float* weight_dev, void* stream) {
  EC_REQUIRE(joints_dev && visible_dev && gauss_host && target_dev && weight_dev && n > 0 && K > 0, EC_ERR_ARG, "bad argument");
  EC_REQUIRE(sigma == 1, EC_ERR_ARG, "sigma must be 1 (configs/test/*.py; 7x7 gaussian)");
  MsraP mp;
  mp.hm = heatmap_size; mp.tmp = 3 * sigma; mp.stride = (double)image_size / (double)heatmap_size;
  memcpy(mp.g, gauss_host, 49 * sizeof(float));
  return msra_targets(joints_dev, visible_dev, target_dev, weight_dev, n * K, mp, (hipStream_t)stream);
}

int ec_profile(ec_handle m, int enable, int max_launches) {
  EC_REQUIRE(m, EC_ERR_ARG, "null handle");
  m->prof_on = enable != 0;
  m->prof_mode = enable;
  m->prof_pass = 0;
  m->prof_used = 0;
  while (enable && m->prof_ev.size() < (size_t)2 * max_launches) {
    hipEvent_t e;
    EC_HIP(hipEventCreate(&e));
    m->prof_ev.push_back(e);
  }
  return EC_OK;
}

int ec_profile_read(ec_handle m, float* total_ms, int* launches) {
  EC_REQUIRE(m && total_ms && launches, EC_ERR_ARG, "null argument");
  float tot = 0.f;
  for (size_t i = 0; i + 1 < m->prof_used; i += 2) {
    EC_HIP(hipEventSynchronize(m->prof_ev[i + 1]));
    float t = 0.f;
    EC_HIP(hipEventElapsedTime(&t, m->prof_ev[i], m->prof_ev[i + 1]));
    tot += t;
  }
  *total_ms = tot;
  *launches = (int)(m->prof_used / 2);
  return EC_OK;
}

int ec_debug_read(ec_handle m, const char* name, float* host_out, int64_t max_elems, int64_t* n_elems) {
  EC_REQUIRE(m && name && n_elems, EC_ERR_ARG, "bad argument");
  auto it = m->taps.find(name);
  if (it == m->taps.end()) { set_error(std::string("unknown tap: ") + name); return EC_ERR_NAME; }
  *n_elems = it->second.second;
  if (!host_out) return EC_OK;
  EC_REQUIRE(max_elems >= *n_elems, EC_ERR_ARG, "host buffer too small");
  EC_HIP(hipDeviceSynchronize());
  EC_HIP(hipMemcpy(host_out, it->second.first, (size_t)*n_elems * sizeof(float), hipMemcpyDeviceToHost));
  return EC_OK;
}

// ------------------------------------------------------------------------------------------------
// single ops
// ------------------------------------------------------------------------------------------------
int ec_op_linear(const float* A, const float* W, const float* bias, const float* gamma, const float* resid, float* C, int M, int N,
                 int K, int act, int precision, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  GemmP p;
  p.M = M; p.N = N; p.K = K; p.lda = K; p.ldb = K; p.ldc = N; p.C = C; p.bias = bias; p.gamma = gamma; p.resid = resid; p.ldr = N;
  p.act = act;
  if (precision == EC_BF16 || precision == EC_F16) {
    const int f16 = precision == EC_F16;
    bf16_t *a16 = nullptr, *w16 = nullptr;
    EC_HIP(hipMalloc((void**)&a16, (size_t)M * K * 2));
    EC_HIP(hipMalloc((void**)&w16, (size_t)N * K * 2));
    int rc = f32_to_bf16(A, a16, (long)M * K, st, f16);
    if (!rc) rc = f32_to_bf16(W, w16, (long)N * K, st, f16);
    p.A = a16; p.B = w16; p.ab_bf16 = 1; p.h_f16 = f16;
    if (!rc) rc = gemm_nt(p, st);
    (void)hipStreamSynchronize(st);
    (void)hipFree(a16); (void)hipFree(w16);
    return rc;
  }
  if (precision == EC_BF16X3 || precision == EC_MIXED) {   // W is packed on the host (as ec_finalize does for the head weights), A stays fp32
    EC_REQUIRE(K % 32 == 0, EC_ERR_ARG, "bf16x3 needs K % 32 == 0");
    std::vector<float> hw((size_t)N * K), packed((size_t)N * K);
    EC_HIP(hipStreamSynchronize(st));
    EC_HIP(hipMemcpy(hw.data(), W, hw.size() * 4, hipMemcpyDeviceToHost));
    if (precision == EC_MIXED) split_pack_weights_h1(hw.data(), N, K, packed.data());   // the single-pass fp16 form of EC_MIXED
    else split_pack_weights(hw.data(), N, K, packed.data());
    float* ws = nullptr;
    EC_HIP(hipMalloc((void**)&ws, packed.size() * 4));
    EC_HIP(hipMemcpy(ws, packed.data(), packed.size() * 4, hipMemcpyHostToDevice));
    p.A = A; p.B = ws; p.split = precision == EC_MIXED ? 2 : 1;
    int rc = gemm_nt(p, st);
    (void)hipStreamSynchronize(st);
    (void)hipFree(ws);
    return rc;
  }
  p.A = A; p.B = W;
  return gemm_nt(p, st);
}

int ec_op_linear_h16(const float* A, const float* W, const float* bias, const float* gamma, uint16_t* C, int M, int N, int K, int act,
                     int precision, int repeats, void* stream) {
  EC_REQUIRE(precision == EC_BF16 || precision == EC_F16, EC_ERR_ARG, "ec_op_linear_h16: precision must be EC_BF16 or EC_F16");
  EC_REQUIRE(bias && repeats >= 1, EC_ERR_ARG, "ec_op_linear_h16: bias is required");
  hipStream_t st = (hipStream_t)stream;
  const int f16 = precision == EC_F16;
  GemmP p;
  p.M = M; p.N = N; p.K = K; p.lda = K; p.ldb = K; p.ldc = N; p.C = C; p.bias = bias; p.gamma = gamma; p.act = act;
  p.ab_bf16 = 1; p.c_bf16 = 1; p.h_f16 = f16;
  bf16_t *a16 = nullptr, *w16 = nullptr;
  EC_HIP(hipMalloc((void**)&a16, (size_t)M * K * 2));
  EC_HIP(hipMalloc((void**)&w16, (size_t)N * K * 2));
  int rc = f32_to_bf16(A, a16, (long)M * K, st, f16);
  if (!rc) rc = f32_to_bf16(W, w16, (long)N * K, st, f16);
  p.A = a16; p.B = w16;
  for (int i = 0; i < repeats && !rc; ++i) rc = gemm_nt(p, st);   // back to back: every launch overwrites C with the same values
  (void)hipStreamSynchronize(st);
  (void)hipFree(a16); (void)hipFree(w16);
  return rc;
}

int ec_op_linear_x2(const float* A, const float* W, const float* bias, const float* gamma, float* C, void* planes, int M, int N, int K, int act,
                    int repeats, void* stream, float* ms) {
  EC_REQUIRE(A && W && bias && repeats >= 1 && M > 0 && N > 0 && K > 0, EC_ERR_ARG, "ec_op_linear_x2: bad argument");
  EC_REQUIRE((act == ACT_GELU) == (planes != nullptr) && (act == ACT_NONE || act == ACT_GELU) && (planes || C) && !(gamma && planes), EC_ERR_ARG,
             "ec_op_linear_x2: act 0 -> fp32 C (gamma: C = (A W^T + b) * gamma + C in place), act 2 -> fp16x2 planes");
  hipStream_t st = (hipStream_t)stream;
  std::vector<float> wh((size_t)N * K);
  EC_HIP(hipStreamSynchronize(st));
  EC_HIP(hipMemcpy(wh.data(), W, wh.size() * 4, hipMemcpyDeviceToHost));
  std::vector<uint8_t> wp;
  int sa = 0;
  RUN(pack_x2_weights(wh.data(), N, K, wp, &sa));
  void *a2 = nullptr, *w2 = nullptr;
  EC_HIP(hipMalloc(&a2, (size_t)M * 4 * K));
  EC_HIP(hipMalloc(&w2, wp.size()));
  EC_HIP(hipMemcpy(w2, wp.data(), wp.size(), hipMemcpyHostToDevice));
  int rc = pack_x2(A, K, a2, 2l * K, M, K, st);
  GemmP p;
  p.A = a2; p.lda = 2l * K; p.B = w2; p.ldb = 2l * K; p.ab_bf16 = 1; p.h_f16 = 1; p.x2 = K / 64; p.x2_sa = sa;
  p.M = M; p.N = N; p.K = 2 * K; p.bias = bias; p.act = act;
  if (planes) { p.C = planes; p.ldc = 2l * N; p.c_x2 = 1; p.tag = 3; }
  else { p.C = C; p.ldc = N; p.gamma = gamma; if (gamma) { p.resid = C; p.ldr = N; p.tag = 4; } else p.tag = 1; }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (ms) { EC_HIP(hipEventCreate(&e0)); EC_HIP(hipEventCreate(&e1)); EC_HIP(hipEventRecord(e0, st)); }
  for (int i = 0; i < repeats && !rc; ++i) rc = gemm_nt(p, st);
  if (ms && !rc) {
    EC_HIP(hipEventRecord(e1, st));
    EC_HIP(hipEventSynchronize(e1));
    float t = 0.f;
    EC_HIP(hipEventElapsedTime(&t, e0, e1));
    *ms = t / (float)repeats;
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipStreamSynchronize(st);
  (void)hipFree(a2); (void)hipFree(w2);
  return rc;
}

int ec_op_gemm_bench(const void* A, const void* W, const float* bias, void* C, int M, int N, int K, int precision, int iters,
                     void* stream, float* ms) {
  EC_REQUIRE(iters > 0 && ms, EC_ERR_ARG, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  GemmP p;
  p.A = A; p.B = W; p.C = C; p.bias = bias; p.M = M; p.N = N; p.K = K; p.lda = K; p.ldb = K; p.ldc = N;
  p.ab_bf16 = precision == EC_BF16 || precision == EC_F16; p.c_bf16 = p.ab_bf16; p.h_f16 = precision == EC_F16;
  p.split = precision == EC_BF16X3;   // timing only: W is interpreted as an already split-packed buffer
  hipEvent_t e0, e1;
  EC_HIP(hipEventCreate(&e0));
  EC_HIP(hipEventCreate(&e1));
  RUN(gemm_nt(p, st));  // warm
  EC_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) RUN(gemm_nt(p, st));
  EC_HIP(hipEventRecord(e1, st));
  EC_HIP(hipEventSynchronize(e1));
  float t = 0.f;
  EC_HIP(hipEventElapsedTime(&t, e0, e1));
  *ms = t / (float)iters;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return EC_OK;
}

int ec_op_bgemm(const float* A, const float* B, float* C, int batch, int M, int N, int K, int transB, void* stream) {
  BgemmP p;
  p.A = A; p.lda = K; p.sA = (long)M * K;
  p.B = B; p.ldb = transB ? K : N; p.sB = (long)N * K;
  p.C = C; p.ldc = N; p.sC = (long)M * N;
  p.M = M; p.N = N; p.K = K; p.batch = batch; p.transB = transB;
  return bgemm_small(p, (hipStream_t)stream);
}

int ec_op_layernorm(const float* x, const float* w, const float* b, float* y, int rows, int cols, float eps, void* stream) {
  LnP p;
  p.x = x; p.ldx = cols; p.y = y; p.ldy = cols; p.w = w; p.b = b; p.rows = rows; p.cols = cols; p.eps = eps;
  return layernorm(p, (hipStream_t)stream);
}

int ec_op_chain(const float* X, int K1, const float* W1, const float* b1, const float* resid, const float* ln1_w, const float* ln1_b,
                float* x1_out, const float* cat, int Kcat, const float* W2, const float* b2, int N2, int act2, const float* table,
                int period, float* out2, const float* W3, const float* b3, const float* ln3_w, const float* ln3_b, float* x3_out,
                int rows, int precision, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  EC_REQUIRE(X && W1 && ln1_w && ln1_b && x1_out && W2 && out2 && rows > 0, EC_ERR_ARG, "ec_op_chain: missing argument");
  EC_REQUIRE(precision == EC_BF16X3 || precision == EC_MIXED, EC_ERR_ARG, "ec_op_chain: precision EC_BF16X3 or EC_MIXED (single-pass fp16)");
  const bool h1 = precision == EC_MIXED;
  EC_REQUIRE(K1 % 128 == 0 && Kcat % 128 == 0 && N2 % 128 == 0 && (Kcat == 0 || cat), EC_ERR_ARG, "ec_op_chain: shapes");
  std::vector<void*> tmp;
  auto pack = [&](const float* Wd, int N, int K, const void** out) -> int {   // device fp32 [N,K] -> chain packing on device
    std::vector<float> h((size_t)N * K), pk((size_t)N * K);
    EC_HIP(hipMemcpy(h.data(), Wd, h.size() * 4, hipMemcpyDeviceToHost));
    pack_chain_weights(h.data(), N, K, pk.data(), h1);
    void* d = nullptr;
    EC_HIP(hipMalloc(&d, pk.size() * 4));
    tmp.push_back(d);
    EC_HIP(hipMemcpy(d, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
    *out = d;
    return 0;
  };
  EC_HIP(hipStreamSynchronize(st));
  ChainP p;
  int top = CH_LDS0;
  auto buf = [&](int k) { const int o = top; top += chain_layout_bytes(k); return o; };
  int rc = 0;
  {
    ChainStage& S = p.st[p.n_stages++];
    rc = pack(W1, 256, K1, &S.W);
    S.bias = b1; S.N = 256; S.K = K1; S.k1 = K1;
    S.g_in = X; S.ld_in = K1; S.g_k = K1; S.g_off = buf(K1); S.a_off = S.g_off;
    S.resid = resid; S.ldr = 256; S.ln_w = ln1_w; S.ln_b = ln1_b;
    S.out = x1_out; S.ldo = 256; S.s_off = buf(256); S.keep = W3 ? 1 : 0;
  }
  if (!rc) {
    ChainStage& S = p.st[p.n_stages++];
    rc = pack(W2, N2, 256 + Kcat, &S.W);
    S.bias = b2; S.N = N2; S.K = 256 + Kcat; S.k1 = 256; S.a_off = p.st[0].s_off;
    if (Kcat) { S.g_in = cat; S.ld_in = Kcat; S.g_k = Kcat; S.g_off = Kcat <= K1 ? p.st[0].g_off : buf(Kcat); S.b_off = S.g_off; }
    S.act = act2; S.table = table; S.ldt = N2; S.period = period > 0 ? period : 1;
    S.out = out2; S.ldo = N2;
    if (W3) S.s_off = buf(N2);
  }
  if (!rc && W3) {
    EC_REQUIRE(ln3_w && ln3_b && x3_out, EC_ERR_ARG, "ec_op_chain: third stage arguments");
    ChainStage& S = p.st[p.n_stages++];
    rc = pack(W3, 256, N2, &S.W);
    S.bias = b3; S.N = 256; S.K = N2; S.k1 = N2; S.a_off = p.st[1].s_off;
    S.resid_keep = 1; S.ln_w = ln3_w; S.ln_b = ln3_b;
    S.out = x3_out; S.ldo = 256;
  }
  p.h1 = h1 ? 1 : 0;
  for (int i = 0; i < p.n_stages; ++i) p.st[i].h1 = p.h1;
  p.rows = rows; p.lds_bytes = top;
  // a residual input that is NOT the output buffer selects the two-workgroups-per-slab form (ChainP::split), as the head uses it
  p.split = (resid && resid != x1_out && ((rows + CH_BM - 1) / CH_BM) * 2 <= 256) ? 2 : 1;
  if (!rc) rc = run_chain(p, st);
  (void)hipStreamSynchronize(st);
  for (void* d : tmp) (void)hipFree(d);
  return rc;
}

int ec_op_attention(const float* q, const float* k, const float* v, const uint8_t* kmask, const float* bias, float* o, int B, int H,
                    int Lq, int Lk, int hd, int precision, void* stream) {
  if (precision == EC_BF16 || precision == EC_F16) {
    // test path: round q,k,v to the 16-bit format on device, run the 16-bit kernel, widen the result
    const int f16 = precision == EC_F16;
    EC_REQUIRE(hd == 64 && !kmask && !bias, EC_ERR_ARG, "ec_op_attention(bf16/fp16): hd = 64, no mask / bias");
    hipStream_t st = (hipStream_t)stream;
    const int E = H * hd;
    bf16_t *q16 = nullptr, *k16 = nullptr, *v16 = nullptr, *o16 = nullptr;
    EC_HIP(hipMalloc((void**)&q16, (size_t)B * Lq * E * 2));
    EC_HIP(hipMalloc((void**)&k16, (size_t)B * Lk * E * 2));
    EC_HIP(hipMalloc((void**)&v16, (size_t)B * Lk * E * 2));
    EC_HIP(hipMalloc((void**)&o16, (size_t)B * Lq * E * 2));
    int rc = f32_to_bf16(q, q16, (long)B * Lq * E, st, f16);
    if (!rc) rc = f32_to_bf16(k, k16, (long)B * Lk * E, st, f16);
    if (!rc) rc = f32_to_bf16(v, v16, (long)B * Lk * E, st, f16);
    AttnP a;
    a.Q = q16; a.K = k16; a.V = v16; a.O = o16;
    a.ldq = a.ldk = a.ldv = a.ldo = E;
    a.sQ = a.sO = (long)Lq * E; a.sK = a.sV = (long)Lk * E;
    a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.hd = hd; a.bf16 = 1; a.f16 = f16;
    if (!rc) rc = attention(a, st);
    (void)hipStreamSynchronize(st);
    if (!rc) {
      std::vector<bf16_t> ho((size_t)B * Lq * E);
      std::vector<float> hf(ho.size());
      EC_HIP(hipMemcpy(ho.data(), o16, ho.size() * 2, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < ho.size(); ++i) hf[i] = f16 ? half2f_host(ho[i]) : bf2f(ho[i]);
      EC_HIP(hipMemcpy(o, hf.data(), hf.size() * 4, hipMemcpyHostToDevice));
    }
    (void)hipFree(q16); (void)hipFree(k16); (void)hipFree(v16); (void)hipFree(o16);
    return rc;
  }
  AttnP a;
  a.Q = q; a.K = k; a.V = v; a.O = o;
  a.ldq = a.ldk = a.ldv = a.ldo = (long)H * hd;
  a.sQ = a.sO = (long)Lq * H * hd; a.sK = a.sV = (long)Lk * H * hd;
  a.kmask = kmask; a.mask_start = 0; a.mask_len = Lk; a.mask_mod = 0;
  a.bias = bias;
  a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.hd = hd;
  a.split = precision == EC_BF16X3;
  return attention(a, (hipStream_t)stream);
}

}  // extern "C"
