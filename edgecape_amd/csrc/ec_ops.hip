// Non-GEMM kernels of the EdgeCape hot path for gfx950: HBM-bound normalisation / layout kernels
// and the small per-sample graph kernels of the skeleton head.  64-wide waves throughout.
#include <stdlib.h>

#include "ec_ops.h"

namespace ec {
namespace {

// ------------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, float4 loads, values kept in registers (cols <= 1024).
// Reference: DINOv2 norm1/norm2/norm (eps 1e-6), head LayerNorms (eps 1e-5, encoder_decoder.py:450-451,566-576).
// ------------------------------------------------------------------------------------------------
// OUT: 0 = fp32 output, 1 = bf16, 2 = IEEE fp16 (the fused branch inputs `add` / `add2` are in the same 16-bit format),
//      3 = bf16 split [hi | lo] in two planes `cols` elements apart (ldy >= 2 cols): the A operand of a K-concatenated bf16x3 GEMM
//      6 = the fp16x2 row [fp16 | e5m2 lo8 | e5m2 hi8] (ldy >= 2 cols 16-bit units; split4_x2)
template <int OUT, bool ADD, bool ADD_F16 = (OUT == 2)>
__global__ __launch_bounds__(256) void layernorm_kernel(LnP p) {
  constexpr bool OUT_BF16 = OUT != 0;
  constexpr bool F16 = OUT == 2;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.rows) return;
  long orow = row;
  if (p.drop_period > 0) {
    const int n = row / p.drop_period, t = row % p.drop_period;
    if (t == 0) return;
    orow = (long)n * (p.drop_period - 1) + (t - 1);
  }
  const float* x = p.x + (long)row * p.ldx;
  f32x4 v[4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < p.cols) {
      v[i] = *(const f32x4*)(x + c);
      if (ADD) {   // residual add fused in front of the norm: x <- x + branch (bf16 branch output of the previous GEMM)
        const bf16x4 a = *(const bf16x4*)((const bf16_t*)p.add + (long)row * p.ldadd + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[i][e] += h2f<ADD_F16>((bf16_t)a[e]);
        if (p.add2) {   // a second pending branch: x <- (x + add) + add2, the same two fp32 additions as two separate passes
          const bf16x4 a2 = *(const bf16x4*)((const bf16_t*)p.add2 + (long)row * p.ldadd + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[i][e] += h2f<ADD_F16>((bf16_t)a2[e]);
        }
        if (p.xsum) *(f32x4*)(p.xsum + (long)row * p.ldx + c) = v[i];   // null: the sum is only normalised, x stays as it was
      }
      s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
  }
  const float mean = wave_sum(s) / (float)p.cols;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < p.cols) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[i][e] - mean;
        q += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)p.cols + p.eps);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < p.cols) {
      const f32x4 w = *(const f32x4*)(p.w + c);
      const f32x4 b = *(const f32x4*)(p.b + c);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * w[e] + b[e];
      if constexpr (OUT == 6) {   // fp16x2 row [fp16 | e5m2 lo8 | e5m2 hi8] (ec_common.h split4_x2): the A operand of an fp16x2 GEMM
        char* y = (char*)p.y + orow * p.ldy * 2;
        u32x2_t hi;
        unsigned lo8, hi8;
        split4_x2(o, hi, lo8, hi8);
        *(u32x2_t*)(y + c * 2) = hi;
        *(unsigned*)(y + 2 * p.cols + c) = lo8;
        *(unsigned*)(y + 3 * p.cols + c) = hi8;
      } else if constexpr (OUT == 3) {
        bf16_t* y = (bf16_t*)p.y + orow * p.ldy + c;
        u32x2_t hi, lo;
        split4_bf16(o, hi, lo);
        *(u32x2_t*)y = hi;
        *(u32x2_t*)(y + p.cols) = lo;
      } else if (OUT_BF16) {
        bf16_t* y = (bf16_t*)p.y + orow * p.ldy + c;
        *(u32x2_t*)y = pack4_h<F16>(o);
      } else {
        *(f32x4*)((float*)p.y + orow * p.ldy + c) = o;
        if (p.y2) *(u32x2_t*)((bf16_t*)p.y2 + orow * p.ldy2 + c) = pack4_h<true>(o);
      }
    }
  }
}

__global__ void add_table_kernel(float* x, long ldx, const float* table, long ldt, int period, int rows, int cols4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * cols4) return;
  const int r = i / cols4, c = (i % cols4) * 4;
  f32x4 a = *(f32x4*)(x + (long)r * ldx + c);
  const f32x4 t = *(const f32x4*)(table + (long)(r % period) * ldt + c);
  a += t;
  *(f32x4*)(x + (long)r * ldx + c) = a;
}

__global__ void copy3d_kernel(float* dst, long ldd, long sd, const float* src, long lds, long ss, int rows, int cols4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * cols4) return;
  const int b = blockIdx.y;
  const int r = i / cols4, c = (i % cols4) * 4;
  *(f32x4*)(dst + b * sd + (long)r * ldd + c) = *(const f32x4*)(src + b * ss + (long)r * lds + c);
}

// Indexed row transfer between the support-side episode cache and the per-call buffers (ec_ops.h XferP): workgroup (r, y) moves row r
// of outer slice y (segments laid end to end along y) with the widest access the segment's alignment allows.
__global__ __launch_bounds__(256) void rows_xfer_kernel(XferP p) {
  const int r = blockIdx.x;
  int y = blockIdx.y, sg = 0;
  while (y >= p.seg[sg].n_outer) { y -= p.seg[sg].n_outer; ++sg; }
  const XferSeg& S = p.seg[sg];
  const long dr = p.idx_is_dst ? p.idx[r] : r, sr = p.idx_is_dst ? r : p.idx[r];
  char* dst = (char*)S.dst + (long)y * S.dst_os + dr * S.row_bytes;
  const char* src = (const char*)S.src + (long)y * S.src_os + sr * S.row_bytes;
  const long nb = S.row_bytes;
  if (S.align >= 16) {
    for (long i = (long)threadIdx.x * 16; i < nb; i += 256 * 16) *(f32x4*)(dst + i) = *(const f32x4*)(src + i);
  } else if (S.align >= 4) {
    for (long i = (long)threadIdx.x * 4; i < nb; i += 256 * 4) *(unsigned*)(dst + i) = *(const unsigned*)(src + i);
  } else {
    for (long i = threadIdx.x; i < nb; i += 256) dst[i] = src[i];
  }
}

__global__ void mean_over_kernel(float* dst, const float* src, long stride, int n, long count) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  float s = 0.f;
  for (int k = 0; k < n; ++k) s += src[k * stride + i];
  dst[i] = s / (float)n;
}

__global__ void pack_x2_kernel(const float* x, long ldx, char* y, long ldy16, int rows, int cols4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * cols4) return;
  const int r = i / cols4, c = (i % cols4) * 4, cols = cols4 * 4;
  u32x2_t hi;
  unsigned lo8, hi8;
  split4_x2(*(const f32x4*)(x + (long)r * ldx + c), hi, lo8, hi8);
  char* row = y + (long)r * ldy16 * 2;
  *(u32x2_t*)(row + c * 2) = hi;
  *(unsigned*)(row + 2 * cols + c) = lo8;
  *(unsigned*)(row + 3 * cols + c) = hi8;
}

template <bool F16>
__global__ void f32_to_bf16_kernel(const float* src, bf16_t* dst, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = f2h<F16>(src[i]);
}
// four elements per thread: 16-byte loads, 8-byte stores (n % 4 == 0, both pointers 16-byte aligned)
template <bool F16>
__global__ void f32_to_bf16_x4_kernel(const float* src, bf16_t* dst, long n4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n4) *(u32x2_t*)(dst + 4 * i) = pack4_h<F16>(*(const f32x4*)(src + 4 * i));
}

__global__ void transpose_pad_bf16_kernel(const float* src, bf16_t* dst, int L, int E, int Lp) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // over E*Lp of batch blockIdx.y
  if (i >= (long)E * Lp) return;
  const int e = i / Lp, t = i % Lp;
  const int b = blockIdx.y;
  dst[(long)b * E * Lp + i] = t < L ? f2bf(src[((long)b * L + t) * E + e]) : (bf16_t)0;
}

// im2col for Conv2d(3, C, k=14, s=14) (DINOv2 PatchEmbed): patches[(n*gh+py)*gw+px][c*196+ky*14+kx] of an H x W image,
// gh = H / 14, gw = W / 14 (floor: stride-14 VALID convolution), row length Kp >= 588 (zero padded) so the GEMM K is a multiple of 128 bytes.
template <int OUT>   // 0 fp32, 1 bf16, 2 fp16, 3 fp16 split [hi | lo | hi], 4 bf16 split [hi | lo] (bf16x3 backbone, GemmP::kwrap)
__global__ __launch_bounds__(256) void im2col14_kernel(const float* img, void* out, int H, int W, int gh, int gw, int Kp) {
  // output rows are TOKEN rows: image n owns rows n*(g*g+1) .. ; row 0 of each image (the cls token) is zero-filled so the
  // patch embedding is ONE GEMM over M = n_img * T contiguous rows (its cls rows are overwritten by set_cls_rows).
  // One workgroup per row: the 588 pixels of the patch go through LDS and leave as 16-byte stores (Kp % 8 == 0, Kp <= 1024).
  __shared__ __attribute__((aligned(16))) float px[1024];
  const int T = gh * gw + 1;
  const int n = blockIdx.x / T, tok = blockIdx.x % T;
  const long orow = blockIdx.x;
  const int pp = tok - 1;
  const int py = pp / gw, pxx = pp % gw;
  const float* src = img + (long)n * 3 * H * W;
  for (int k = threadIdx.x; k < Kp; k += blockDim.x) {
    float v = 0.f;
    if (tok > 0 && k < 588) {
      const int c = k / 196, r = k % 196, ky = r / 14, kx = r % 14;
      v = src[((long)c * H + (py * 14 + ky)) * W + pxx * 14 + kx];
    }
    px[k] = v;
  }
  __syncthreads();
  if constexpr (OUT == 0) {
    for (int i = threadIdx.x; i < Kp / 4; i += blockDim.x) *(f32x4*)((float*)out + orow * Kp + i * 4) = *(const f32x4*)(px + i * 4);
  } else if constexpr (OUT == 4 || OUT == 5) {
    const int c4n = Kp / 4;
    for (int i = threadIdx.x; i < c4n; i += blockDim.x) {
      u32x2_t h, l;
      split4_h<OUT == 5>(*(const f32x4*)(px + i * 4), h, l);
      bf16_t* o = (bf16_t*)out + orow * 2 * Kp + i * 4;
      *(u32x2_t*)o = h; *(u32x2_t*)(o + Kp) = l;
    }
  } else if constexpr (OUT == 3) {
    // split-precision patch embedding (fp16 backbone): the row is [hi | lo | hi], hi = fp16(v), lo = fp16(v - hi); against the weight
    // rows [W_hi | W_hi | W_lo] one K = 3 Kp GEMM adds hi*W_hi + lo*W_hi + hi*W_lo - the product to ~2^-22
    const int c8n = Kp / 8;
    for (int i = threadIdx.x; i < 3 * c8n; i += blockDim.x) {
      const int plane = i / c8n, c8 = i - plane * c8n;
      const f32x4 a = *(const f32x4*)(px + c8 * 8), b = *(const f32x4*)(px + c8 * 8 + 4);
      u32x2_t h0 = pack4_h<true>(a), h1 = pack4_h<true>(b);
      if (plane == 1) {
        f32x4 ra, rb;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ra[e] = a[e] - h2f<true>((bf16_t)((e & 1) ? (h0[e >> 1] >> 16) : (h0[e >> 1] & 0xffffu)));
          rb[e] = b[e] - h2f<true>((bf16_t)((e & 1) ? (h1[e >> 1] >> 16) : (h1[e >> 1] & 0xffffu)));
        }
        h0 = pack4_h<true>(ra); h1 = pack4_h<true>(rb);
      }
      typedef __attribute__((ext_vector_type(4))) unsigned u32x4_;
      *(u32x4_*)((bf16_t*)out + orow * 3 * Kp + plane * Kp + c8 * 8) = u32x4_{h0[0], h0[1], h1[0], h1[1]};
    }
  } else {
    for (int i = threadIdx.x; i < Kp / 8; i += blockDim.x) {
      const u32x2_t h0 = pack4_h<OUT == 2>(*(const f32x4*)(px + i * 8)), h1 = pack4_h<OUT == 2>(*(const f32x4*)(px + i * 8 + 4));
      typedef __attribute__((ext_vector_type(4))) unsigned u32x4_;
      *(u32x4_*)((bf16_t*)out + orow * Kp + i * 8) = u32x4_{h0[0], h0[1], h1[0], h1[1]};
    }
  }
}
__global__ void set_cls_kernel(float* x, long ldx, const float* cls, const float* pos0, int T, int C) {
  const int n = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) x[(long)n * T * ldx + c] = cls[c] + pos0[c];
}

// [n, C, HW] <-> [n, HW, C] via a 32x32 LDS tile
__global__ void transpose_kernel(const float* src, float* dst, int R, int Cc) {
  // src [n][R][Cc] -> dst [n][Cc][R]
  __shared__ float t[32][33];
  const int n = blockIdx.z;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty 0..7
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    t[i][tx] = (r < R && c < Cc) ? src[((long)n * R + r) * Cc + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (c < Cc && r < R) dst[((long)n * Cc + c) * R + r] = t[tx][i];
  }
}

// ------------------------------------------------------------------------------------------------
// Support-keypoint pooling weights (head.py:175-184).  The reference upsamples the g x g feature
// map to hm x hm bilinearly (align_corners=False) and takes a heatmap-weighted mean.  Bilinear
// interpolation is linear, so  sum_p t[p] * interp(feat)[p] = sum_cell W[cell] * feat[cell]  with
// W = (R_y^T t R_x) / (sum t + 1e-8): one block per (sample, keypoint) builds W separably in LDS;
// a batched GEMM then contracts W with the features.  Works for arbitrary (dense) heatmaps.
// ------------------------------------------------------------------------------------------------
__device__ inline void bilinear_src(int dst, float scale, int in_size, int& i0, int& i1, float& l1) {
  float s = scale * ((float)dst + 0.5f) - 0.5f;   // ATen area_pixel_compute_source_index, align_corners=False
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  i0 = i0 < in_size - 1 ? i0 : in_size - 1;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = s - (float)i0;
}

// Fused support-keypoint pooling (head.py:175-186): the heatmap's bilinear tap weights over the g x g token grid (the adjoint of
// the reference's resize, normalised by the heatmap sum), then the pooled feature
//   pooled[b,k,:] (+)= sum_cell w[cell] * F[b,cell,:]
// directly, visiting only the non-zero weights in cell order.  The targets of this pipeline are Gaussian blobs (sigma 2 on the
// 64x64 heatmap = ~4x4 of the 18x18 token grid), so ~16-30 of the 324 cells are non-zero and the dense [K,HW]@[HW,C]
// contraction (and its launch) disappears; a dense heatmap is still exact, just slower (every cell is visited).
// Both resize passes run over (cell-row, column) pairs with per-coordinate tap tables, summing in ascending source order.
// MODE 0: the fused kernel.  MODE 1 / 2 (round 3, ec_forward_pipelined): the same kernel cut in two at the compacted tap list - MODE 1
// reads the caller's heatmap and mask and writes the list (tap_n[bk] = number of non-zero cells, -1 for a padded slot; tap_i / tap_w
// [bk][g*g]: cell index and weight, in the fused kernel's order), MODE 2 gathers the feature rows with it.  Identical arithmetic in
// identical order: the two halves together are bit-equal to the fused kernel.  The first half needs no backbone output, so a pipelined
// call runs it BESIDE its backbone and the caller's inputs are consumed long before the caller's stream reaches the end of the call.
template <int MODE>
__global__ __launch_bounds__(256) void pool_gather_kernel(const float* target, const float* mask_s, float inv_shots, const float* F,
                                                          float* pooled, float beta, int K, int hm, int gh, int gw, int C, int* tap_n,
                                                          int* tap_i, float* tap_w) {
  // token grid gh rows x gw columns (round 4: the two axes have their own tap tables; a square grid gives the old arithmetic)
  extern __shared__ float sm[];
  const int gg = gh * gw;
  float* t = sm;                       // hm*hm heatmap
  float* tmp = t + hm * hm;            // gh*hm : tmp[cy][x]
  float* wts = tmp + gh * hm;          // gh*gw tap weights
  float* tl = wts + gg;                // hm   : lambda of source ROW y
  float* tlx = tl + hm;                // hm   : lambda of source COLUMN x
  int* ti0 = (int*)(tlx + hm);         // hm   : first target cell row of source row y
  int* ti1 = ti0 + hm;                 // hm   : second
  int* tx0 = ti1 + hm;                 // hm   : first target cell column of source column x
  int* tx1 = tx0 + hm;                 // hm   : second
  int* nzi = tx1 + hm;                 // gh*gw: compacted non-zero cells
  float* red = (float*)(nzi + gg);     // 4 + 1 (count)
  const int bk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = bk / K;
  float* out = pooled + (long)bk * C;
  if constexpr (MODE == 2) {
    const int n = tap_n[bk];
    if (n < 0) {
      if (beta == 0.f)
        for (int c = tid; c < C; c += 256) out[c] = 0.f;
      return;
    }
    for (int i = tid; i < n; i += 256) {
      const int cell = tap_i[(long)bk * gg + i];
      nzi[i] = cell;
      wts[cell] = tap_w[(long)bk * gg + i];
    }
    if (tid == 0) ((int*)red)[4] = n;
    __syncthreads();
  } else {
  const float msk = mask_s[bk];
  if (msk == 0.f) {   // padded keypoint slot: pooled feature is multiplied by mask_s = 0 (head.py:187)
    if constexpr (MODE == 1) {
      if (tid == 0) tap_n[bk] = -1;
    } else {
      if (beta == 0.f)
        for (int c = tid; c < C; c += 256) out[c] = 0.f;
    }
    return;
  }
  const float* src = target + (long)bk * hm * hm;
  const float scale = (float)gh / (float)hm, scale_x = (float)gw / (float)hm;
  float s = 0.f;
  for (int i = tid; i < hm * hm; i += 256) {
    const float v = src[i];
    t[i] = v;
    s += v;
  }
  if (tid < hm) {
    int i0, i1; float l;
    bilinear_src(tid, scale, gh, i0, i1, l);
    ti0[tid] = i0; ti1[tid] = i1; tl[tid] = l;
    bilinear_src(tid, scale_x, gw, i0, i1, l);
    tx0[tid] = i0; tx1[tid] = i1; tlx[tid] = l;
  }
  s = wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  const float total = red[0] + red[1] + red[2] + red[3];
  const float inv_scale = (float)hm / (float)gh, inv_scale_x = (float)hm / (float)gw;
  // tmp[cy][x] = sum_y wy(cy, y) t[y][x], y ascending; only sources within one cell of cy can contribute
  for (int idx = tid; idx < gh * hm; idx += 256) {
    const int cy = idx / hm, x = idx - cy * hm;
    const int ylo = max(0, (int)((float)(cy - 1) * inv_scale) - 1), yhi = min(hm - 1, (int)((float)(cy + 2) * inv_scale) + 1);
    float acc = 0.f;
    for (int y = ylo; y <= yhi; ++y) {
      const float v = t[y * hm + x], l = tl[y];
      if (ti0[y] == cy) acc += (1.f - l) * v;
      if (ti1[y] == cy) acc += l * v;
    }
    tmp[idx] = acc;
  }
  __syncthreads();
  const float norm = msk * inv_shots / (total + 1e-8f);
  for (int cell = tid; cell < gg; cell += 256) {
    const int cy = cell / gw, cx = cell - cy * gw;
    const int xlo = max(0, (int)((float)(cx - 1) * inv_scale_x) - 1), xhi = min(hm - 1, (int)((float)(cx + 2) * inv_scale_x) + 1);
    float acc = 0.f;
    for (int x = xlo; x <= xhi; ++x) {
      const float l = tlx[x];
      float w = 0.f;
      if (tx0[x] == cx) w += 1.f - l;
      if (tx1[x] == cx) w += l;
      if (w != 0.f) acc += w * tmp[cy * hm + x];
    }
    wts[cell] = acc * norm;
  }
  __syncthreads();
  if (wave == 0) {   // ordered compaction of the non-zero cells
    int n = 0;
    for (int c0 = 0; c0 < gg; c0 += 64) {
      const int cell = c0 + lane;
      const bool nz = cell < gg && wts[cell] != 0.f;
      const unsigned long long m = __ballot(nz);
      if (nz) nzi[n + __popcll(m & ((1ull << lane) - 1ull))] = cell;
      n += __popcll(m);
    }
    if (lane == 0) ((int*)red)[4] = n;
  }
  __syncthreads();
  if constexpr (MODE == 1) {
    const int n = ((const int*)red)[4];
    for (int i = tid; i < n; i += 256) {
      tap_i[(long)bk * gg + i] = nzi[i];
      tap_w[(long)bk * gg + i] = wts[nzi[i]];
    }
    if (tid == 0) tap_n[bk] = n;
    return;
  }
  }   // MODE != 2
  const int nnz = ((const int*)red)[4];
  const float* Fb = F + (long)b * gg * C;
  for (int c = tid; c < C; c += 256) {
    float acc = 0.f;
    int i = 0;
    for (; i + 4 <= nnz; i += 4) {   // four rows in flight; summation order stays i ascending
      const int c0 = nzi[i], c1 = nzi[i + 1], c2 = nzi[i + 2], c3 = nzi[i + 3];
      const float f0 = Fb[(long)c0 * C + c], f1 = Fb[(long)c1 * C + c], f2 = Fb[(long)c2 * C + c], f3 = Fb[(long)c3 * C + c];
      acc = fmaf(wts[c0], f0, acc);
      acc = fmaf(wts[c1], f1, acc);
      acc = fmaf(wts[c2], f2, acc);
      acc = fmaf(wts[c3], f3, acc);
    }
    for (; i < nnz; ++i) acc = fmaf(wts[nzi[i]], Fb[(long)nzi[i] * C + c], acc);
    out[c] = beta != 0.f ? beta * out[c] + acc : acc;
  }
}

// ------------------------------------------------------------------------------------------------
// adj_mx_from_edges + normalize_adj + (gt_adj > 0) + soft_normalize_adj  (skeleton.py:171-205,72,91)
// one block per sample
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adj_build_kernel(const int32_t* edges, const int32_t* offsets, const float* mask_s,
                                                        float* valid, uint8_t* kmask, uint8_t* kmask_fixed, float* binary,
                                                        float* adj_r1, int K) {
  extern __shared__ unsigned char flag[];   // K*K
  __shared__ int nvalid;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) nvalid = 0;
  for (int i = tid; i < K * K; i += 256) flag[i] = 0;
  __syncthreads();
  const int e0 = offsets[b], e1 = offsets[b + 1];
  for (int e = e0 + tid; e < e1; e += 256) {
    const int a = edges[2 * e], c = edges[2 * e + 1];
    if (a >= 0 && a < K && c >= 0 && c < K) {
      flag[a * K + c] = 1;
      flag[c * K + a] = 1;
    }
  }
  int cnt = 0;
  const bool first = blockIdx.y == 0;   // the row blocks of a sample share the masks: block 0 writes them
  for (int k = tid; k < K; k += 256) {
    const bool v = mask_s[b * K + k] != 0.f;   // kp_mask = ~mask_s.bool() (head.py:189)
    if (first) {
      valid[b * K + k] = v ? 1.f : 0.f;
      kmask[b * K + k] = v ? 0 : 1;
    }
    cnt += v ? 1 : 0;
  }
  if (cnt) atomicAdd(&nvalid, cnt);
  __syncthreads();
  for (int k = tid; k < K; k += 256) {
    const bool v = mask_s[b * K + k] != 0.f;
    // tgt_key_padding_mask_remove_all_true (skeleton.py:98-99): un-mask key 0 of all-padded samples
    if (first) kmask_fixed[b * K + k] = (v || (nvalid == 0 && k == 0)) ? 0 : 1;
  }
  const int wave = tid >> 6, lane = tid & 63;
  for (int i = blockIdx.y * 4 + wave; i < K; i += 4 * gridDim.y) {
    const bool vi = mask_s[b * K + i] != 0.f;
    float rs = 0.f;
    for (int j = lane; j < K; j += 64) {
      const bool vj = mask_s[b * K + j] != 0.f;
      rs += (vi && vj && flag[i * K + j]) ? 1.f : 0.f;
    }
    rs = wave_sum(rs);
    for (int j = lane; j < K; j += 64) {
      const bool vj = mask_s[b * K + j] != 0.f;
      const float u = (vi && vj && flag[i * K + j]) ? 1.f : 0.f;
      binary[((long)b * K + i) * K + j] = u;
      adj_r1[((long)b * K + i) * K + j] = u / (rs + 1e-8f);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// rowplan (ec_ops.h): ONE workgroup of 16 waves.  Pass 1: wave w counts the samples w, w + 16, ... (ballots over the K <= 128 mask
// bits); a serial prefix over the samples by thread 0 (ns <= a few hundred: sub-microsecond against the ~5 us the launch costs);
// pass 2: every wave writes its samples' entries.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void rowplan_kernel(const float* mask_s, int bs, int ns, int K, int* plan, int* rowmap, int* fan_base,
                                                       unsigned long long* fan_bits) {
  extern __shared__ int pl_lds[];        // [ns] active offsets, [ns] copy offsets
  int* a_off = pl_lds;
  int* c_off = pl_lds + ns;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  auto masks = [&](int i, unsigned long long& v0, unsigned long long& v1) {
    const float* mrow = mask_s + (long)(i % bs) * K;
    v0 = __ballot(lane < K && mrow[lane] != 0.f);
    v1 = __ballot(lane + 64 < K && mrow[min(lane + 64, K - 1)] != 0.f);
    if (v0 == 0ull && v1 == 0ull) v0 = 1ull;   // no valid token: token 0 is a row of its own (see ec_ops.h)
  };
  for (int i = wave; i < ns; i += 16) {
    unsigned long long v0, v1;
    masks(i, v0, v1);
    const int nv = __popcll(v0) + __popcll(v1);
    if (lane == 0) {
      a_off[i] = nv + (nv < K ? 1 : 0);          // valid tokens + one representative of the masked ones
      c_off[i] = nv < K ? K - nv - 1 : 0;        // the other masked tokens are copies
    }
  }
  __syncthreads();
  if (tid == 0) {
    int a = 0, c = 0;
    for (int i = 0; i < ns; ++i) {
      const int na = a_off[i], nc = c_off[i];
      a_off[i] = a; c_off[i] = c;
      a += na; c += nc;
    }
    plan[0] = a; plan[1] = c;
  }
  __syncthreads();
  for (int i = wave; i < ns; i += 16) {
    unsigned long long v0, v1;
    masks(i, v0, v1);
    const int nv = __popcll(v0) + __popcll(v1);
    const unsigned long long below = (1ull << lane) - 1ull;
    // representative: the first masked token
    const unsigned long long m0 = ~v0 & (K >= 64 ? ~0ull : ((1ull << K) - 1ull));
    const unsigned long long m1 = K > 64 ? (~v1 & (K >= 128 ? ~0ull : ((1ull << (K - 64)) - 1ull))) : 0ull;
    const int rep = m0 ? __ffsll((long long)m0) - 1 : (m1 ? 64 + __ffsll((long long)m1) - 1 : -1);
    const long base = (long)i * K;
    const int ao = a_off[i], co = c_off[i];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int k = lane + 64 * half;
      if (k >= K) continue;
      const unsigned long long v = half ? v1 : v0, mk = half ? m1 : m0;
      const int vrank = (half ? __popcll(v0) : 0) + __popcll(v & below);
      const int mrank = (half ? __popcll(m0) : 0) + __popcll(mk & below);
      if ((v >> lane) & 1ull) {
        rowmap[ao + vrank] = (int)(base + k); fan_base[ao + vrank] = 0; fan_bits[2 * (ao + vrank)] = 0ull; fan_bits[2 * (ao + vrank) + 1] = 0ull;
      } else if (k == rep) {   // the representative: its rows are also the sample's other masked tokens' (bit k of the two words)
        rowmap[ao + nv] = (int)(base + k); fan_base[ao + nv] = (int)base;
        fan_bits[2 * (ao + nv)] = rep < 64 ? m0 & ~(1ull << rep) : m0;
        fan_bits[2 * (ao + nv) + 1] = rep >= 64 ? m1 & ~(1ull << (rep - 64)) : m1;
      }
    }
  }
}

// SkeletonPredictor(learn_skeleton=False): adj = stack(diag(valid), normalised ground-truth adjacency) (skeleton.py:70-74, 187-194)
__global__ __launch_bounds__(256) void adj_gt_kernel(const float* adj_r1, const float* valid, float* adj_out, float* adj1, int K) {
  const int b = blockIdx.y, idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= K * K) return;
  const int i = idx / K, j = idx - i * K;
  const long KK = (long)K * K;
  const float a = adj_r1[b * KK + idx];
  adj1[b * KK + idx] = a;
  adj_out[b * 2 * KK + KK + idx] = a;
  adj_out[b * 2 * KK + idx] = i == j ? valid[(long)b * K + i] : 0.f;
}

__global__ __launch_bounds__(256) void rownorm_kernel(const float* x, float* y, int rows, int cols) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float s = 0.f;
  for (int c = lane; c < cols; c += 64) {
    const float v = x[(long)row * cols + c];
    s += v * v;
  }
  const float inv = 1.f / (sqrtf(wave_sum(s)) + 1e-8f);   // skeleton.py:137
  for (int c = lane; c < cols; c += 64) y[(long)row * cols + c] = x[(long)row * cols + c] * inv;
}

// predict_skeleton tail + markov normalisation (skeleton.py:139-150,158): one wave per adjacency row, grid (bs, ceil(K/4))
// gram = 1: P holds the UN-normalised Gram matrix X X^T of the refined keypoint tokens; the cosine similarity of skeleton.py:137-139
// (x / (|x| + 1e-8) on both sides) is formed here from its diagonal, |x_i| = sqrt(G_ii) - the separate row-normalisation launch on
// the support lane's critical path is gone.
__global__ __launch_bounds__(256) void adj_combine_kernel(const float* P, const float* binary, const float* valid,
                                                          const float* zc_w, const float* zc_b, float* adj_out, float* adj1,
                                                          float* attn_adj, int bs, int K, int gram) {
  const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float w = zc_w[0], c0 = zc_b[0];
  const float* Pb = P + (long)b * K * K;
  const float* Bb = binary + (long)b * K * K;
  float* a0 = adj_out + (long)b * 2 * K * K;
  float* a1 = a0 + (long)K * K;
  float* i0 = attn_adj + (long)b * K * K;                       // hop 0 = I
  float* m1 = attn_adj + ((long)bs + b) * K * K;                // hop 1 = A
  float* A1 = adj1 + (long)b * K * K;
  for (int i = blockIdx.y * 4 + wave; i < K; i += 4 * gridDim.y) {
    const float vi = valid[b * K + i];
    const float ni = gram ? 1.f / (sqrtf(Pb[i * K + i]) + 1e-8f) : 1.f;
    float u[4], rs = 0.f;   // K <= 256: four columns per lane
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int j = lane + t * 64;
      u[t] = 0.f;
      if (j < K) {
        const float nj = gram ? 1.f / (sqrtf(Pb[j * K + j]) + 1e-8f) : 1.f;
        const float sym = (Pb[i * K + j] * ni * nj + Pb[j * K + i] * nj * ni) / 2.f;
        float v = Bb[i * K + j] + (sym * w + c0);
        v = fmaxf(v, 0.f);
        u[t] = v * (vi * valid[b * K + j]);
        rs += u[t];
      }
    }
    rs = wave_sum(rs);
    float rs2 = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      u[t] = u[t] / (rs + 1e-8f);
      rs2 += u[t];
    }
    rs2 = wave_sum(rs2);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int j = lane + t * 64;
      if (j < K) {
        a0[i * K + j] = (i == j) ? vi : 0.f;
        a1[i * K + j] = u[t];
        A1[i * K + j] = u[t];
        i0[i * K + j] = (i == j) ? 1.f : 0.f;
        m1[i * K + j] = u[t] / (rs2 + 1e-8f);
      }
    }
  }
}

// compile-time sizes (the shipped configuration: max_hops + 1 = 5 -> 12 -> 8 heads): everything stays in registers
struct BiasMlpLayers {   // weights of up to 4 decoder layers' Markov-bias MLPs; layer = blockIdx.y, outputs `stride` floats apart
  const float* w1[4]; const float* b1[4]; const float* w2[4]; const float* b2[4];
  long stride;
};
template <int H1, int HID, int NH>
__global__ __launch_bounds__(256) void bias_mlp_fixed_kernel(const float* attn_adj, BiasMlpLayers L, float* out, int bs, int K) {
  __shared__ float sw1[HID * H1], sb1[HID], sw2[NH * HID], sb2[NH];
  const float *w1 = L.w1[blockIdx.y], *b1 = L.b1[blockIdx.y], *w2 = L.w2[blockIdx.y], *b2 = L.b2[blockIdx.y];
  out += (long)blockIdx.y * L.stride;
  for (int i = threadIdx.x; i < HID * H1; i += 256) sw1[i] = w1[i];
  for (int i = threadIdx.x; i < HID; i += 256) sb1[i] = b1[i];
  for (int i = threadIdx.x; i < NH * HID; i += 256) sw2[i] = w2[i];
  for (int i = threadIdx.x; i < NH; i += 256) sb2[i] = b2[i];
  __syncthreads();
  const long KK = (long)K * K;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= bs * KK) return;
  const int b = idx / KK;
  const long ij = idx % KK;
  float a[H1], hdn[HID];
#pragma unroll
  for (int d = 0; d < H1; ++d) a[d] = attn_adj[((long)d * bs + b) * KK + ij];
#pragma unroll
  for (int o = 0; o < HID; ++o) {
    float s = sb1[o];
#pragma unroll
    for (int d = 0; d < H1; ++d) s += sw1[o * H1 + d] * a[d];
    hdn[o] = fmaxf(s, 0.f);
  }
#pragma unroll
  for (int hh = 0; hh < NH; ++hh) {
    float s = sb2[hh];
#pragma unroll
    for (int o = 0; o < HID; ++o) s += sw2[hh * HID + o] * hdn[o];
    out[((long)b * NH + hh) * KK + ij] = s;
  }
}

__global__ void bias_mlp_kernel(const float* attn_adj, const float* w1, const float* b1, const float* w2, const float* b2,
                                float* out, int hops1, int hidden, int nhead, int bs, int K) {
  extern __shared__ float wsm[];
  float* sw1 = wsm;                       // hidden*hops1
  float* sb1 = sw1 + hidden * hops1;
  float* sw2 = sb1 + hidden;              // nhead*hidden
  float* sb2 = sw2 + nhead * hidden;
  for (int i = threadIdx.x; i < hidden * hops1; i += blockDim.x) sw1[i] = w1[i];
  for (int i = threadIdx.x; i < hidden; i += blockDim.x) sb1[i] = b1[i];
  for (int i = threadIdx.x; i < nhead * hidden; i += blockDim.x) sw2[i] = w2[i];
  for (int i = threadIdx.x; i < nhead; i += blockDim.x) sb2[i] = b2[i];
  __syncthreads();
  const long KK = (long)K * K;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= bs * KK) return;
  const int b = idx / KK;
  const long ij = idx % KK;
  float a[8], hdn[16];
  for (int d = 0; d < hops1; ++d) a[d] = attn_adj[((long)d * bs + b) * KK + ij];
  for (int o = 0; o < hidden; ++o) {
    float s = sb1[o];
    for (int d = 0; d < hops1; ++d) s += sw1[o * hops1 + d] * a[d];
    hdn[o] = fmaxf(s, 0.f);
  }
  for (int hh = 0; hh < nhead; ++hh) {
    float s = sb2[hh];
    for (int o = 0; o < hidden; ++o) s += sw2[hh * hidden + o] * hdn[o];
    out[((long)b * nhead + hh) * KK + ij] = s;
  }
}

// ProposalGenerator tail (encoder_decoder.py:76-112): one wave per (sample, keypoint) row of the similarity map
__global__ __launch_bounds__(256) void proposals_kernel(const float* sim, float* prop_loss, float* prop, int rows, int gh, int gw) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int HW = gh * gw;
  const float* s = sim + (long)row * HW;
  constexpr int MAXC = 16;  // g <= 32
  float v[MAXC];
  float mx = -INFINITY;
  int am = 0x7fffffff;
#pragma unroll
  for (int t = 0; t < MAXC; ++t) {
    const int p = lane + t * 64;
    v[t] = p < HW ? s[p] : -INFINITY;
    if (v[t] > mx) { mx = v[t]; am = p; }
  }
  // wave arg-max, first index on ties (torch.argmax)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(mx, o, 64);
    const int oi = __shfl_xor(am, o, 64);
    if (ov > mx || (ov == mx && oi < am)) { mx = ov; am = oi; }
  }
  float se = 0.f;
#pragma unroll
  for (int t = 0; t < MAXC; ++t) {
    const int p = lane + t * 64;
    v[t] = p < HW ? expf(v[t] - mx) : 0.f;
    se += v[t];
  }
  se = wave_sum(se);
  // the reference reshapes the one-hot - a flat index over (h, w) - to (w, h) before the 3x3 max-pool (:93-97): the local window is
  // taken in THAT layout, (p / h, p % h); for square maps it is the natural (row, col), for h != w it is not a spatial
  // neighbourhood, and the reference's arithmetic is what is reproduced
  const int ar = am / gh, ac = am % gh;
  float sx = 0.f, sy = 0.f, lx = 0.f, ly = 0.f, ls = 0.f;
#pragma unroll
  for (int t = 0; t < MAXC; ++t) {
    const int p = lane + t * 64;
    if (p < HW) {
      const float pr = v[t] / se;
      const int r = p / gw, c = p % gw;
      const float gx = (float)c + 0.5f, gy = (float)r + 0.5f;
      sx += pr * gx;
      sy += pr * gy;
      const int dr = p / gh - ar, dc = p % gh - ac;
      if (dr >= -1 && dr <= 1 && dc >= -1 && dc <= 1) {
        ls += pr;
        lx += pr * gx;
        ly += pr * gy;
      }
    }
  }
  sx = wave_sum(sx); sy = wave_sum(sy); lx = wave_sum(lx); ly = wave_sum(ly); ls = wave_sum(ls);
  if (lane == 0) {
    prop_loss[row * 2 + 0] = sx / (float)gw;
    prop_loss[row * 2 + 1] = sy / (float)gh;
    const float d = ls + 1e-10f;
    prop[row * 2 + 0] = (lx / d) / (float)gw;
    prop[row * 2 + 1] = (ly / d) / (float)gh;
  }
}

// SinePositionalEncoding.forward_coordinates (positional_encoding.py:96-122): out = cat(pos_y, pos_x)
__global__ void sincos_kernel(const float* coords, const float* dim_t, float* out, long ldo, int rows, int nf) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)rows * 2 * nf) return;
  const int row = idx / (2 * nf), c = idx % (2 * nf);
  const int isx = c >= nf;            // first nf features come from y
  const int i = isx ? c - nf : c;
  const float e = coords[row * 2 + (isx ? 0 : 1)] * 6.283185307179586f;
  const float a = e / dim_t[i];
  out[(long)row * ldo + c] = (i & 1) ? cosf(a) : sinf(a);
}

__device__ inline float inv_sigmoid(float x) {   // head.py:27-31, eps 1e-3
  x = fminf(fmaxf(x, 0.f), 1.f);
  const float x1 = fmaxf(x, 1e-3f), x2 = fmaxf(1.f - x, 1e-3f);
  return logf(x1 / x2);
}

__global__ __launch_bounds__(256) void kpt_out_kernel(const float* h, long ldh, const float* W, const float* b,
                                                      const float* prev, float* out, int rows, int d) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float s0 = 0.f, s1 = 0.f;
  for (int c = lane; c < d; c += 64) {
    const float v = h[(long)row * ldh + c];
    s0 += v * W[c];
    s1 += v * W[d + c];
  }
  s0 = wave_sum(s0);
  s1 = wave_sum(s1);
  if (lane < 2) {
    const float dl = (lane == 0 ? s0 : s1) + b[lane];
    const float z = dl + inv_sigmoid(prev[row * 2 + lane]);
    out[row * 2 + lane] = 1.f / (1.f + expf(-z));
  }
}

__global__ void set_identity_kernel(float* dst, int K) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < K * K) dst[(long)b * K * K + i] = (i / K == i % K) ? 1.f : 0.f;
}

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace

int layernorm(const LnP& p, hipStream_t st) {
  EC_REQUIRE(p.cols % 4 == 0 && p.cols <= 1024, -1, "layernorm: cols must be a multiple of 4 and <= 1024");
  EC_REQUIRE(p.ldx % 4 == 0 && p.ldy % 4 == 0, -1, "layernorm: strides must be multiples of 4");
  EC_REQUIRE(!p.add || p.ldadd % 4 == 0, -1, "layernorm: fused residual add needs a 4-aligned stride");
  EC_REQUIRE(!p.add2 || p.add, -1, "layernorm: add2 without add");
  const dim3 grid(cdiv(p.rows, 4));
  // y_bf16: 0 fp32, 1 bf16, 2 fp16 output; a fused branch add is 16-bit in add_fmt's format (defaults to the output's, bf16 if fp32)
  const int afmt = p.add_fmt ? p.add_fmt : (p.y_bf16 ? p.y_bf16 : 1);
  EC_REQUIRE(!p.add || p.y_bf16 == 0 || afmt == p.y_bf16, -1, "layernorm: fused add and output must share the 16-bit format");
  if (p.y_bf16 == 3 || p.y_bf16 == 6) {
    EC_REQUIRE(!p.add && p.ldy >= 2 * (long)p.cols, -1, "layernorm: split output takes no fused add and needs ldy >= 2 cols");
    if (p.y_bf16 == 6) hipLaunchKernelGGL((layernorm_kernel<6, false, false>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((layernorm_kernel<3, false>), grid, dim3(256), 0, st, p);
  } else if (p.add) {
    if (p.y_bf16 == 2) hipLaunchKernelGGL((layernorm_kernel<2, true>), grid, dim3(256), 0, st, p);
    else if (p.y_bf16 == 1) hipLaunchKernelGGL((layernorm_kernel<1, true>), grid, dim3(256), 0, st, p);
    else if (afmt == 2) hipLaunchKernelGGL((layernorm_kernel<0, true, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((layernorm_kernel<0, true, false>), grid, dim3(256), 0, st, p);
  } else {
    if (p.y_bf16 == 2) hipLaunchKernelGGL((layernorm_kernel<2, false>), grid, dim3(256), 0, st, p);
    else if (p.y_bf16 == 1) hipLaunchKernelGGL((layernorm_kernel<1, false>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((layernorm_kernel<0, false>), grid, dim3(256), 0, st, p);
  }
  EC_LAUNCH_CHECK();
  return 0;
}

int add_table(float* x, long ldx, const float* table, long ldt, int period, int rows, int cols, hipStream_t st) {
  EC_REQUIRE(cols % 4 == 0 && ldx % 4 == 0 && ldt % 4 == 0, -1, "add_table: cols/strides must be multiples of 4");
  hipLaunchKernelGGL(add_table_kernel, dim3(cdiv((long)rows * cols / 4, 256)), dim3(256), 0, st, x, ldx, table, ldt, period,
                     rows, cols / 4);
  EC_LAUNCH_CHECK();
  return 0;
}

int copy3d(float* dst, long ldd, long sd, const float* src, long lds, long ss, int batch, int rows, int cols, hipStream_t st) {
  EC_REQUIRE(cols % 4 == 0 && ldd % 4 == 0 && lds % 4 == 0 && sd % 4 == 0 && ss % 4 == 0, -1, "copy3d: alignment");
  hipLaunchKernelGGL(copy3d_kernel, dim3(cdiv((long)rows * cols / 4, 256), batch), dim3(256), 0, st, dst, ldd, sd, src, lds, ss,
                     rows, cols / 4);
  EC_LAUNCH_CHECK();
  return 0;
}

int copy2d(float* dst, long ldd, const float* src, long lds, int rows, int cols, hipStream_t st) {
  return copy3d(dst, ldd, 0, src, lds, 0, 1, rows, cols, st);
}

// ---------------------------------------------------------------------------------------------------------------
// On-device input pipeline: TopDownAffineFewShot + ToTensor + NormalizeTensor
// (EdgeCape/datasets/pipelines/top_down_transform.py:35-58, configs/test/1shot_split1.py:117-125) and
// TopDownGenerateTargetFewShot._msra_generate_target (top_down_transform.py:165-194).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void preprocess_affine_kernel(PreprocBatch pb, float* out, int H) {
  // one thread per destination pixel; dst -> src through the inverse 2x3 matrix (what warpAffine does without
  // WARP_INVERSE_MAP), bilinear, constant-0 border, then x/255, (x - mean) / std, HWC -> CHW.
  const int img = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= H * H) return;
  const int y = i / H, x = i - y * H;
  const float* M = pb.inv[img];
  const float sx = M[0] * (float)x + M[1] * (float)y + M[2];
  const float sy = M[3] * (float)x + M[4] * (float)y + M[5];
  const int x0 = (int)floorf(sx), y0 = (int)floorf(sy);
  const float fx = sx - (float)x0, fy = sy - (float)y0;
  const int Hs = pb.hs[img], Ws = pb.ws[img];
  const unsigned char* src = pb.src[img];
  const long pitch = pb.pitch[img];
  float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const int xx = x0 + dx, yy = y0 + dy;
      const float w = (dx ? fx : 1.f - fx) * (dy ? fy : 1.f - fy);
      if (xx >= 0 && xx < Ws && yy >= 0 && yy < Hs) {
        const unsigned char* px = src + (long)yy * pitch + (long)xx * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] = fmaf(w, (float)px[c], acc[c]);
      }
    }
  float* o = out + (long)img * 3 * H * H + (long)y * H + x;
#pragma unroll
  for (int c = 0; c < 3; ++c) o[(long)c * H * H] = (acc[c] * (1.f / 255.f) - pb.mean[c]) / pb.stdv[c];
}

// The same stage with cv2.warpAffine's OWN arithmetic (uint8 source, INTER_LINEAR, BORDER_CONSTANT 0; OpenCV imgwarp.cpp
// WarpAffineInvoker + remapBilinear, the classic fixed-point path): source coordinates in 1/1024 px from float64 products rounded
// half-to-even (cvRound), + 16, >> 5 -> 1/32 px; the four taps weighted by (32 - fy | fy) * (32 - fx | fx) * 32 (the int16 table
// BilinearTab_i, sum 32768), (sum + 2^14) >> 15 saturated to uint8; then ToTensor / NormalizeTensor in their float32 arithmetic
// ((u8 / 255 - mean) / std with IEEE divisions).  Bit-exact against oracle/pipeline_oracle.py cv2_warp_affine_linear_u8.
__global__ __launch_bounds__(256) void preprocess_affine_cv2_kernel(PreprocBatchCv2 pb, float* out, int H) {
  const int img = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= H * H) return;
  const int y = i / H, x = i - y * H;
  const double* M = pb.minv[img];
  // every product and sum rounded on its own (no FMA contraction), as the host code of OpenCV evaluates them
  const double xd = (double)x, yd = (double)y;
  const long ad = (long)__builtin_rint(__dmul_rn(__dmul_rn(M[0], xd), 1024.0));
  const long bd = (long)__builtin_rint(__dmul_rn(__dmul_rn(M[3], xd), 1024.0));
  const long X0 = (long)__builtin_rint(__dmul_rn(__dadd_rn(__dmul_rn(M[1], yd), M[2]), 1024.0)) + 16;
  const long Y0 = (long)__builtin_rint(__dmul_rn(__dadd_rn(__dmul_rn(M[4], yd), M[5]), 1024.0)) + 16;
  const long X = (X0 + ad) >> 5, Y = (Y0 + bd) >> 5;
  const long sxl = X >> 5, syl = Y >> 5;
  const int sx = (int)(sxl < -32768 ? -32768 : sxl > 32767 ? 32767 : sxl);       // saturate_cast<short>
  const int sy = (int)(syl < -32768 ? -32768 : syl > 32767 ? 32767 : syl);
  const int fx = (int)(X & 31), fy = (int)(Y & 31);
  const int Hs = pb.hs[img], Ws = pb.ws[img];
  const unsigned char* src = pb.src[img];
  const long pitch = pb.pitch[img];
  int acc[3] = {0, 0, 0};
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const int xx = sx + dx, yy = sy + dy;
      const int w = (dy ? fy : 32 - fy) * (dx ? fx : 32 - fx) * 32;
      if (xx >= 0 && xx < Ws && yy >= 0 && yy < Hs) {
        const unsigned char* px = src + (long)yy * pitch + (long)xx * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] += w * (int)px[c];
      }
    }
  float* o = out + (long)img * 3 * H * H + (long)y * H + x;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    int v = (acc[c] + (1 << 14)) >> 15;
    v = v < 0 ? 0 : v > 255 ? 255 : v;
    o[(long)c * H * H] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)v, 255.f), pb.mean[c]), pb.stdv[c]);
  }
}

__global__ __launch_bounds__(256) void msra_target_kernel(const float* joints, const float* visible, float* target, float* weight,
                                                          MsraP mp) {
  // one workgroup per (sample, keypoint): zero the hm x hm map, then paste the in-bounds part of the (2*3*sigma+1)^2 gaussian
  const int jk = blockIdx.x;
  const int hm = mp.hm, tmp = mp.tmp, size = 2 * mp.tmp + 1;
  // reference arithmetic: float32 joint / float64 stride + 0.5, int() truncation (top_down_transform.py:170-172)
  const int mu_x = (int)((double)joints[2 * (long)jk] / mp.stride + 0.5);
  const int mu_y = (int)((double)joints[2 * (long)jk + 1] / mp.stride + 0.5);
  const int ulx = mu_x - tmp, uly = mu_y - tmp, brx = mu_x + tmp + 1, bry = mu_y + tmp + 1;
  float w = visible[jk];
  if (ulx >= hm || uly >= hm || brx < 0 || bry < 0) w = 0.f;
  float* t = target + (long)jk * hm * hm;
  for (int i = threadIdx.x; i < hm * hm; i += blockDim.x) {
    const int y = i / hm, x = i - y * hm;
    float v = 0.f;
    if (w > 0.5f && x >= ulx && x < brx && y >= uly && y < bry) v = mp.g[(y - uly) * size + (x - ulx)];
    t[i] = v;
  }
  if (threadIdx.x == 0) weight[jk] = w;
}

int rows_xfer(XferP p, const int* idx_host, int n_rows, bool idx_is_dst, hipStream_t st) {
  EC_REQUIRE(p.n_seg > 0 && p.n_seg <= 8 && n_rows >= 0 && idx_host, -1, "rows_xfer: bad arguments");
  int ny = 0;
  for (int i = 0; i < p.n_seg; ++i) {
    XferSeg& S = p.seg[i];
    EC_REQUIRE(S.dst && S.src && S.row_bytes > 0 && S.n_outer > 0, -1, "rows_xfer: bad segment");
    const uintptr_t bits = (uintptr_t)S.dst | (uintptr_t)S.src | (uintptr_t)S.row_bytes | (uintptr_t)S.dst_os | (uintptr_t)S.src_os;
    S.align = (bits % 16 == 0) ? 16 : (bits % 4 == 0) ? 4 : 1;
    ny += S.n_outer;
  }
  p.idx_is_dst = idx_is_dst ? 1 : 0;
  for (int r0 = 0; r0 < n_rows; r0 += XFER_MAX_ROWS) {   // the row indices travel as kernel arguments: no staging buffer, no copy
    const int n = std::min(XFER_MAX_ROWS, n_rows - r0);
    XferP q = p;
    for (int i = 0; i < n; ++i) q.idx[i] = idx_host[r0 + i];
    for (int i = 0; i < q.n_seg; ++i) {   // this chunk's identity side starts at row r0
      XferSeg& S = q.seg[i];
      if (idx_is_dst) S.src = (const char*)S.src + (long)r0 * S.row_bytes;
      else S.dst = (char*)S.dst + (long)r0 * S.row_bytes;
    }
    hipLaunchKernelGGL(rows_xfer_kernel, dim3(n, ny), dim3(256), 0, st, q);
    EC_LAUNCH_CHECK();
  }
  return 0;
}

int preprocess_affine(const PreprocBatch& pb, int n, float* out, int H, hipStream_t st) {
  hipLaunchKernelGGL(preprocess_affine_kernel, dim3(cdiv((long)H * H, 256), n), dim3(256), 0, st, pb, out, H);
  EC_LAUNCH_CHECK();
  return 0;
}

int preprocess_affine_cv2(const PreprocBatchCv2& pb, int n, float* out, int H, hipStream_t st) {
  hipLaunchKernelGGL(preprocess_affine_cv2_kernel, dim3(cdiv((long)H * H, 256), n), dim3(256), 0, st, pb, out, H);
  EC_LAUNCH_CHECK();
  return 0;
}

int msra_targets(const float* joints, const float* visible, float* target, float* weight, int n_kpts_total, const MsraP& mp,
                 hipStream_t st) {
  hipLaunchKernelGGL(msra_target_kernel, dim3(n_kpts_total), dim3(256), 0, st, joints, visible, target, weight, mp);
  EC_LAUNCH_CHECK();
  return 0;
}

int mean_over(float* dst, const float* src, long stride, int n, long count, hipStream_t st) {
  hipLaunchKernelGGL(mean_over_kernel, dim3(cdiv(count, 256)), dim3(256), 0, st, dst, src, stride, n, count);
  EC_LAUNCH_CHECK();
  return 0;
}

int pack_x2(const float* x, long ldx, void* y, long ldy16, int rows, int cols, hipStream_t st) {
  EC_REQUIRE(cols % 4 == 0 && ldx % 4 == 0 && ldy16 >= 2l * cols && ldy16 % 2 == 0, -1, "pack_x2: cols % 4 == 0, ldy >= 2 cols");
  hipLaunchKernelGGL(pack_x2_kernel, dim3(cdiv((long)rows * (cols / 4), 256)), dim3(256), 0, st, x, ldx, (char*)y, ldy16, rows, cols / 4);
  EC_LAUNCH_CHECK();
  return 0;
}

int f32_to_bf16(const float* src, bf16_t* dst, long n, hipStream_t st, int f16) {
  if (n % 4 == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 8) == 0) {
    if (f16) hipLaunchKernelGGL(f32_to_bf16_x4_kernel<true>, dim3(cdiv(n / 4, 256)), dim3(256), 0, st, src, dst, n / 4);
    else hipLaunchKernelGGL(f32_to_bf16_x4_kernel<false>, dim3(cdiv(n / 4, 256)), dim3(256), 0, st, src, dst, n / 4);
    EC_LAUNCH_CHECK();
    return 0;
  }
  if (f16) hipLaunchKernelGGL(f32_to_bf16_kernel<true>, dim3(cdiv(n, 256)), dim3(256), 0, st, src, dst, n);
  else hipLaunchKernelGGL(f32_to_bf16_kernel<false>, dim3(cdiv(n, 256)), dim3(256), 0, st, src, dst, n);
  EC_LAUNCH_CHECK();
  return 0;
}

int transpose_pad_bf16(const float* src, bf16_t* dst, int B, int L, int E, int Lp, hipStream_t st) {
  hipLaunchKernelGGL(transpose_pad_bf16_kernel, dim3(cdiv((long)E * Lp, 256), B), dim3(256), 0, st, src, dst, L, E, Lp);
  EC_LAUNCH_CHECK();
  return 0;
}

int im2col14(const float* img, void* patches, int out_bf16, int n_img, int H, int W, int gh, int gw, int Kp, hipStream_t st) {
  EC_REQUIRE(Kp % 8 == 0 && Kp <= 1024, -1, "im2col14: padded row length must be a multiple of 8, at most 1024");
  const dim3 grid(n_img * (gh * gw + 1));
  if (out_bf16 == 5) hipLaunchKernelGGL(im2col14_kernel<5>, grid, dim3(256), 0, st, img, patches, H, W, gh, gw, Kp);
  else if (out_bf16 == 4) hipLaunchKernelGGL(im2col14_kernel<4>, grid, dim3(256), 0, st, img, patches, H, W, gh, gw, Kp);
  else if (out_bf16 == 3) hipLaunchKernelGGL(im2col14_kernel<3>, grid, dim3(256), 0, st, img, patches, H, W, gh, gw, Kp);
  else if (out_bf16 == 2) hipLaunchKernelGGL(im2col14_kernel<2>, grid, dim3(256), 0, st, img, patches, H, W, gh, gw, Kp);
  else if (out_bf16) hipLaunchKernelGGL(im2col14_kernel<1>, grid, dim3(256), 0, st, img, patches, H, W, gh, gw, Kp);
  else hipLaunchKernelGGL(im2col14_kernel<0>, grid, dim3(256), 0, st, img, patches, H, W, gh, gw, Kp);
  EC_LAUNCH_CHECK();
  return 0;
}

int set_cls_rows(float* x, long ldx, const float* cls, const float* pos0, int n_img, int T, int C, hipStream_t st) {
  hipLaunchKernelGGL(set_cls_kernel, dim3(n_img), dim3(256), 0, st, x, ldx, cls, pos0, T, C);
  EC_LAUNCH_CHECK();
  return 0;
}

int nchw_to_tokens(const float* src, float* dst, int n, int C, int HW, hipStream_t st) {
  hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(HW, 32), cdiv(C, 32), n), dim3(256), 0, st, src, dst, C, HW);
  EC_LAUNCH_CHECK();
  return 0;
}
int tokens_to_nchw(const float* src, float* dst, int n, int C, int HW, hipStream_t st) {
  hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(C, 32), cdiv(HW, 32), n), dim3(256), 0, st, src, dst, HW, C);
  EC_LAUNCH_CHECK();
  return 0;
}

int pool_gather(const float* target, const float* mask_s, float inv_shots, const float* F, float* pooled, float beta, int bs, int K,
                int hm, int gh, int gw, int C, hipStream_t st) {
  const size_t lds = (size_t)(hm * hm + gh * hm + gh * gw + 2 * hm) * sizeof(float) + (size_t)(4 * hm + gh * gw) * sizeof(int) + 8 * sizeof(float);
  EC_REQUIRE(lds <= 64 * 1024, -1, "pool_gather: heatmap too large for LDS");
  hipLaunchKernelGGL(pool_gather_kernel<0>, dim3(bs * K), dim3(256), lds, st, target, mask_s, inv_shots, F, pooled, beta, K, hm, gh, gw, C,
                     (int*)nullptr, (int*)nullptr, (float*)nullptr);
  EC_LAUNCH_CHECK();
  return 0;
}

// The two halves of pool_gather (see pool_gather_kernel): tap lists from the heatmaps, then the gather.  tap_n [bs*K], tap_i / tap_w [bs*K, gh*gw].
int pool_taps(const float* target, const float* mask_s, float inv_shots, int* tap_n, int* tap_i, float* tap_w, int bs, int K, int hm, int gh,
              int gw, hipStream_t st) {
  const size_t lds = (size_t)(hm * hm + gh * hm + gh * gw + 2 * hm) * sizeof(float) + (size_t)(4 * hm + gh * gw) * sizeof(int) + 8 * sizeof(float);
  EC_REQUIRE(lds <= 64 * 1024, -1, "pool_taps: heatmap too large for LDS");
  hipLaunchKernelGGL(pool_gather_kernel<1>, dim3(bs * K), dim3(256), lds, st, target, mask_s, inv_shots, (const float*)nullptr,
                     (float*)nullptr, 0.f, K, hm, gh, gw, 0, tap_n, tap_i, tap_w);
  EC_LAUNCH_CHECK();
  return 0;
}

int pool_apply(const int* tap_n, const int* tap_i, const float* tap_w, const float* F, float* pooled, float beta, int bs, int K, int hm, int gh,
               int gw, int C, hipStream_t st) {
  const size_t lds = (size_t)(hm * hm + gh * hm + gh * gw + 2 * hm) * sizeof(float) + (size_t)(4 * hm + gh * gw) * sizeof(int) + 8 * sizeof(float);
  hipLaunchKernelGGL(pool_gather_kernel<2>, dim3(bs * K), dim3(256), lds, st, (const float*)nullptr, (const float*)nullptr, 0.f, F, pooled, beta,
                     K, hm, gh, gw, C, const_cast<int*>(tap_n), const_cast<int*>(tap_i), const_cast<float*>(tap_w));
  EC_LAUNCH_CHECK();
  return 0;
}

int adj_build(const int32_t* edges, const int32_t* offsets, const float* mask_s, float* valid, uint8_t* kmask,
              uint8_t* kmask_fixed, float* binary, float* adj_r1, int bs, int K, hipStream_t st) {
  EC_REQUIRE(K * K <= 64 * 1024, -1, "adj_build: K too large");
  hipLaunchKernelGGL(adj_build_kernel, dim3(bs, 5), dim3(256), (size_t)K * K, st, edges, offsets, mask_s, valid, kmask, kmask_fixed,
                     binary, adj_r1, K);
  EC_LAUNCH_CHECK();
  return 0;
}

int rowplan(const float* mask_s, int bs, int ns, int K, int* plan, int* rowmap, int* fan_base, unsigned long long* fan_bits, hipStream_t st) {
  EC_REQUIRE(K >= 1 && K <= 128 && ns >= 1 && ns <= 8192 && bs >= 1, -1, "rowplan: K <= 128, samples <= 8192");
  hipLaunchKernelGGL(rowplan_kernel, dim3(1), dim3(1024), (size_t)ns * 2 * sizeof(int), st, mask_s, bs, ns, K, plan, rowmap, fan_base, fan_bits);
  EC_LAUNCH_CHECK();
  return 0;
}

int adj_gt(const float* adj_r1, const float* valid, float* adj_out, float* adj1, int bs, int K, hipStream_t st) {
  hipLaunchKernelGGL(adj_gt_kernel, dim3(cdiv((long)K * K, 256), bs), dim3(256), 0, st, adj_r1, valid, adj_out, adj1, K);
  EC_LAUNCH_CHECK();
  return 0;
}

int rownorm(const float* x, float* y, int rows, int cols, hipStream_t st) {
  hipLaunchKernelGGL(rownorm_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, x, y, rows, cols);
  EC_LAUNCH_CHECK();
  return 0;
}

int adj_combine(const float* P, const float* binary, const float* valid, const float* zc_w, const float* zc_b, float* adj_out,
                float* adj1, float* attn_adj, int bs, int K, hipStream_t st, int gram) {
  EC_REQUIRE(K <= 256, -1, "adj_combine: K must be <= 256");
  hipLaunchKernelGGL(adj_combine_kernel, dim3(bs, (K + 3) / 4), dim3(256), 0, st, P, binary, valid, zc_w, zc_b, adj_out, adj1, attn_adj, bs, K, gram);
  EC_LAUNCH_CHECK();
  return 0;
}

int set_identity(float* dst, int bs, int K, hipStream_t st) {
  hipLaunchKernelGGL(set_identity_kernel, dim3(cdiv(K * K, 256), bs), dim3(256), 0, st, dst, K);
  EC_LAUNCH_CHECK();
  return 0;
}

int bias_mlp(const float* attn_adj, const float* w1, const float* b1, const float* w2, const float* b2, float* out, int hops1,
             int hidden, int nhead, int bs, int K, hipStream_t st) {
  EC_REQUIRE(hops1 <= 8 && hidden <= 16, -1, "bias_mlp: unsupported MLP size");
  if (hops1 == 5 && hidden == 12 && nhead == 8) {
    BiasMlpLayers L = {};
    L.w1[0] = w1; L.b1[0] = b1; L.w2[0] = w2; L.b2[0] = b2;
    hipLaunchKernelGGL((bias_mlp_fixed_kernel<5, 12, 8>), dim3(cdiv((long)bs * K * K, 256), 1), dim3(256), 0, st, attn_adj, L, out, bs, K);
    EC_LAUNCH_CHECK();
    return 0;
  }
  const size_t lds = (size_t)(hidden * hops1 + hidden + nhead * hidden + nhead) * sizeof(float);
  hipLaunchKernelGGL(bias_mlp_kernel, dim3(cdiv((long)bs * K * K, 256)), dim3(256), lds, st, attn_adj, w1, b1, w2, b2, out, hops1,
                     hidden, nhead, bs, K);
  EC_LAUNCH_CHECK();
  return 0;
}

// All decoder layers' bias MLPs in one launch (they read the same Markov stack); returns 0 if the shape has no fused kernel.
int bias_mlp_layers(const float* attn_adj, const float* const* w1, const float* const* b1, const float* const* w2, const float* const* b2,
                    int n_layers, float* out, long out_stride, int hops1, int hidden, int nhead, int bs, int K, hipStream_t st) {
  if (!(hops1 == 5 && hidden == 12 && nhead == 8) || n_layers < 1 || n_layers > 4) return 0;
  BiasMlpLayers L = {};
  for (int i = 0; i < n_layers; ++i) { L.w1[i] = w1[i]; L.b1[i] = b1[i]; L.w2[i] = w2[i]; L.b2[i] = b2[i]; }
  L.stride = out_stride;
  hipLaunchKernelGGL((bias_mlp_fixed_kernel<5, 12, 8>), dim3(cdiv((long)bs * K * K, 256), n_layers), dim3(256), 0, st, attn_adj, L, out, bs,
                     K);
  EC_LAUNCH_CHECK();
  return 1;
}

int proposals(const float* sim, float* prop_loss, float* prop, int rows, int gh, int gw, hipStream_t st) {
  EC_REQUIRE(gh * gw <= 16 * 64, -1, "proposals: grid too large");
  hipLaunchKernelGGL(proposals_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, sim, prop_loss, prop, rows, gh, gw);
  EC_LAUNCH_CHECK();
  return 0;
}

int sincos_coords(const float* coords, const float* dim_t, float* out, long ldo, int rows, int num_feats, hipStream_t st) {
  hipLaunchKernelGGL(sincos_kernel, dim3(cdiv((long)rows * 2 * num_feats, 256)), dim3(256), 0, st, coords, dim_t, out, ldo, rows,
                     num_feats);
  EC_LAUNCH_CHECK();
  return 0;
}

int kpt_out(const float* h, long ldh, const float* W, const float* b, const float* prev, float* out, int rows, int d,
            hipStream_t st) {
  hipLaunchKernelGGL(kpt_out_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, h, ldh, W, b, prev, out, rows, d);
  EC_LAUNCH_CHECK();
  return 0;
}

}  // namespace ec
