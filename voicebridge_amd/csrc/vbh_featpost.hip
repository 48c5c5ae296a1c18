// vbh_featpost.hip -- fused CMVN -> (deltas | splice) -> [global transform] -> [per-speaker
// affine transform] on gfx950, plus the per-speaker CMVN statistics reduction.
//
// Replaces the chain of temp-file stages that Decode() builds
// (VB/scr/steps/decode_gmm.cpp:201-293, VB/ = VoiceBridge/VoiceBridge/kaldi-win/):
//   ApplyCmvn       kaldi-master/src/transform/cmvn.cc:64-113
//   ComputeDeltas   kaldi-master/src/feat/feature-functions.cc:54-111,160-171
//   SpliceFrames    kaldi-master/src/feat/feature-functions.cc:205-226
//   TransformFeats  VB/src/featbin/transform-feats.cpp:89-107 (linear or affine, global or
//                   per speaker via utt2spk)
//   AccCmvnStats    kaldi-master/src/transform/cmvn.cc:30-62
//
// One workgroup owns 64 consecutive frames of one utterance: the frames and their halo are
// read once (13 floats per frame), every intermediate lives in LDS, the final D floats per
// frame are written once -> 4*(in_dim + out_dim) algorithmic bytes per frame.
// The two transforms are the path's only dense contractions ([64 x K] . [K x R], K <= 118);
// they run on the f32 MFMA (v_mfma_f32_32x32x2_f32, exact f32) out of LDS.
#include "vbh_internal.h"

#include <cmath>

namespace {

constexpr int kChunk = 64;
constexpr int kMaxDim = 128;     // largest intermediate feature dimension (splice: 9*13 = 117)

using f32x16 = __attribute__((ext_vector_type(16))) float;

struct PostParams {
  int in_dim, apply_cmvn, norm_vars, mode;
  int order, window;        // deltas
  int left, right;          // splice
  int mid_dim;              // after deltas/splice
  int lda_rows, lda_cols;   // 0 = none
  int spk_rows, spk_cols;   // 0 = none
  int out_dim;
  int halo;
  int nscale_w;             // width of a delta scales row = 2*order*window+1
};

// C[64 x R] = V[64 x K] . W^T (+ offset column), V in LDS as [64][ldv], W in LDS as [R][ldw]
// (row r = output dim), written to O[64][ldo].  Four waves: wave w owns rows (w&1)*32.. and
// column tiles (w>>1), (w>>1)+2, ... of 32 outputs.  K is walked two at a time
// (mfma_f32_32x32x2: lane l supplies A[i=l&31][k=l>>5] and B[k=l>>5][j=l&31]).
__device__ __forceinline__ void lds_transform_mfma(const float *V, int ldv, const float *W, int ldw, int R, int K,
                                                   bool affine, float *O, int ldo, int lane, int wid) {
  const int r0 = (wid & 1) * 32, khalf = lane >> 5, l31 = lane & 31;
  for (int ct = wid >> 1; ct * 32 < R; ct += 2) {
    const int col = ct * 32 + l31;
    const bool col_ok = col < R;
    f32x16 acc = {0};
    for (int k = 0; k < K; k += 2) {
      const int kk = k + khalf;
      const float a = (kk < K) ? V[(r0 + l31) * ldv + kk] : 0.f;
      const float b = (kk < K && col_ok) ? W[col * ldw + kk] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    const float off = (affine && col_ok) ? W[col * ldw + K] : 0.f;
    if (col_ok) {
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int row = r0 + (i & 3) + 8 * (i >> 2) + 4 * khalf;
        O[row * ldo + col] = acc[i] + off;
      }
    }
  }
}

__global__ __launch_bounds__(256) void featpost_kernel(PostParams p, const float *__restrict__ in,
                                                       const long long *__restrict__ frame_offsets,
                                                       const int *__restrict__ chunk_utt, const int *__restrict__ chunk_f0,
                                                       int n_chunks, const int *__restrict__ utt2spk,
                                                       const float *__restrict__ cmvn_norm,   // [n_spk][2][in_dim] offset, scale
                                                       const float *__restrict__ delta_scales, // [(order+1)][nscale_w]
                                                       const float *__restrict__ lda,          // [lda_rows][lda_cols]
                                                       const float *__restrict__ spk_xform,    // [n_spk][spk_rows][spk_cols]
                                                       float *__restrict__ out) {
  extern __shared__ __align__(16) float lds[];
  const int Din = p.in_dim, H = p.halo;
  const int raw_rows = kChunk + 2 * H;
  const int ld_mid = p.mid_dim | 1, ld_a = (p.lda_rows ? p.lda_rows : 1) | 1, ld_o = p.out_dim | 1;
  const int ldw1 = (p.lda_cols + 1) | 1, ldw2 = (p.spk_cols + 1) | 1;
  float *s_raw = lds;                               // [raw_rows][Din]
  float *s_mid = s_raw + raw_rows * Din;            // [64][ld_mid]
  float *s_a = s_mid + kChunk * ld_mid;             // [64][ld_a]   after the global transform
  float *s_o = s_a + kChunk * ld_a;                 // [64][ld_o]   after the speaker transform
  float *s_w1 = s_o + kChunk * ld_o;                // [lda_rows][ldw1]
  float *s_w2 = s_w1 + p.lda_rows * ldw1;           // [spk_rows][ldw2]
  float *s_sc = s_w2 + p.spk_rows * ldw2;           // delta scales
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;

  for (int i = tid; i < p.lda_rows * p.lda_cols; i += 256) s_w1[(i / p.lda_cols) * ldw1 + i % p.lda_cols] = lda[i];
  if (p.mode == 1) for (int i = tid; i < (p.order + 1) * p.nscale_w; i += 256) s_sc[i] = delta_scales[i];
  int cur_spk = -1;

  for (int chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    const int utt = chunk_utt[chunk], f0 = chunk_f0[chunk], spk = utt2spk[utt];
    const long long row0 = frame_offsets[utt];
    const int T = (int)(frame_offsets[utt + 1] - row0);
    const int nf = (T - f0 < kChunk) ? T - f0 : kChunk;
    __syncthreads();   // previous chunk's readers are done with LDS
    if (p.spk_rows && spk != cur_spk) {
      const float *w = spk_xform + (size_t)spk * p.spk_rows * p.spk_cols;
      for (int i = tid; i < p.spk_rows * p.spk_cols; i += 256) s_w2[(i / p.spk_cols) * ldw2 + i % p.spk_cols] = w[i];
      cur_spk = spk;
    }
    // ---- load frames f0-H .. f0+63+H (clamped to the utterance) and apply CMVN (cmvn.cc:104-112)
    for (int i = tid; i < raw_rows * Din; i += 256) {
      const int rr = i / Din, d = i - rr * Din;
      int t = f0 - H + rr;
      t = t < 0 ? 0 : (t >= T ? T - 1 : t);
      float x = in[(row0 + t) * Din + d];
      if (p.apply_cmvn) {
        const float *nm = cmvn_norm + (size_t)spk * 2 * Din;
        if (p.norm_vars) x = __fmul_rn(x, nm[Din + d]);
        x = __fadd_rn(x, nm[d]);
      }
      s_raw[i] = x;
    }
    __syncthreads();
    // ---- deltas (feature-functions.cc:90-111) or splice (:205-226) or copy
    for (int i = tid; i < kChunk * p.mid_dim; i += 256) {
      const int t = i / p.mid_dim, c = i - t * p.mid_dim;
      float v;
      if (p.mode == 1) {
        const int o = c / Din, d = c - o * Din, mo = o * p.window;
        const float *sc = s_sc + o * p.nscale_w;
        v = 0.f;
        for (int j = -mo; j <= mo; j++) {
          const float s = sc[j + mo];
          if (s != 0.f) v = fmaf(s, s_raw[(t + H + j) * Din + d], v);
        }
      } else if (p.mode == 2) {
        const int j = c / Din, d = c - j * Din;
        v = s_raw[(t + H + j - p.left) * Din + d];
      } else {
        v = s_raw[(t + H) * Din + c];
      }
      s_mid[t * ld_mid + c] = v;
    }
    __syncthreads();
    const float *cur = s_mid; int ld_cur = ld_mid, dim_cur = p.mid_dim;
    if (p.lda_rows) {
      lds_transform_mfma(cur, ld_cur, s_w1, ldw1, p.lda_rows, dim_cur, p.lda_cols == dim_cur + 1, s_a, ld_a, lane, wid);
      __syncthreads();
      cur = s_a; ld_cur = ld_a; dim_cur = p.lda_rows;
    }
    if (p.spk_rows) {
      lds_transform_mfma(cur, ld_cur, s_w2, ldw2, p.spk_rows, dim_cur, p.spk_cols == dim_cur + 1, s_o, ld_o, lane, wid);
      __syncthreads();
      cur = s_o; ld_cur = ld_o; dim_cur = p.spk_rows;
    }
    for (int i = tid; i < nf * p.out_dim; i += 256) {
      const int t = i / p.out_dim, c = i - t * p.out_dim;
      out[(row0 + f0 + t) * p.out_dim + c] = cur[t * ld_cur + c];
    }
  }
}

// per (speaker-ordered) utterance partial sums in double; one workgroup per utterance,
// atomically added to the speaker's stats (float products as in cmvn.cc:44-47).
__global__ __launch_bounds__(256) void cmvn_stats_kernel(const float *__restrict__ feats, int dim,
                                                         const long long *__restrict__ frame_offsets,
                                                         const int *__restrict__ utt2spk, double *__restrict__ stats) {
  const int utt = blockIdx.x, spk = utt2spk[utt];
  const long long row0 = frame_offsets[utt];
  const int T = (int)(frame_offsets[utt + 1] - row0);
  __shared__ double s_sum[256], s_sq[256];
  // thread -> (column d = tid % dim, row phase = tid / dim); rows strided by 256/dim
  const int rows_par = 256 / dim;
  const int d = threadIdx.x % dim, ph = threadIdx.x / dim;
  double sum = 0.0, sq = 0.0;
  if (ph < rows_par)
    for (int t = ph; t < T; t += rows_par) {
      const float x = feats[(row0 + t) * dim + d];
      sum += (double)x;
      sq += (double)(x * x);
    }
  s_sum[threadIdx.x] = sum; s_sq[threadIdx.x] = sq;
  __syncthreads();
  if (threadIdx.x < dim) {
    double a = 0.0, b = 0.0;
    for (int q = 0; q < rows_par; q++) { a += s_sum[q * dim + threadIdx.x]; b += s_sq[q * dim + threadIdx.x]; }
    double *st = stats + (size_t)spk * 2 * (dim + 1);
    atomicAdd(&st[threadIdx.x], a);
    atomicAdd(&st[(dim + 1) + threadIdx.x], b);
    if (threadIdx.x == 0) atomicAdd(&st[dim], (double)T);
  }
}

}  // namespace

extern "C" {

int32_t vbh_featpost_out_dim(const vbh_featpost_cfg *c) {
  int d = c->in_dim;
  if (c->mode == 1) d = c->in_dim * (c->delta_order + 1);
  else if (c->mode == 2) d = c->in_dim * (c->splice_left + c->splice_right + 1);
  if (c->lda_rows) d = c->lda_rows;
  if (c->spk_rows) d = c->spk_rows;
  return d;
}

int vbh_featpost(vbh_ctx *ctx, const vbh_featpost_cfg *c, const float *d_in, const int64_t *h_frame_offsets,
                 int32_t n_utt, const int32_t *h_utt2spk, int32_t n_spk, const double *h_cmvn_stats,
                 const float *h_lda, const float *h_spk_xform, float *d_out) {
  VBH_CHECK(ctx && c && h_frame_offsets, "vbh_featpost: NULL argument");
  VBH_CHECK(c->in_dim >= 1 && c->in_dim <= kMaxDim, "vbh_featpost: in_dim %d out of range", c->in_dim);
  VBH_CHECK(c->mode >= 0 && c->mode <= 2, "vbh_featpost: bad mode %d", c->mode);
  if (vbh_use_device(ctx)) return -1;
  PostParams p{};
  p.in_dim = c->in_dim; p.apply_cmvn = c->apply_cmvn; p.norm_vars = c->norm_vars; p.mode = c->mode;
  p.order = c->delta_order; p.window = c->delta_window; p.left = c->splice_left; p.right = c->splice_right;
  p.mid_dim = c->in_dim; p.halo = 0;
  if (c->mode == 1) {
    VBH_CHECK(c->delta_order >= 0 && c->delta_order <= 4 && c->delta_window >= 1 && c->delta_window <= 8, "vbh_featpost: bad delta options");
    p.mid_dim = c->in_dim * (c->delta_order + 1); p.halo = c->delta_order * c->delta_window;
  } else if (c->mode == 2) {
    VBH_CHECK(c->splice_left >= 0 && c->splice_right >= 0 && c->splice_left + c->splice_right <= 32, "vbh_featpost: bad splice options");
    p.mid_dim = c->in_dim * (c->splice_left + c->splice_right + 1);
    p.halo = c->splice_left > c->splice_right ? c->splice_left : c->splice_right;
  }
  VBH_CHECK(p.mid_dim <= kMaxDim, "vbh_featpost: intermediate dim %d > %d", p.mid_dim, kMaxDim);
  int dim = p.mid_dim;
  p.lda_rows = c->lda_rows; p.lda_cols = c->lda_cols;
  if (c->lda_rows) {
    VBH_CHECK(h_lda, "vbh_featpost: lda transform missing");
    VBH_CHECK(c->lda_cols == dim || c->lda_cols == dim + 1, "vbh_featpost: transform has %d columns, feature dim is %d", c->lda_cols, dim);
    VBH_CHECK(c->lda_rows <= kMaxDim, "vbh_featpost: lda_rows too large");
    dim = c->lda_rows;
  }
  p.spk_rows = c->spk_rows; p.spk_cols = c->spk_cols;
  if (c->spk_rows) {
    VBH_CHECK(h_spk_xform && h_utt2spk, "vbh_featpost: speaker transforms / utt2spk missing");
    VBH_CHECK(c->spk_cols == dim || c->spk_cols == dim + 1, "vbh_featpost: speaker transform has %d columns, feature dim is %d", c->spk_cols, dim);
    VBH_CHECK(c->spk_rows <= kMaxDim, "vbh_featpost: spk_rows too large");
    dim = c->spk_rows;
  }
  p.out_dim = dim;
  p.nscale_w = 2 * p.order * p.window + 1;
  if (c->apply_cmvn) VBH_CHECK(h_cmvn_stats && h_utt2spk && n_spk > 0, "vbh_featpost: cmvn stats / utt2spk missing");

  // chunks
  std::vector<long long> foff(n_utt + 1);
  std::vector<int> cu, cf, u2s(n_utt > 0 ? n_utt : 1, 0);
  for (int u = 0; u <= n_utt; u++) foff[u] = h_frame_offsets[u];
  for (int u = 0; u < n_utt; u++) {
    const int T = (int)(foff[u + 1] - foff[u]);
    VBH_CHECK(T >= 0, "vbh_featpost: frame offsets not monotone");
    if (h_utt2spk) { VBH_CHECK(h_utt2spk[u] >= 0 && h_utt2spk[u] < (n_spk > 0 ? n_spk : 1), "vbh_featpost: utt2spk[%d] out of range", u); u2s[u] = h_utt2spk[u]; }
    for (int f0 = 0; f0 < T; f0 += kChunk) { cu.push_back(u); cf.push_back(f0); }
  }
  if (cu.empty()) return 0;
  VBH_CHECK(d_in && d_out, "vbh_featpost: NULL device buffer");
  // CMVN offset/scale per speaker in double, stored as float (cmvn.cc:84-103)
  std::vector<float> norm;
  if (c->apply_cmvn) {
    const int D = c->in_dim;
    norm.resize((size_t)n_spk * 2 * D);
    for (int s = 0; s < n_spk; s++) {
      const double *st = h_cmvn_stats + (size_t)s * 2 * (D + 1);
      const double count = st[D];
      VBH_CHECK(count >= 1.0, "vbh_featpost: insufficient CMVN stats for speaker %d (count %g)", s, count);
      for (int d = 0; d < D; d++) {
        double mean = st[d] / count, offset, scale;
        if (!c->norm_vars) { scale = 1.0; offset = -mean; }
        else {
          double var = st[(D + 1) + d] / count - mean * mean;
          if (var < 1.0e-20) var = 1.0e-20;
          scale = 1.0 / sqrt(var);
          VBH_CHECK(!(scale != scale || 1 / scale == 0.0), "vbh_featpost: NaN/inf in CMVN for speaker %d", s);
          offset = -(mean * scale);
        }
        norm[(size_t)s * 2 * D + d] = (float)offset;
        norm[(size_t)s * 2 * D + D + d] = (float)scale;
      }
    }
  }
  // delta scales (feature-functions.cc:54-88)
  std::vector<float> scales;
  if (c->mode == 1) {
    const int W = p.nscale_w;
    scales.assign((size_t)(p.order + 1) * W, 0.0f);
    std::vector<int> dims(p.order + 1);
    // row i is stored centred: entry [j + i*window] for j in -i*window..i*window
    scales[0] = 1.0f; dims[0] = 1;
    for (int i = 1; i <= p.order; i++) {
      const float *prev = scales.data() + (size_t)(i - 1) * W;
      float *curp = scales.data() + (size_t)i * W;
      const int prev_off = (dims[i - 1] - 1) / 2, cur_off = prev_off + p.window;
      dims[i] = dims[i - 1] + 2 * p.window;
      float normalizer = 0.0f;
      for (int j = -p.window; j <= p.window; j++) {
        normalizer += (float)(j * j);
        for (int k = -prev_off; k <= prev_off; k++) curp[j + k + cur_off] += (float)j * prev[k + prev_off];
      }
      const float alpha = (float)(1.0 / normalizer);
      for (int k = 0; k < dims[i]; k++) curp[k] *= alpha;
    }
  }
  hipStream_t s = ctx->stream;
  const size_t n_lda = c->lda_rows ? (size_t)c->lda_rows * c->lda_cols : 0, n_spkx = c->spk_rows ? (size_t)n_spk * c->spk_rows * c->spk_cols : 0;
  vbh_arena *ar = vbh_arena_begin(ctx, vbh_arena_need(foff.size() * 8) + vbh_arena_need(cu.size() * 4) + vbh_arena_need(cf.size() * 4) +
                                           vbh_arena_need(u2s.size() * 4) + vbh_arena_need(norm.size() * 4) + vbh_arena_need(scales.size() * 4) +
                                           vbh_arena_need(n_lda * 4) + vbh_arena_need(n_spkx * 4));
  if (!ar) return -1;
  const long long *d_foff = vbh_arena_put(ar, foff.data(), foff.size());
  const int *d_cu = vbh_arena_put(ar, cu.data(), cu.size()), *d_cf = vbh_arena_put(ar, cf.data(), cf.size());
  const int *d_u2s = vbh_arena_put(ar, u2s.data(), u2s.size());
  const float *d_norm = vbh_arena_put(ar, norm.data(), norm.size()), *d_scales = vbh_arena_put(ar, scales.data(), scales.size());
  const float *d_lda = n_lda ? vbh_arena_put(ar, h_lda, n_lda) : nullptr;
  const float *d_spk = n_spkx ? vbh_arena_put(ar, h_spk_xform, n_spkx) : nullptr;
  if (vbh_arena_commit(ctx, ar)) return -1;
  const int raw_rows = kChunk + 2 * p.halo;
  const size_t lds_floats = (size_t)raw_rows * p.in_dim + (size_t)kChunk * (p.mid_dim | 1) +
                            (size_t)kChunk * ((p.lda_rows ? p.lda_rows : 1) | 1) + (size_t)kChunk * (p.out_dim | 1) +
                            (size_t)p.lda_rows * ((p.lda_cols + 1) | 1) + (size_t)p.spk_rows * ((p.spk_cols + 1) | 1) +
                            (size_t)(p.order + 1) * p.nscale_w + 8;
  const size_t lds_bytes = lds_floats * sizeof(float);
  VBH_CHECK(lds_bytes <= 160 * 1024, "vbh_featpost: LDS need %zu B exceeds 160 KiB", lds_bytes);
  VBH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(featpost_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  const int n_chunks = (int)cu.size();
  // chunks of one speaker are contiguous, so a grid-stride walk re-stages the speaker
  // transform rarely; cap the grid at a few workgroups per CU
  const int grid = n_chunks < 4 * ctx->num_cus ? n_chunks : 4 * ctx->num_cus;
  hipLaunchKernelGGL(featpost_kernel, dim3(grid), dim3(256), lds_bytes, s, p, d_in, d_foff, d_cu, d_cf, n_chunks,
                     d_u2s, d_norm, d_scales, d_lda, d_spk, d_out);
  VBH_HIP(hipGetLastError());
  return 0;
}

int vbh_cmvn_stats(vbh_ctx *ctx, const float *d_feats, int32_t dim, const int64_t *h_frame_offsets, int32_t n_utt,
                   const int32_t *h_utt2spk, int32_t n_spk, double *h_stats_out) {
  VBH_CHECK(ctx && h_frame_offsets && h_utt2spk && h_stats_out, "vbh_cmvn_stats: NULL argument");
  VBH_CHECK(dim >= 1 && dim <= 256, "vbh_cmvn_stats: dim %d out of range", dim);
  if (vbh_use_device(ctx)) return -1;
  const size_t n = (size_t)n_spk * 2 * (dim + 1);
  memset(h_stats_out, 0, n * sizeof(double));
  if (n_utt == 0) return 0;
  std::vector<long long> foff(h_frame_offsets, h_frame_offsets + n_utt + 1);
  std::vector<int> u2s(h_utt2spk, h_utt2spk + n_utt);
  for (int u = 0; u < n_utt; u++) VBH_CHECK(u2s[u] >= 0 && u2s[u] < n_spk, "vbh_cmvn_stats: utt2spk[%d] out of range", u);
  hipStream_t s = ctx->stream;
  ScratchBuf<long long> d_foff(ctx, VBH_SCR_CMVN + 0); ScratchBuf<int> d_u2s(ctx, VBH_SCR_CMVN + 1); ScratchBuf<double> d_stats(ctx, VBH_SCR_CMVN + 2);
  if (d_foff.upload(foff.data(), foff.size(), s) || d_u2s.upload(u2s.data(), u2s.size(), s) || d_stats.alloc(n)) return -1;
  VBH_HIP(hipMemsetAsync(d_stats.p, 0, n * sizeof(double), s));
  hipLaunchKernelGGL(cmvn_stats_kernel, dim3(n_utt), dim3(256), 0, s, d_feats, dim, d_foff.p, d_u2s.p, d_stats.p);
  VBH_HIP(hipGetLastError());
  VBH_HIP(hipMemcpyAsync(h_stats_out, d_stats.p, n * sizeof(double), hipMemcpyDeviceToHost, s));
  VBH_HIP(hipStreamSynchronize(s));
  return 0;
}

}  // extern "C"
