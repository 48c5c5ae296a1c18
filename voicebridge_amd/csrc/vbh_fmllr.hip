// vbh_fmllr.hip -- fMLLR sufficient statistics on gfx950.
//
// Replaces (paths relative to /root/reference/kaldi-master/src/):
//   transform/fmllr-diag-gmm.cc:100-113   FmllrDiagGmmAccs::AccumulateForGmm (component posteriors of a pdf by
//                                          soft-max of its per-Gaussian log-likelihoods, gmm/diag-gmm.cc:528-543,601-615)
//   transform/fmllr-diag-gmm.cc:30-44     AccumulateFromPosteriors: per frame a = sum_g gamma_g miv_g, b = sum_g gamma_g iv_g (float)
//   transform/fmllr-diag-gmm.cc:562-596   CommitSingleFrameStats: beta += count, K += a x+^T, G_d += b[d] x+ x+^T (double)
//   gmmbin/gmm-est-fmllr.cc:36-52         AccumulateForUtterance (posteriors on transition-ids summed per pdf)
//
//   transform/fmllr-diag-gmm.cc:195-273   ComputeFmllrMatrixDiagGmmFull / FmllrInnerUpdate: the row-by-row update (fmllr_update_kernel below)
//   transform/fmllr-diag-gmm.cc:496-508   FmllrAuxFuncDiagGmm
//
// Three kernels.
//   fmllr_update_kernel: one wave per speaker, everything in double.  The 39 inverses of the 40 x 40 G_d (Gauss-Jordan with partial
//     pivoting in LDS, lane = column), then 40 iterations of: inverse(A) from scratch, and per row d the two 40 x 40 matrix-vector
//     products with inverse(G_d), the quadratic in alpha, and the rank-one (Sherman-Morrison) change of inverse(A) -- host/fmllr.cc's
//     Update(), operation for operation.  Sequential by nature (a row needs the rows before it): ~6 M cycles = 2.5 ms whatever
//     the number of speakers up to one per CU, against 4-5 ms of a host core PER SPEAKER; and the 128 MB of statistics of a
//     256-speaker batch never leave the device.
//   fmllr_frame_stats_kernel: one wave per frame, lane = feature dimension; the frame's (pdf, weight) pairs are
//     walked, per-Gaussian log-likelihoods are wave reductions, a/b/count leave as [T x D] / [T] float arrays.
//     Bytes per frame: 4D in + 8D + 4 out + the touched Gaussians (2 x 4D each, L2-resident model); HBM-bound, tiny.
//   fmllr_accumulate_kernel: the dense part.  Per speaker  G[d][i][j] = sum_t b_t[d] x+_t[i] x+_t[j]  is the
//     contraction  [D x T] . [T x (D+1)^2]  in double: v_mfma_f64_16x16x4_f64 with A = b (rows d), B built on the
//     fly as x+[i] * x+[j] (the product of two floats is exact in double), only the tiles of the lower triangle
//     (72 of 120 column tiles at D = 39) in groups of 4 waves x 3 tiles; K = sum_t a_t x+_t^T is three more
//     column tiles (A = a, B = x+) in a group of their own.  A workgroup owns one slice of one speaker's frames
//     and one group, keeps 9 accumulator tiles per wave in registers over the whole slice, fetches the next
//     64-frame stage into registers under the MFMAs of the current one, and adds its tiles to the speaker's
//     statistics with f64 atomics once at the end.  beta = sum_t count_t rides along in one wave.
//     Roofline: f64 MFMA.  Flops per frame = 2 * 48 * 16 * (72 + 3) = 115 200 (padded rows included);
//     bytes per frame = 4 * (D + 2D + 1) = 472: arithmetic intensity 244 flop/B, far above the f64 ridge (~10).
#include "vbh_internal.h"

#include <algorithm>
#include <cmath>

namespace {

using f64x4 = __attribute__((ext_vector_type(4))) double;

constexpr int TF = 64;          // frames staged in LDS per step
constexpr int NTW = 3;          // column tiles per wave: 9 accumulator tiles, 240 registers, two workgroups per CU (6 tiles: one per CU, 9.2 ms; 4: 8.1 ms; 3: 7.6 ms; 2: 8.1 ms; three waves per SIMD spill: 10.2 ms)
constexpr int NWAVE = 4;        // waves per workgroup (one per SIMD: five waves put two on one SIMD and ran 1.7x slower): 24 column tiles per group
constexpr int XLD = 49;         // padded row of the x+ image (48 columns used: 40 real + zeros)
constexpr int ALD = 49;         // padded row of the a / b image (48 rows of the MFMA M dimension)

struct Seg { long long row0; int nrows; int pad; };
struct Slice { int spk, seg_begin, seg_end, pad; };
struct ColTile { int type, i, j0, pad; };      // type 0: G tile (x+[i] * x+[j0..j0+15]); 1: K tile (x+[j0..j0+15]); -1: none

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// per-Gaussian log-likelihood of the frame held one dimension per lane (lanes >= D hold zeros): one wave reduction
__device__ __forceinline__ float gauss_loglike(const float *__restrict__ gconsts, const float *__restrict__ miv, const float *__restrict__ iv,
                                               int g, int D, int lane, float x, float x2) {
  float v = 0.f;
  if (lane < D) v = miv[(size_t)g * D + lane] * x + -0.5f * (iv[(size_t)g * D + lane] * x2);
  return gconsts[g] + wave_sum(v);
}

__global__ __launch_bounds__(256) void fmllr_frame_stats_kernel(
    const float *__restrict__ feats, long long T, int D,
    const long long *__restrict__ pair_off, const int *__restrict__ pair_pdf, const float *__restrict__ pair_w,
    const int *__restrict__ pdf_offs, const float *__restrict__ ali_gconsts, const float *__restrict__ ali_miv, const float *__restrict__ ali_iv,
    const float *__restrict__ miv, const float *__restrict__ iv,
    float *__restrict__ a_out, float *__restrict__ b_out, float *__restrict__ count_out) {
  const int lane = threadIdx.x & 63;
  const long long t = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  const float x = lane < D ? feats[t * D + lane] : 0.f, x2 = x * x;
  float a = 0.f, b = 0.f, count = 0.f;
  for (long long q = pair_off[t]; q < pair_off[t + 1]; q++) {
    const int p = pair_pdf[q];
    const float w = pair_w[q];
    const int g0 = pdf_offs[p], g1 = pdf_offs[p + 1];
    if (g1 - g0 <= 64) {
      // the usual case: lane k keeps the log-likelihood of the pdf's k-th Gaussian, so every Gaussian is evaluated once
      float mine = -INFINITY;
      for (int g = g0; g < g1; g++) { const float ll = gauss_loglike(ali_gconsts, ali_miv, ali_iv, g, D, lane, x, x2); if (lane == g - g0) mine = ll; }
      const float mx = wave_max(mine);
      const float e = lane < g1 - g0 ? expf(mine - mx) : 0.f;
      const float post_mine = e * (1.0f / wave_sum(e)) * w;
      count += wave_sum(post_mine);
      for (int g = g0; g < g1; g++) {
        const float post = __shfl(post_mine, g - g0);
        if (lane < D) { a += post * miv[(size_t)g * D + lane]; b += post * iv[(size_t)g * D + lane]; }
      }
    } else {
      float mx = -INFINITY;
      for (int g = g0; g < g1; g++) mx = fmaxf(mx, gauss_loglike(ali_gconsts, ali_miv, ali_iv, g, D, lane, x, x2));
      float sum = 0.f;
      for (int g = g0; g < g1; g++) sum += expf(gauss_loglike(ali_gconsts, ali_miv, ali_iv, g, D, lane, x, x2) - mx);
      const float inv = 1.0f / sum;
      for (int g = g0; g < g1; g++) {
        const float post = expf(gauss_loglike(ali_gconsts, ali_miv, ali_iv, g, D, lane, x, x2) - mx) * inv * w;
        count += post;
        if (lane < D) { a += post * miv[(size_t)g * D + lane]; b += post * iv[(size_t)g * D + lane]; }
      }
    }
  }
  if (lane < D) { a_out[t * D + lane] = a; b_out[t * D + lane] = b; }
  if (lane == 0) count_out[t] = count;
}

// the same per-frame terms from explicit Gaussian-level posteriors (gmm-est-fmllr-gpost: FmllrDiagGmmAccs::AccumulateFromPosteriors
// fed by gmm-post-to-gpost's output, K/gmmbin/gmm-est-fmllr-gpost.cc:36-50): pair q of a frame is a pdf and one posterior per Gaussian
__global__ __launch_bounds__(256) void fmllr_frame_stats_gpost_kernel(
    long long T, int D, const long long *__restrict__ pair_off, const int *__restrict__ pair_pdf, const long long *__restrict__ gpost_off,
    const float *__restrict__ gpost, const int *__restrict__ pdf_offs, const float *__restrict__ miv, const float *__restrict__ iv,
    float *__restrict__ a_out, float *__restrict__ b_out, float *__restrict__ count_out) {
  const int lane = threadIdx.x & 63;
  const long long t = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  float a = 0.f, b = 0.f, count = 0.f;
  for (long long q = pair_off[t]; q < pair_off[t + 1]; q++) {
    const int g0 = pdf_offs[pair_pdf[q]];
    const long long o0 = gpost_off[q], o1 = gpost_off[q + 1];
    float sum = 0.f;
    for (long long o = o0; o < o1; o++) {
      const float post = gpost[o];
      sum += post;
      if (lane < D && post != 0.f) { const size_t g = (size_t)(g0 + (int)(o - o0)); a += post * miv[g * D + lane]; b += post * iv[g * D + lane]; }
    }
    count += sum;
  }
  if (lane < D) { a_out[t * D + lane] = a; b_out[t * D + lane] = b; }
  if (lane == 0) count_out[t] = count;
}

__global__ __launch_bounds__(NWAVE * 64) void fmllr_accumulate_kernel(
    const float *__restrict__ feats, const float *__restrict__ a_in, const float *__restrict__ b_in, const float *__restrict__ count_in, int D,
    const Slice *__restrict__ slices, const Seg *__restrict__ segs, const ColTile *__restrict__ tiles,
    double *__restrict__ beta, double *__restrict__ K, double *__restrict__ G) {
  __shared__ double sX[TF * XLD];     // x+ of the staged frames, columns >= D+1 zero
  __shared__ double sB[TF * ALD];     // A operand of the staged frames: b for the groups of G tiles, a for the group of K tiles; columns >= D zero
  constexpr int NT = NWAVE * 64;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int n1 = D + 1;
  const Slice sl = slices[blockIdx.x];
  const bool kgroup = blockIdx.y == gridDim.y - 1;      // the last group holds the K tiles (its A operand is a, not b)
  const float *__restrict__ asrc = kgroup ? a_in : b_in;
  const bool counts = kgroup && wid == 1;               // this wave also sums the frame counts (beta)
  ColTile ct[NTW];
#pragma unroll
  for (int q = 0; q < NTW; q++) ct[q] = tiles[(blockIdx.y * NWAVE + wid) * NTW + q];
  f64x4 acc[NTW][3];
#pragma unroll
  for (int q = 0; q < NTW; q++)
#pragma unroll
    for (int m = 0; m < 3; m++) acc[q][m] = f64x4{0.0, 0.0, 0.0, 0.0};
  double cnt = 0.0;
  const int r = lane & 15, kk = lane >> 4;
  // the stage after the current one is fetched into registers while the MFMAs of the current one run
  constexpr int NLD = (TF * 48 + NT - 1) / NT;     // elements of the stage per thread
  float px[NLD], pb[NLD];
  float pcnt = 0.f;
  int sg = sl.seg_begin, f0 = 0;
  Seg seg = segs[sg];
  auto fetch = [&](const Seg &sgm, int ff0) {
    const int nfr = min(TF, sgm.nrows - ff0);
#pragma unroll
    for (int e = 0; e < NLD; e++) {
      const int idx = tid + e * NT, f = idx / 48, c = idx - f * 48;
      const long long row = sgm.row0 + ff0 + f;
      px[e] = 0.f; pb[e] = 0.f;
      if (f < nfr) {           // (f < TF follows: nfr <= TF)
        if (c < D) { px[e] = feats[row * D + c]; pb[e] = asrc[row * D + c]; }
        else if (c == D) px[e] = 1.0f;
      }
    }
    pcnt = 0.f;
    if (counts && lane < nfr) pcnt = count_in[sgm.row0 + ff0 + lane];      // TF == 64: one frame per lane
  };
  fetch(seg, 0);
  bool more = true;
  while (more) {
    const int nf = min(TF, seg.nrows - f0);
    __syncthreads();        // the previous stage's readers are done
#pragma unroll
    for (int e = 0; e < NLD; e++) {
      const int idx = tid + e * NT, f = idx / 48, c = idx - f * 48;
      if (f < TF) { sX[f * XLD + c] = (double)px[e]; sB[f * ALD + c] = (double)pb[e]; }
    }
    cnt += (double)pcnt;
    __syncthreads();
    // next stage: the rest of this segment, else the next segment of the slice
    f0 += TF;
    if (f0 >= seg.nrows) { sg++; f0 = 0; if (sg < sl.seg_end) seg = segs[sg]; else more = false; }
    if (more) fetch(seg, f0);
    for (int k0 = 0; k0 < TF; k0 += 4) {
      if (k0 >= nf) break;                       // (uniform) the rest of the stage is zeros
      const double *xr = &sX[(k0 + kk) * XLD], *br = &sB[(k0 + kk) * ALD];
      const double b0 = br[r], b1 = br[16 + r], b2 = br[32 + r];
#pragma unroll
      for (int q = 0; q < NTW; q++) {
        if (ct[q].type < 0) continue;            // wave-uniform
        const double xj = xr[ct[q].j0 + r];
        const bool kt = ct[q].type == 1;
        const double bv = kt ? xj : xr[ct[q].i] * xj;
        acc[q][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, bv, acc[q][0], 0, 0, 0);
        acc[q][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, bv, acc[q][1], 0, 0, 0);
        acc[q][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(b2, bv, acc[q][2], 0, 0, 0);
      }
    }
  }
  // ---- add this slice's tiles to the speaker's statistics.  C/D map of the f64 MFMA: col = lane & 15,
  //      row = (lane >> 4) + 4 * reg
  double *Gs = G + (size_t)sl.spk * D * n1 * n1, *Ks = K + (size_t)sl.spk * D * n1;
#pragma unroll
  for (int q = 0; q < NTW; q++) {
    if (ct[q].type < 0) continue;
    const int j = ct[q].j0 + r;
#pragma unroll
    for (int m = 0; m < 3; m++)
#pragma unroll
      for (int v = 0; v < 4; v++) {
        const int d = m * 16 + kk + 4 * v;
        const double val = acc[q][m][v];
        if (d >= D || j >= n1 || val == 0.0) continue;
        if (ct[q].type == 0) { if (j <= ct[q].i) atomicAdd(&Gs[((size_t)d * n1 + ct[q].i) * n1 + j], val); }
        else atomicAdd(&Ks[(size_t)d * n1 + j], val);
      }
  }
  if (counts) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (lane == 0 && cnt != 0.0) atomicAdd(&beta[sl.spk], cnt);
  }
}


// ---- the row-by-row update (host/fmllr.cc FmllrAccs::Update, "full"), one wave per speaker
constexpr int ULD = 49;          // padded row of the work matrices in LDS (n <= 48)

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// in-place inverse of the n x n matrix m (LDS, row stride ULD) by Gauss-Jordan elimination with partial pivoting, lane = column;
// returns log|det| (every lane), sets *singular.  The pivot of a column is the FIRST row of the largest magnitude, as the host's loop finds it.
__device__ double invert_lds(double *m, int n, int lane, int *perm, bool *singular) {
  double logdet = 0.0;
  for (int c = 0; c < n; c++) {
    // pivot: rows c..n-1 of column c, one per lane
    const double mag = (lane >= c && lane < n) ? fabs(m[lane * ULD + c]) : -1.0;
    double best = mag;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) best = fmax(best, __shfl_xor(best, o));
    const unsigned long long at = __ballot(mag == best && lane >= c && lane < n);
    const int piv = at ? __ffsll((long long)at) - 1 : c;
    if (!(best > 0.0)) { *singular = true; return 0.0; }
    if (lane == 0) perm[c] = piv;
    if (piv != c && lane < n) { const double a = m[c * ULD + lane], b = m[piv * ULD + lane]; m[c * ULD + lane] = b; m[piv * ULD + lane] = a; }
    __syncthreads();
    const double d = m[c * ULD + c];
    logdet += log(fabs(d));
    const double inv = 1.0 / d;
    __syncthreads();
    if (lane < n) m[c * ULD + lane] = (lane == c ? 1.0 : m[c * ULD + lane]) * inv;
    __syncthreads();
    const double pc = lane < n ? m[c * ULD + lane] : 0.0;
#pragma unroll 4
    for (int r = 0; r < n; r++) {
      if (r == c) continue;
      const double f = m[r * ULD + c];           // (every lane reads it before lane c overwrites it: one wave, its LDS operations in program order)
      if (lane < n) m[r * ULD + lane] = (lane == c ? 0.0 : m[r * ULD + lane]) - f * pc;
    }
    __syncthreads();
  }
  for (int c = n - 1; c >= 0; c--) {
    const int p = perm[c];
    if (p != c) {
      for (int r = 0; r < n; r++) {
        double v = 0.0;
        if (lane == c) v = m[r * ULD + p]; else if (lane == p) v = m[r * ULD + c];
        __syncthreads();
        if (lane == c || lane == p) m[r * ULD + lane] = v;
      }
      __syncthreads();
    }
  }
  return logdet;
}

// G of (speaker, d): the accumulation kernel fills the lower triangle
__device__ __forceinline__ double g_at(const double *__restrict__ g, int n1, int i, int j) { return i >= j ? g[i * n1 + j] : g[j * n1 + i]; }

// FmllrAuxFuncDiagGmm (fmllr-diag-gmm.cc:496-508) of W (LDS, row stride ULD): beta log|det A| + sum W.K - 1/2 sum_d w_d^T G_d w_d
__device__ double aux_func(const double *W, double *work, int *perm, int D, double beta, const double *__restrict__ K, const double *__restrict__ G, int lane, bool *singular) {
  const int n1 = D + 1;
  for (int r = 0; r < D; r++) if (lane < D) work[r * ULD + lane] = W[r * ULD + lane];
  __syncthreads();
  double obj = beta * invert_lds(work, D, lane, perm, singular);
  if (*singular) return 0.0;
  double wk = 0.0;
  for (int r = 0; r < D; r++) if (lane < n1) wk += W[r * ULD + lane] * K[r * n1 + lane];
  obj += wave_sum_d(wk);
  for (int d = 0; d < D; d++) {
    const double *g = G + (size_t)d * n1 * n1;
    double s = 0.0;
    if (lane < n1) { for (int j = 0; j < n1; j++) s += g_at(g, n1, lane, j) * W[d * ULD + j]; s *= W[d * ULD + lane]; }
    obj -= 0.5 * wave_sum_d(s);
  }
  return obj;
}

__global__ __launch_bounds__(64) void fmllr_update_kernel(int D, int num_iters, float min_count, const double *__restrict__ beta_all, const double *__restrict__ K_all,
                                                         const double *__restrict__ G_all, double *__restrict__ invG_all, float *__restrict__ xform_all,
                                                         float *__restrict__ impr_all, int *__restrict__ status_all) {
  __shared__ double Wm[48 * ULD], Mm[48 * ULD], work[48 * ULD];      // (work doubles as the row's inverse(G_d) during the iterations)
  __shared__ double row[64], rinv[64], delta[64], uvec[64], vvec[64];
  __shared__ int perm[64];
  const int s = blockIdx.x, lane = threadIdx.x, n1 = D + 1;
  const double beta = beta_all[s];
  float *xform = xform_all + (size_t)s * D * n1;
  if (lane == 0) { impr_all[s] = 0.0f; status_all[s] = 0; }
  if (!(beta > (double)min_count)) return;          // "Not updating fMLLR since below min-count"
  const double *K = K_all + (size_t)s * D * n1, *G = G_all + (size_t)s * D * n1 * n1;
  double *invG = invG_all + (size_t)s * D * n1 * n1;
  bool singular = false;
  // ---- inverse(G_d), stored TRANSPOSED (T[j][i] = inverse[i][j]) so that a lane's row of it is read across lanes in one line
  for (int d = 0; d < D; d++) {
    const double *g = G + (size_t)d * n1 * n1;
    for (int r = 0; r < n1; r++) if (lane < n1) work[r * ULD + lane] = g_at(g, n1, r, lane);
    __syncthreads();
    (void)invert_lds(work, n1, lane, perm, &singular);
    if (singular) { if (lane == 0) status_all[s] = 1; return; }
    double *t = invG + (size_t)d * n1 * n1;
    for (int j = 0; j < n1; j++) if (lane < n1) t[j * n1 + lane] = work[lane * ULD + j];
    __syncthreads();
  }
  __threadfence_block();
  for (int r = 0; r < D; r++) if (lane < n1) Wm[r * ULD + lane] = (double)xform[r * n1 + lane];
  __syncthreads();
  const float old_objf = (float)aux_func(Wm, work, perm, D, beta, K, G, lane, &singular);
  if (singular) { if (lane == 0) status_all[s] = 1; return; }
  // inverse(G_d) of the row being updated lives in LDS (`work`); the next row's is fetched into registers while this row is worked on
  // (read straight from memory in the matrix-vector loops, each row was eighty dependent global round trips of a lone wave: 30 ms per
  // call; staged at the top of the row: 20 ms; fetched one row ahead: see tests/test_gpu_fmllr.py's print)
  constexpr int NPF = (48 * 48 + 63) / 64;
  const int nn = n1 * n1;
  double nxt[NPF];
  auto fetch = [&](int d_) {
    const double *t_ = invG + (size_t)d_ * nn;
#pragma unroll
    for (int u = 0; u < NPF; u++) { const int e = u * 64 + lane; nxt[u] = e < nn ? t_[e] : 0.0; }
  };
  auto stage = [&]() {
#pragma unroll
    for (int u = 0; u < NPF; u++) { const int e = u * 64 + lane; if (e < nn) work[(e / n1) * ULD + (e % n1)] = nxt[u]; }
  };
  if (num_iters > 0) { fetch(0); __syncthreads(); stage(); __syncthreads(); }
  for (int iter = 0; iter < num_iters; iter++) {
    for (int r = 0; r < D; r++) if (lane < D) Mm[r * ULD + lane] = Wm[r * ULD + lane];
    __syncthreads();
    (void)invert_lds(Mm, D, lane, perm, &singular);
    if (singular) { if (lane == 0) status_all[s] = 1; return; }
    for (int d = 0; d < D; d++) {
      const double *k = K + (size_t)d * n1;          // (work[j * ULD + i] = inverse(G_d)[i][j])
      fetch(d + 1 < D ? d + 1 : 0);
      const double rw = lane < D ? Mm[lane * ULD + d] : 0.0;          // row[c] = inverse(A)[c][d]; row[D] = 0
      row[lane] = rw;
      __syncthreads();
      double ri = 0.0;
      if (lane < n1) {
#pragma unroll 8
        for (int j = 0; j < n1; j++) ri += work[j * ULD + lane] * row[j];
      }
      const double kl = lane < n1 ? k[lane] : 0.0;
      const double e1 = wave_sum_d(lane < n1 ? ri * rw : 0.0), e2 = wave_sum_d(lane < n1 ? ri * kl : 0.0);
      const double discr = sqrt(e2 * e2 + 4 * e1 * beta);
      const double alpha1 = (-e2 + discr) / (2 * e1), alpha2 = (-e2 - discr) / (2 * e1);
      const double auxf1 = beta * log(fabs(alpha1 * e1 + e2)) - 0.5 * alpha1 * alpha1 * e1;
      const double auxf2 = beta * log(fabs(alpha2 * e1 + e2)) - 0.5 * alpha2 * alpha2 * e1;
      const double alpha = auxf1 > auxf2 ? alpha1 : alpha2;
      __syncthreads();
      row[lane] = rw * alpha + kl;
      __syncthreads();
      double nw = 0.0;
      if (lane < n1) {
#pragma unroll 8
        for (int j = 0; j < n1; j++) nw += work[j * ULD + lane] * row[j];
      }
      const double dl = lane < D ? nw - Wm[d * ULD + lane] : 0.0;
      if (lane < n1) Wm[d * ULD + lane] = nw;
      delta[lane] = dl;
      // A' = A + e_d delta^T  ->  inverse(A') = M - (M e_d)(delta^T M) / (1 + delta^T M e_d)
      const double ul = lane < D ? Mm[lane * ULD + d] : 0.0;
      uvec[lane] = ul;
      const double denom = 1.0 + wave_sum_d(dl * ul);
      if (denom == 0.0 || !isfinite(denom)) { if (lane == 0) status_all[s] = 1; return; }
      __syncthreads();
      double sv = 0.0;
      if (lane < D) {
#pragma unroll 8
        for (int r = 0; r < D; r++) sv += delta[r] * Mm[r * ULD + lane];
      }
      sv /= denom;
      if (lane < D) {
#pragma unroll 8
        for (int r = 0; r < D; r++) Mm[r * ULD + lane] -= uvec[r] * sv;
      }
      __syncthreads();
      stage();
      __syncthreads();
    }
  }
  const float new_objf = (float)aux_func(Wm, work, perm, D, beta, K, G, lane, &singular);
  if (singular) { if (lane == 0) status_all[s] = 1; return; }
  const float impr = new_objf - old_objf;
  // "No applying fMLLR transform change because objective function did not increase."
  if (impr < 0.0f && !(fabsf(new_objf - old_objf) <= 0.001f * (fabsf(new_objf) + fabsf(old_objf)))) return;
  for (int r = 0; r < D; r++) if (lane < n1) xform[r * n1 + lane] = (float)Wm[r * ULD + lane];
  if (lane == 0) impr_all[s] = impr;
}

}  // namespace

extern "C" {

int vbh_fmllr_work(int32_t dim, int64_t T, double *bytes, double *flops) {
  const int n1 = dim + 1;
  int g_tiles = 0;
  for (int i = 0; i < n1; i++) for (int jt = 0; jt * 16 < n1; jt++) if (jt * 16 <= i) g_tiles++;
  const int k_tiles = (n1 + 15) / 16, row_tiles = (dim + 15) / 16;
  if (flops) *flops = (double)T * 2.0 * 16.0 * 16.0 * row_tiles * (g_tiles + k_tiles);
  if (bytes) *bytes = (double)T * 4.0 * (3.0 * dim + 1.0);
  return 0;
}

}  // extern "C"

// everything both entry points share: argument checks, slices, the per-frame pass chosen by the caller, the contraction, the copy back
template <class FrameStats>
static int fmllr_run(vbh_ctx *ctx, const vbh_gmm_view &vm, const float *d_feats, int32_t n_utt, const int64_t *h_frame_offsets, const int32_t *h_utt2spk,
                     int32_t n_spk, const int64_t *h_pair_offsets, const int32_t *h_pair_pdf, double *h_beta, double *h_K, double *h_G, float *kernel_ms,
                     FrameStats frame_stats) {
  VBH_CHECK(n_utt >= 1 && n_spk >= 1, "vbh_fmllr_accumulate: empty batch");
  const int D = vm.D, n1 = D + 1;
  VBH_CHECK(D >= 1 && D <= 47, "vbh_fmllr_accumulate: dim %d out of range [1,47]", D);
  const long long T = h_frame_offsets[n_utt];
  VBH_CHECK(T >= 1, "vbh_fmllr_accumulate: no frames");
  const long long n_pairs = h_pair_offsets[T];
  for (long long q = 0; q < n_pairs; q++) VBH_CHECK(h_pair_pdf[q] >= 0 && h_pair_pdf[q] < vm.P, "vbh_fmllr_accumulate: pdf-id out of range");
  for (int u = 0; u < n_utt; u++) VBH_CHECK(h_utt2spk[u] >= -1 && h_utt2spk[u] < n_spk, "vbh_fmllr_accumulate: speaker index out of range");
  // ---- column tiles: lower triangle of the (D+1) x (D+1) scatter, then K
  std::vector<ColTile> tiles;
  for (int i = 0; i < n1; i++) for (int jt = 0; jt * 16 < n1; jt++) if (jt * 16 <= i) tiles.push_back(ColTile{0, i, jt * 16, 0});
  const int per_group = NWAVE * NTW;              // column tiles of one workgroup
  const int g_groups = ((int)tiles.size() + per_group - 1) / per_group;       // 3 at D = 39 (72 tiles), 4 up to D = 47
  tiles.resize((size_t)g_groups * per_group, ColTile{-1, 0, 0, 0});
  for (int jt = 0; jt * 16 < n1; jt++) tiles.push_back(ColTile{1, 0, jt * 16, 0});       // K: a group of its own (A operand a instead of b)
  const int n_groups = g_groups + 1;
  tiles.resize((size_t)n_groups * per_group, ColTile{-1, 0, 0, 0});
  // ---- slices: every speaker's frames cut into pieces so that the grid fills the chip
  std::vector<std::vector<int>> spk_utts(n_spk);
  std::vector<long long> spk_frames(n_spk, 0);
  for (int u = 0; u < n_utt; u++) if (h_utt2spk[u] >= 0) { spk_utts[h_utt2spk[u]].push_back(u); spk_frames[h_utt2spk[u]] += h_frame_offsets[u + 1] - h_frame_offsets[u]; }
  const long long want_slices = 2LL * ctx->num_cus;
  long long total = 0;
  for (long long f : spk_frames) total += f;
  const long long per_slice = std::max<long long>(4 * TF, ((total + want_slices - 1) / want_slices + TF - 1) / TF * TF);
  std::vector<Seg> segs;
  std::vector<Slice> slices;
  for (int s = 0; s < n_spk; s++) {
    long long in_slice = 0;
    int seg_begin = (int)segs.size();
    for (int u : spk_utts[s]) {
      long long row = h_frame_offsets[u], left = h_frame_offsets[u + 1] - row;
      while (left > 0) {
        const long long take = std::min(left, per_slice - in_slice);
        segs.push_back(Seg{row, (int)take, 0});
        row += take; left -= take; in_slice += take;
        if (in_slice == per_slice) { slices.push_back(Slice{s, seg_begin, (int)segs.size(), 0}); seg_begin = (int)segs.size(); in_slice = 0; }
      }
    }
    if ((int)segs.size() > seg_begin) slices.push_back(Slice{s, seg_begin, (int)segs.size(), 0});
  }
  VBH_CHECK(!slices.empty(), "vbh_fmllr_accumulate: no frames assigned to a speaker");
  hipStream_t st = ctx->stream;
  // (work buffers from the context's scratch slots: no allocation or free per call, vbh_internal.h)
  ScratchBuf<long long> d_poff(ctx, VBH_SCR_FMLLR + 0); ScratchBuf<int> d_ppdf(ctx, VBH_SCR_FMLLR + 1);
  ScratchBuf<float> d_a(ctx, VBH_SCR_FMLLR + 2), d_b(ctx, VBH_SCR_FMLLR + 3), d_cnt(ctx, VBH_SCR_FMLLR + 4);
  ScratchBuf<Seg> d_segs(ctx, VBH_SCR_FMLLR + 5); ScratchBuf<Slice> d_slices(ctx, VBH_SCR_FMLLR + 6); ScratchBuf<ColTile> d_tiles(ctx, VBH_SCR_FMLLR + 7);
  ScratchBuf<double> d_beta(ctx, VBH_SCR_FMLLR + 8), d_K(ctx, VBH_SCR_FMLLR + 9), d_G(ctx, VBH_SCR_FMLLR + 10);
  static_assert(sizeof(long long) == sizeof(int64_t), "int64");
  if (d_poff.upload(reinterpret_cast<const long long *>(h_pair_offsets), (size_t)T + 1, st) || d_ppdf.upload(h_pair_pdf, (size_t)std::max<long long>(n_pairs, 1), st) ||
      d_a.alloc((size_t)T * D) || d_b.alloc((size_t)T * D) || d_cnt.alloc((size_t)T) ||
      d_segs.upload(segs.data(), segs.size(), st) || d_slices.upload(slices.data(), slices.size(), st) || d_tiles.upload(tiles.data(), tiles.size(), st) ||
      d_beta.alloc(n_spk) || d_K.alloc((size_t)n_spk * D * n1) || d_G.alloc((size_t)n_spk * D * n1 * n1))
    return -1;
  VBH_HIP(hipMemsetAsync(d_beta.p, 0, sizeof(double) * n_spk, st));
  VBH_HIP(hipMemsetAsync(d_K.p, 0, sizeof(double) * (size_t)n_spk * D * n1, st));
  VBH_HIP(hipMemsetAsync(d_G.p, 0, sizeof(double) * (size_t)n_spk * D * n1 * n1, st));
  const long long fblocks = (T + 3) / 4;
  VBH_CHECK(fblocks < 2147483647LL, "vbh_fmllr_accumulate: too many frames");
  if (frame_stats(st, (unsigned)fblocks, T, n_pairs, d_poff.p, d_ppdf.p, d_a.p, d_b.p, d_cnt.p)) return -1;
  VBH_HIP(hipGetLastError());
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (kernel_ms) { VBH_HIP(hipEventCreate(&e0)); VBH_HIP(hipEventCreate(&e1)); VBH_HIP(hipEventRecord(e0, st)); }
  hipLaunchKernelGGL(fmllr_accumulate_kernel, dim3((unsigned)slices.size(), n_groups), dim3(NWAVE * 64), 0, st, d_feats, d_a.p, d_b.p, d_cnt.p, D, d_slices.p, d_segs.p,
                     d_tiles.p, d_beta.p, d_K.p, d_G.p);
  VBH_HIP(hipGetLastError());
  if (kernel_ms) VBH_HIP(hipEventRecord(e1, st));
  ctx->fmllr_dim = D; ctx->fmllr_nspk = n_spk;       // (the statistics stay in the scratch slots: vbh_fmllr_update_full)
  VBH_HIP(hipMemcpyAsync(h_beta, d_beta.p, sizeof(double) * n_spk, hipMemcpyDeviceToHost, st));
  if (h_K) VBH_HIP(hipMemcpyAsync(h_K, d_K.p, sizeof(double) * (size_t)n_spk * D * n1, hipMemcpyDeviceToHost, st));
  if (h_G) VBH_HIP(hipMemcpyAsync(h_G, d_G.p, sizeof(double) * (size_t)n_spk * D * n1 * n1, hipMemcpyDeviceToHost, st));
  VBH_HIP(hipStreamSynchronize(st));
  if (kernel_ms) { VBH_HIP(hipEventElapsedTime(kernel_ms, e0, e1)); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); }
  // the kernel fills the lower triangles (j <= i); mirror them
  if (h_G) for (size_t m = 0; m < (size_t)n_spk * D; m++) {
    double *g = h_G + m * n1 * n1;
    for (int i = 0; i < n1; i++) for (int j = i + 1; j < n1; j++) g[(size_t)i * n1 + j] = g[(size_t)j * n1 + i];
  }
  return 0;
}

extern "C" {

int vbh_fmllr_accumulate(vbh_ctx *ctx, const vbh_gmm *ali_gmm, const vbh_gmm *adapt_gmm, const float *d_feats, int32_t n_utt,
                         const int64_t *h_frame_offsets, const int32_t *h_utt2spk, int32_t n_spk, const int64_t *h_pair_offsets,
                         const int32_t *h_pair_pdf, const float *h_pair_weight, double *h_beta, double *h_K, double *h_G,
                         float *kernel_ms) {
  VBH_CHECK(ctx && ali_gmm && adapt_gmm && d_feats && h_frame_offsets && h_utt2spk && h_pair_offsets && h_beta && (h_K != nullptr) == (h_G != nullptr),
            "vbh_fmllr_accumulate: NULL argument");
  if (vbh_use_device(ctx)) return -1;
  vbh_gmm_view va, vm;
  vbh_gmm_view_get(ali_gmm, &va);
  vbh_gmm_view_get(adapt_gmm, &vm);
  VBH_CHECK(va.D == vm.D && va.P == vm.P && va.G == vm.G, "vbh_fmllr_accumulate: the two models differ in shape");
  const int D = vm.D;
  ScratchBuf<float> d_pw(ctx, VBH_SCR_FMLLR + 11);
  return fmllr_run(ctx, vm, d_feats, n_utt, h_frame_offsets, h_utt2spk, n_spk, h_pair_offsets, h_pair_pdf, h_beta, h_K, h_G, kernel_ms,
                   [&](hipStream_t st, unsigned fblocks, long long T, long long n_pairs, const long long *d_poff, const int *d_ppdf, float *d_a, float *d_b, float *d_cnt) -> int {
                     if (d_pw.upload(h_pair_weight, (size_t)std::max<long long>(n_pairs, 1), st)) return -1;
                     hipLaunchKernelGGL(fmllr_frame_stats_kernel, dim3(fblocks), dim3(256), 0, st, d_feats, T, D, d_poff, d_ppdf, d_pw.p, va.pdf_offs, va.gconsts,
                                        va.miv, va.iv, vm.miv, vm.iv, d_a, d_b, d_cnt);
                     return 0;
                   });
}

int vbh_fmllr_accumulate_gpost(vbh_ctx *ctx, const vbh_gmm *adapt_gmm, const float *d_feats, int32_t n_utt, const int64_t *h_frame_offsets,
                               const int32_t *h_utt2spk, int32_t n_spk, const int64_t *h_pair_offsets, const int32_t *h_pair_pdf,
                               const int64_t *h_gpost_offsets, const float *h_gpost, const int32_t *h_pdf_num_gauss, double *h_beta, double *h_K,
                               double *h_G, float *kernel_ms) {
  VBH_CHECK(ctx && adapt_gmm && d_feats && h_frame_offsets && h_utt2spk && h_pair_offsets && h_gpost_offsets && h_beta && (h_K != nullptr) == (h_G != nullptr),
            "vbh_fmllr_accumulate_gpost: NULL argument");
  if (vbh_use_device(ctx)) return -1;
  vbh_gmm_view vm;
  vbh_gmm_view_get(adapt_gmm, &vm);
  const int D = vm.D;
  if (n_utt >= 1) {        // every pair must carry exactly one posterior per Gaussian of its pdf
    const long long n_pairs = h_pair_offsets[h_frame_offsets[n_utt]];
    for (long long q = 0; q < n_pairs; q++) {
      VBH_CHECK(h_pair_pdf[q] >= 0 && h_pair_pdf[q] < vm.P, "vbh_fmllr_accumulate_gpost: pdf-id out of range");
      VBH_CHECK(h_gpost_offsets[q + 1] - h_gpost_offsets[q] == h_pdf_num_gauss[h_pair_pdf[q]], "vbh_fmllr_accumulate_gpost: posterior vector does not match the pdf");
    }
  }
  ScratchBuf<long long> d_goff(ctx, VBH_SCR_FMLLR + 12); ScratchBuf<float> d_gp(ctx, VBH_SCR_FMLLR + 13);
  return fmllr_run(ctx, vm, d_feats, n_utt, h_frame_offsets, h_utt2spk, n_spk, h_pair_offsets, h_pair_pdf, h_beta, h_K, h_G, kernel_ms,
                   [&](hipStream_t st, unsigned fblocks, long long T, long long n_pairs, const long long *d_poff, const int *d_ppdf, float *d_a, float *d_b, float *d_cnt) -> int {
                     const long long n_g = h_gpost_offsets[n_pairs];
                     if (d_goff.upload(reinterpret_cast<const long long *>(h_gpost_offsets), (size_t)n_pairs + 1, st) ||
                         d_gp.upload(h_gpost, (size_t)std::max<long long>(n_g, 1), st)) return -1;
                     hipLaunchKernelGGL(fmllr_frame_stats_gpost_kernel, dim3(fblocks), dim3(256), 0, st, T, D, d_poff, d_ppdf, d_goff.p, d_gp.p, vm.pdf_offs, vm.miv,
                                        vm.iv, d_a, d_b, d_cnt);
                     return 0;
                   });
}

int vbh_fmllr_update_full(vbh_ctx *ctx, int32_t dim, int32_t n_spk, float min_count, int32_t num_iters, float *h_xform, float *h_impr,
                          float *kernel_ms) {
  VBH_CHECK(ctx && h_xform && h_impr, "vbh_fmllr_update_full: NULL argument");
  VBH_CHECK(ctx->fmllr_dim == dim && ctx->fmllr_nspk == n_spk && dim >= 1 && n_spk >= 1,
            "vbh_fmllr_update_full: the context holds no statistics of %d speakers in %d dimensions (the last vbh_fmllr_accumulate* call left %d / %d)", n_spk, dim,
            ctx->fmllr_nspk, ctx->fmllr_dim);
  VBH_CHECK(num_iters >= 0, "vbh_fmllr_update_full: num_iters < 0");
  if (vbh_use_device(ctx)) return -1;
  const int D = dim, n1 = D + 1;
  hipStream_t st = ctx->stream;
  ScratchBuf<double> d_beta(ctx, VBH_SCR_FMLLR + 8), d_K(ctx, VBH_SCR_FMLLR + 9), d_G(ctx, VBH_SCR_FMLLR + 10), d_invG(ctx, VBH_SCR_FMLLR + 14);
  ScratchBuf<float> d_out(ctx, VBH_SCR_FMLLR + 15);
  const size_t nx = (size_t)n_spk * D * n1;
  // (the three statistics slots are re-opened at the sizes they were filled with: nothing moves)
  if (d_beta.alloc(n_spk) || d_K.alloc(nx) || d_G.alloc(nx * n1) || d_invG.alloc(nx * n1) || d_out.alloc(nx + 2 * (size_t)n_spk)) return -1;
  float *d_x = d_out.p, *d_impr = d_out.p + nx;
  int *d_status = reinterpret_cast<int *>(d_out.p + nx + n_spk);
  VBH_HIP(hipMemcpyAsync(d_x, h_xform, sizeof(float) * nx, hipMemcpyHostToDevice, st));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (kernel_ms) { VBH_HIP(hipEventCreate(&e0)); VBH_HIP(hipEventCreate(&e1)); VBH_HIP(hipEventRecord(e0, st)); }
  hipLaunchKernelGGL(fmllr_update_kernel, dim3((unsigned)n_spk), dim3(64), 0, st, D, (int)num_iters, min_count, d_beta.p, d_K.p, d_G.p, d_invG.p, d_x, d_impr, d_status);
  VBH_HIP(hipGetLastError());
  if (kernel_ms) VBH_HIP(hipEventRecord(e1, st));
  std::vector<int> status(n_spk);
  VBH_HIP(hipMemcpyAsync(h_xform, d_x, sizeof(float) * nx, hipMemcpyDeviceToHost, st));
  VBH_HIP(hipMemcpyAsync(h_impr, d_impr, sizeof(float) * n_spk, hipMemcpyDeviceToHost, st));
  VBH_HIP(hipMemcpyAsync(status.data(), d_status, sizeof(int) * n_spk, hipMemcpyDeviceToHost, st));
  VBH_HIP(hipStreamSynchronize(st));
  if (kernel_ms) { VBH_HIP(hipEventElapsedTime(kernel_ms, e0, e1)); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); }
  for (int s_ = 0; s_ < n_spk; s_++) VBH_CHECK(status[s_] == 0, "fMLLR: singular matrix (speaker %d)", s_);
  return 0;
}

}  // extern "C"
