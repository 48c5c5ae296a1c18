// vbh_gmm.hip -- all-pdf diagonal-GMM log-likelihoods on gfx950.
//
// Replaces (paths relative to /root/reference/kaldi-master/src/):
//   gmm/decodable-am-diag-gmm.cc:28-72  DecodableAmDiagGmmUnmapped::LogLikelihoodZeroBased
//                                        (gconsts + means_invvars.x - 0.5 inv_vars.x^2, LogSumExp)
//   gmm/diag-gmm.cc:546-562              DiagGmm::LogLikelihoods (matrix form: the reference
//                                        itself evaluates a frame batch as two sgemm, K = D)
//   gmm/diag-gmm.cc:114-152              ComputeGconsts (host, at create)
//   matrix/kaldi-vector.cc:757-775       LogSumExp with the max + log(FLT_EPSILON) cutoff
//
// Formulation.  The per-Gaussian log-likelihoods of a tile of 64 frames are an exact-f32 MFMA contraction
//   C[g][t] = sum_k M[g][k] X[k][t],   M[g] = [ miv_g, -0.5 iv_g, gconst_g, 0.. ],   X[:,t] = [ x, x^2, 1, 0.. ]   (K = 2D+1, padded to 8)
// (v_mfma_f32_32x32x2_f32: a k-ordered fmaf chain, bit-identical to VALU f32), followed by a log-sum-exp over each pdf's own
// Gaussians.  Two kernels:
//   * gmm_rows_kernel (the product): the MODEL is the MFMA's A operand (row-blocks of 32 Gaussians straight from global memory, L2/MALL
//     resident: 4.8 MB at G = 15000, D = 39), the frames its B operand (in LDS); the accumulator layout then gives every lane the 32
//     Gaussians of ONE frame, the LogSumExp is a register loop, and a pdf may run from one row-block into the next (dense packing).
//     See the comment above the kernel;
//   * gmm_loglikes_kernel (round 1's form, behind VBH_GMM_V1): frames as the A operand, Gaussian tiles of 128 columns as B, the
//     accumulator tile spilled to LDS for the epilogue; a pdf with more than 128 Gaussians in column groups that gmm_fold_parts_kernel
//     combines.
//
// Roofline: per T frames  bytes = 4(2GD+G) + 4TD + 4TP,
// flops = TG(4D+2) + TG + TP.  At the reference sizes this is fp32-matrix bound (AI ~ 200
// flop/B), so bench.py prices this kernel against the 157.3 TFLOP/s f32 MFMA peak and also
// reports its HBM fraction.
#include "vbh_internal.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <mutex>
#include <type_traits>

namespace {

constexpr int TM = 64;        // frames per workgroup
constexpr int TN = 128;       // Gaussian columns per tile
constexpr int LDA = TM + 1;   // padded row of the A tile (k-major)
constexpr int LDC = TN + 1;   // padded row of the spilled accumulator tile
constexpr int MAXPRE = 16;    // float4 prefetch registers per thread: Kpad = 8*NPRE <= 128

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;   // native vector: stays in VGPRs across the tile loop

struct GmmTile { int g0, ng, p0, np; };

// Workgroup barrier for data exchanged through LDS only.  __syncthreads() is a full fence: it also
// waits for the wave's global stores to be acknowledged (vmcnt(0)), i.e. every tile paid the write
// latency of its 84-byte output runs at the barrier that frees the tile buffer.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// NPRE = Kpad/8: a B tile is Kpad*TN/4 = 256*NPRE float4, i.e. exactly NPRE per thread, so
// the prefetch registers need no predicate (a predicated register array goes to scratch).
template <int NPRE>
__global__ __launch_bounds__(256, 2) void gmm_loglikes_kernel(
    const float *__restrict__ feats, long long T, int D, int P,
    const float *__restrict__ Bp,          // [n_tiles][Kpad][TN]
    const GmmTile *__restrict__ tiles, int n_tiles, int tiles_per_group,
    const int *__restrict__ pdf_offsets,   // [P+1]
    float *__restrict__ out) {
  extern __shared__ __align__(16) float lds[];
  constexpr int Kpad = NPRE * 8;
  float *sA = lds;                                   // [Kpad][LDA]
  float *sB = sA + Kpad * LDA + ((4 - (Kpad * LDA) % 4) % 4);  // 16-byte aligned, [Kpad][TN], aliased by C
  float *sC = sB;                                    // [TM][LDC]
  const int bc_floats = (Kpad * TN > TM * LDC) ? Kpad * TN : TM * LDC;
  int *sOff = reinterpret_cast<int *>(sB + bc_floats);  // [TN+1] pdf column starts of the tile

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const long long t0 = (long long)blockIdx.x * TM;
  const int j_begin = blockIdx.y * tiles_per_group;
  int j_end = j_begin + tiles_per_group;
  if (j_end > n_tiles) j_end = n_tiles;

  // ---- A tile: [x, x^2, 1, 0...] for 64 frames, k-major
  for (int idx = tid; idx < TM * D; idx += 256) {
    const int t = idx / D, d = idx - t * D;
    float x = 0.f;
    if (t0 + t < T) x = feats[(t0 + t) * D + d];
    sA[d * LDA + t] = x;
    sA[(D + d) * LDA + t] = x * x;
  }
  for (int idx = tid; idx < (Kpad - 2 * D) * TM; idx += 256) {
    const int k = 2 * D + idx / TM, t = idx % TM;
    sA[k * LDA + t] = (k == 2 * D) ? 1.0f : 0.0f;
  }

  f32x4 pre[NPRE];
  {
    const f32x4 *src = reinterpret_cast<const f32x4 *>(Bp + (size_t)j_begin * Kpad * TN);
#pragma unroll
    for (int i = 0; i < NPRE; i++) pre[i] = src[tid + i * 256];
  }

  const int r = wid & 1, cpair = wid >> 1;   // wave tile: rows r*32.., column tiles 2*cpair, 2*cpair+1
  const int khalf = lane >> 5, l31 = lane & 31;

  // the tile's descriptor and its pdf column starts (np <= TN < 256: one per thread) are fetched one
  // tile ahead too: loaded at the top of an iteration they were two dependent global loads in
  // front of every tile's first barrier
  GmmTile tile = tiles[j_begin];
  int off = tid <= tile.np ? pdf_offsets[tile.p0 + tid] : 0;
  for (int j = j_begin; j < j_end; j++) {
    // ---- commit the prefetched B tile to LDS
    {
      f32x4 *dst = reinterpret_cast<f32x4 *>(sB);
#pragma unroll
      for (int i = 0; i < NPRE; i++) dst[tid + i * 256] = pre[i];
      if (tid <= tile.np) sOff[tid] = off - tile.g0;
    }
    lds_barrier();
    const GmmTile nt = tiles[j + 1];     // (one spare descriptor at the end)
    {   // next tile's loads stay in flight under the MFMA loop (the tile array has one
        // spare zero tile at the end, so the last iteration needs no branch)
      const f32x4 *src = reinterpret_cast<const f32x4 *>(Bp + (size_t)(j + 1) * Kpad * TN);
#pragma unroll
      for (int i = 0; i < NPRE; i++) pre[i] = src[tid + i * 256];
    }
    // ---- 64 x 128 x Kpad contraction, two 32x32 accumulators per wave
    f32x16 acc0 = {0}, acc1 = {0};
    const float *pa = sA + khalf * LDA + r * 32 + l31;
    const float *pb = sB + khalf * TN + cpair * 64 + l31;
    // operands are read from LDS one chunk (4 k-steps = 8 MFMAs) ahead of the MFMAs that use them:
    // with the reads issued right in front of their MFMAs the loop ran at LDS latency, not MFMA rate
    {
      constexpr int CH = 4, NCH = Kpad / 2 / CH;      // Kpad is a multiple of 8
      float fa[2][CH], fb0[2][CH], fb1[2][CH];
#pragma unroll
      for (int i = 0; i < CH; i++) { fa[0][i] = pa[i * 2 * LDA]; fb0[0][i] = pb[i * 2 * TN]; fb1[0][i] = pb[i * 2 * TN + 32]; }
#pragma unroll
      for (int c = 0; c < NCH; c++) {
        if (c + 1 < NCH) {
#pragma unroll
          for (int i = 0; i < CH; i++) {
            const int kk = (c + 1) * CH + i;
            fa[(c + 1) & 1][i] = pa[kk * 2 * LDA]; fb0[(c + 1) & 1][i] = pb[kk * 2 * TN]; fb1[(c + 1) & 1][i] = pb[kk * 2 * TN + 32];
          }
        }
        __builtin_amdgcn_sched_barrier(0);     // keep the scheduler from sinking those reads to their uses
#pragma unroll
        for (int i = 0; i < CH; i++) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c & 1][i], fb0[c & 1][i], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c & 1][i], fb1[c & 1][i], acc1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    lds_barrier();   // every wave is done reading sB
    const int noff = tid <= nt.np ? pdf_offsets[nt.p0 + tid] : 0;
    // ---- spill accumulators: C/D map col = lane&31, row = (i&3) + 8*(i>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int row = r * 32 + (i & 3) + 8 * (i >> 2) + 4 * khalf;
      sC[row * LDC + cpair * 64 + l31] = acc0[i];
      sC[row * LDC + cpair * 64 + 32 + l31] = acc1[i];
    }
    lds_barrier();
    // ---- per (frame, pdf) LogSumExp (kaldi-vector.cc:757-775).  A wave takes every 4th pdf of
    //      the tile and lane = frame: the pdf's column range is wave-uniform (scalar loop bounds,
    //      no integer division, no divergent trip counts) and the 64 rows are read conflict-free
    //      (LDC is odd).  The result overwrites the pdf's first column of its own row ...
    {
      const float kMinLogDiff = -15.9423847198486328125f;   // logf(FLT_EPSILON)
      float *row = sC + lane * LDC;
      for (int q = wid; q < tile.np; q += 4) {
        const int cs = __builtin_amdgcn_readfirstlane(sOff[q]), ce = __builtin_amdgcn_readfirstlane(sOff[q + 1]);
        // columns in chunks of 8 independent LDS reads (one read latency per chunk, not per column)
        const int n = ce - cs;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) { const float x = row[cs + (i < n ? i : 0)]; v[i] = i < n ? x : -INFINITY; }
        float m = fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7])));
        for (int c0 = cs + 8; c0 < ce; c0 += 8) {
          const int nn = ce - c0;
          float w[8];
#pragma unroll
          for (int i = 0; i < 8; i++) { const float x = row[c0 + (i < nn ? i : 0)]; w[i] = i < nn ? x : -INFINITY; }
          m = fmaxf(m, fmaxf(fmaxf(fmaxf(w[0], w[1]), fmaxf(w[2], w[3])), fmaxf(fmaxf(w[4], w[5]), fmaxf(w[6], w[7]))));
        }
        float res = m;
        if (m > -INFINITY) {
          const float cutoff = m + kMinLogDiff;
          float sum = 0.f;
#pragma unroll
          for (int i = 0; i < 8; i++) if (v[i] >= cutoff) sum += __expf(v[i] - m);    // (-inf padding fails the test)
          for (int c0 = cs + 8; c0 < ce; c0 += 8) {
            const int nn = ce - c0;
            float w[8];
#pragma unroll
            for (int i = 0; i < 8; i++) { const float x = row[c0 + (i < nn ? i : 0)]; w[i] = i < nn ? x : -INFINITY; }
#pragma unroll
            for (int i = 0; i < 8; i++) if (w[i] >= cutoff) sum += __expf(w[i] - m);
          }
          res = m + __logf(sum);
        }
        row[cs] = res;
      }
    }
    lds_barrier();
    // ---- ... and goes out with lanes along the pdfs of a frame (84-byte runs instead of one
    //      4-byte store per cache line)
    {
      const int np = tile.np;
      const int dt = 256 / np, dq = 256 - dt * np;
      int t = tid / np, q = tid - t * np;
      while (t < TM) {
        if (t0 + t < T) out[(t0 + t) * P + tile.p0 + q] = sC[t * LDC + sOff[q]];
        t += dt; q += dq;
        if (q >= np) { q -= np; t++; }
      }
    }
    lds_barrier();   // sC (= sB) is free for the next tile
    tile = nt; off = noff;
  }
}


// =====================================================================================
// gmm_rows_kernel -- the same contraction with the roles of the MFMA operands swapped: the MODEL is the A operand (32
// Gaussians = rows of a 32 x 32 block), the FRAMES are the B operand (columns).  What that buys:
//   * the accumulator layout puts a frame in a lane (column = lane & 31) and 16 Gaussians of the block in its registers;
//     one v_permlane32_swap per register pair hands each of the 64 lanes all 32 Gaussians of ONE of the tile's 64 frames,
//     so the LogSumExp of a pdf is a register loop (online form: one running maximum and one rescaled sum per pdf) -- the
//     accumulator tile never goes through LDS;
//   * a wave's model operand is a strip nobody else reads (32 Gaussians x K), so it comes straight from global memory
//     (L2/MALL resident) into registers, one row-block ahead; LDS holds only the 64 frames (read-only after the first
//     barrier) and a small per-wave staging area for the output runs;
//   * the four waves of a workgroup work on different row-blocks and never meet: NO barrier in the block loop.
// The model is packed once (vbh_gmm_create) into row-blocks of 32 Gaussians, the row-blocks into SUPERBLOCKS (runs of consecutive
// row-blocks, sbs[] = {first, end} row-block of each); wave w of a workgroup takes the superblocks w, w + 4, ... of its group and walks
// the row-blocks of a superblock in order.  Two packings:
//   * dense (the default LogSumExp, VBH_LSE_FAST): the Gaussians one after the other, a pdf may run over the end of a row-block into
//     the next one of the same superblock -- the online LogSumExp just carries its (maximum, sum) pair on in the lane -- so only the
//     last row-block of a superblock has unused rows (gap rows: Gaussians with gconst = -inf, which add exp(-inf) = 0); the
//     superblocks are ~8 row-blocks, near-equal, their number a multiple of 4 (the four waves get the same work).  Per block:
//     {bit r set = row r is the LAST Gaussian of a pdf, first pdf that ends in the block, number that end there (<= 16)};
//   * block-local (VBH_LSE_STRICT / _TWOPASS, whose two passes need all of a pdf at once): pdfs never straddle a row-block, a pdf
//     with more than 32 Gaussians is scored in parts of <= 32 that gmm_fold_parts_kernel combines; every superblock is one
//     row-block.  Per block: {bit r set = row r STARTS a pdf, first pdf, number of pdfs (<= 16)}.
struct RbMeta { unsigned mask; int p0; int np; int pad; };
struct __attribute__((packed, aligned(4))) f32x4u { float x, y, z, w; };      // four floats at a 4-byte-aligned address (one 16-byte store)
constexpr int LDP = 31;       // row of a wave's staging area: [64 frames][<= 15 pdfs waiting + <= 16 of the current block] (the flush keeps <= LDP - 16 waiting)

// LSE: how the LogSumExp over a pdf's Gaussians is evaluated (kaldi-vector.cc:757-775; vbh_gmm_set_lse_mode)
//   VBH_LSE_FAST (default): one pass over the lane's registers -- running maximum m and sum of exp(x - m), rescaled when the maximum
//     moves; of the two exponentials of the textbook update one is always exp(0), so ONE exp(-|x - m|) per Gaussian does it; no cutoff
//     (the terms the reference drops, below max + log FLT_EPSILON, are below 1.2e-7 of the sum each); f32, hardware exp2 / log2;
//   VBH_LSE_STRICT: the reference's arithmetic: two passes (the pdf's maximum, then the terms not below max + log FLT_EPSILON), expf of
//     the f32 difference, the sum in f64, max + log(sum) in f64, rounded once;
//   VBH_LSE_TWOPASS: the strict mode's two passes and cutoff in f32 with the hardware functions: measured 7 % slower than FAST
//     (83.5 vs 77.2 ms per 3.07 M frames: the maxima wait in LDS between the passes), kept for A/B runs.
template <int NC, int LSE>    // NC = Kpad / 8: a lane's strip of a row-block is NC float4 (4 k-steps each)
__global__ __launch_bounds__(256, 3) void gmm_rows_kernel(
    const float *__restrict__ feats, long long T, int D, int P,
    const f32x4 *__restrict__ Mp,            // [n_rb][NC][64 lanes]: Mp[(rb*NC + c)*64 + lane][j] = row (lane & 31), k = 2(4c + j) + (lane >> 5)
    const RbMeta *__restrict__ meta, const int *__restrict__ sbs, int n_sb, int sb_per_group,
    float *__restrict__ out, int ld) {                 // out rows are ld floats apart (ld >= P)
  extern __shared__ __align__(16) float lds[];
  constexpr int Kpad = NC * 8;
  float *sA = lds;                                   // [Kpad][LDA]: [x, x^2, 1, 0..] of 64 frames, k-major
  float *sStage = sA + Kpad * LDA + ((4 - (Kpad * LDA) % 4) % 4);
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long t0 = (long long)blockIdx.x * TM;
  const int sb_begin = blockIdx.y * sb_per_group;
  int sb_end = sb_begin + sb_per_group;
  if (sb_end > n_sb) sb_end = n_sb;
  for (int idx = tid; idx < TM * D; idx += 256) {
    const int t = idx / D, d = idx - t * D;
    float x = 0.f;
    if (t0 + t < T) x = feats[(t0 + t) * D + d];
    sA[d * LDA + t] = x;
    sA[(D + d) * LDA + t] = x * x;
  }
  for (int idx = tid; idx < (Kpad - 2 * D) * TM; idx += 256) {
    const int k = 2 * D + idx / TM, t = idx % TM;
    sA[k * LDA + t] = (k == 2 * D) ? 1.0f : 0.0f;
  }
  __syncthreads();
  int sb = sb_begin + wid;
  if (sb >= sb_end) return;                         // (no barrier below)
  const int khalf = lane >> 5, l31 = lane & 31;
  float *stage = sStage + wid * (TM * LDP);
  const float *pb = sA + khalf * LDA + l31;
  int rb = sbs[2 * sb], rb_end = sbs[2 * sb + 1];
  // where this wave's output runs go: out + (t0 ld + p0) is wave-uniform, the rest fits 32 bits
  const int fmax = T - t0 < TM ? (int)(T - t0) : TM;
  float *const out0 = out + t0 * ld;
  int pend = 0, pend_p0 = 0;              // VBH_LSE_FAST: finished pdfs waiting in the staging area, the first one's index
  const bool lines = (ld & 15) == 0 && (reinterpret_cast<size_t>(out) & 63) == 0;     // rows start on 64-byte boundaries
  f32x4 sa[NC], sb2[NC];
#pragma unroll
  for (int c = 0; c < NC; c++) sa[c] = Mp[((size_t)rb * NC + c) * 64 + lane];
  float cm = -3.0e38f, csum = 0.0f;       // VBH_LSE_FAST: the open pdf's running maximum and sum, carried from row-block to row-block
  // one row-block with its strip in `cur`; the next one's strip goes to `nxt` meanwhile (the caller swaps the two: no register copies)
  auto do_block = [&](f32x4 (&cur)[NC], f32x4 (&nxt)[NC]) __attribute__((always_inline)) -> bool {
    // the row-block after this one: the next of the superblock, or the first of the wave's next superblock
    int nrb = rb + 1, nsb = sb, nend = rb_end;
    if (nrb == rb_end) {
      nsb = sb + 4;
      if (nsb < sb_end) { nrb = sbs[2 * nsb]; nend = sbs[2 * nsb + 1]; } else nrb = -1;
    }
    // its strip: in flight under this block's MFMAs
    {
      const int rbn = nrb >= 0 ? nrb : rb;
#pragma unroll
      for (int c = 0; c < NC; c++) nxt[c] = Mp[((size_t)rbn * NC + c) * 64 + lane];
    }
    const RbMeta mt = meta[rb];
    const unsigned mask = (unsigned)__builtin_amdgcn_readfirstlane((int)mt.mask);
    const int p0 = __builtin_amdgcn_readfirstlane(mt.p0), np = __builtin_amdgcn_readfirstlane(mt.np);
    // ---- 32 Gaussians x 64 frames x Kpad: two 32 x 32 accumulators (frames 0-31, 32-63)
    f32x16 acc0 = {0}, acc1 = {0};
    {
      float b0[2][4], b1[2][4];
#pragma unroll
      for (int j = 0; j < 4; j++) { b0[0][j] = pb[j * 2 * LDA]; b1[0][j] = pb[j * 2 * LDA + 32]; }
#pragma unroll
      for (int c = 0; c < NC; c++) {
        if (c + 1 < NC) {
#pragma unroll
          for (int j = 0; j < 4; j++) { const int kk = (c + 1) * 4 + j; b0[(c + 1) & 1][j] = pb[kk * 2 * LDA]; b1[(c + 1) & 1][j] = pb[kk * 2 * LDA + 32]; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[c][j], b0[c & 1][j], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[c][j], b1[c & 1][j], acc1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- every lane gets the 32 rows of its own frame (frame = lane): lanes 0-31 give away their share of frames 32-63
    //      and take the rows 4-7, 12-15, ... of frames 0-31 from lanes 32-63, and the other way round
    float v0[16], v1[16];     // v0[i]: row (i&3) + 8(i>>2), v1[i]: that row + 4
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc0[i]), __float_as_uint(acc1[i]), false, false);
      v0[i] = __uint_as_float(r[0]); v1[i] = __uint_as_float(r[1]);
    }
    // ---- LogSumExp per pdf (kaldi-vector.cc:757-775).  The pdf boundaries are wave-uniform (mask), so the passes are straight
    //      register loops with scalar branches; a finished pdf's value goes to the lane's own row of the staging area.
#define VBH_ROW(r) ((((r) >> 2) & 1) ? v1[((r) & 3) + 4 * ((r) >> 3)] : v0[((r) & 3) + 4 * ((r) >> 3)])
    if (LSE == VBH_LSE_FAST) {
      // (the raw hardware exp2 / log2: the library forms of __expf / __logf add a range fix for denormal results -- a compare, a select
      // and an ldexp per call -- that a term of a sum whose largest term is 1 does not need; the maximum rides on the compare the sum needs)
      const float kLog2e = 1.44269504088896340736f, kLn2 = 0.69314718055994530942f;
      float m = cm, sum = csum;
      float *so = stage + lane * LDP;
      if (pend == 0) pend_p0 = p0;
      int q = pend;
#pragma unroll
      for (int r = 0; r < 32; r++) {
        const float x = VBH_ROW(r);
        // sum' = sum exp(m - m') + exp(x - m') with m' = max(m, x): one of the two factors is exp(0) = 1
        const float d = x - m;
        const float e = __builtin_amdgcn_exp2f(-fabsf(d) * kLog2e);
        const bool up = d > 0.0f;
        sum = up ? fmaf(sum, e, 1.0f) : sum + e;
        m = up ? x : m;
        if ((mask >> r) & 1u) {      // (wave-uniform) row r is the last Gaussian of a pdf
          so[q] = sum > 0.0f ? fmaf(__builtin_amdgcn_logf(sum), kLn2, m) : -INFINITY;
          q++;
          m = -3.0e38f; sum = 0.0f;
        }
      }
      cm = m; csum = sum;
      pend = q;
      // ---- out: the finished pdfs wait in the staging area until a frame's run is 16 of them or more (or the superblock ends: the
      //      wave's next one is elsewhere in the model): runs of 64+ bytes per frame, lanes along (frame, pdf) -- scattered 4-byte
      //      stores, one per pdf and lane, were measured to cost twice what the whole LogSumExp does
      if (pend >= 16 || nsb != sb) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // groups of four pdfs, lanes along (frame, group), 16 bytes each.  Inside a superblock the run goes up to the last 64-byte
        // boundary of the row, the pdfs behind it wait for the next run; superblocks start at pdfs that are multiples of 8: with ld a
        // multiple of 16 a run is whole 32-byte sectors and, but for a superblock's first and last, whole 64-byte halves of a line
        // (L2 writes back dirty 64-byte halves: one that two runs share is written twice).  At the end of a superblock the whole
        // run leaves, what does not fill a group of four from the lane that made it (lane = frame).
        const bool last = nsb != sb;
        // (a run that starts at a pdf that is no multiple of 4 -- VBH_GMM_CUT_STEP only: the packing cuts at multiples of 4 -- could
        // leave more than 15 pdfs waiting, and the next block's 16 would overrun the lane's row: then everything leaves, unaligned)
        int n_out = last ? pend : (((pend_p0 + pend) & ~15) - pend_p0) & ~3;
        if (pend - n_out > LDP - 16) n_out = pend & ~3;
        const int nch = n_out >> 2, n4 = nch << 2;
        if (nch > 0) {
          const unsigned inv = (65536u + (unsigned)nch - 1u) / (unsigned)nch;
          const int total = TM * nch;
          float *const ob = out0 + pend_p0;
          for (int idx = lane; idx < total; idx += 64) {
            const int f = (int)(((unsigned)idx * inv) >> 16), c = idx - f * nch;
            if (f < fmax) {
              const float *src = stage + f * LDP + 4 * c;
              if (lines) {      // whole 64-byte halves of lines, nothing of them to come later: streamed past the L2's allocation
                f32x4 v = {src[0], src[1], src[2], src[3]};
                __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(ob + f * ld + 4 * c));
              } else {
                f32x4u v; v.x = src[0]; v.y = src[1]; v.z = src[2]; v.w = src[3];
                *reinterpret_cast<f32x4u *>(ob + f * ld + 4 * c) = v;
              }
            }
          }
        }
        if (last) {
          if (lane < fmax)
            for (int j = n4; j < pend; j++) out0[lane * ld + pend_p0 + j] = so[j];
          pend = 0;
        } else {
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
          for (int j = n4; j < pend; j++) so[j - n4] = so[j];
          pend_p0 += n4;
          pend -= n4;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    } else {
      const float kMinLogDiff = -15.9423847198486328125f;   // logf(FLT_EPSILON)
      float *so = stage + lane * LDP;
      {
        float m = -INFINITY;
        int q = 0;
#pragma unroll
        for (int r = 0; r < 32; r++) {
          const float x = VBH_ROW(r);
          if (r > 0 && ((mask >> r) & 1u)) { so[q] = m; q++; m = x; }
          else m = fmaxf(m, x);
        }
        so[q] = m;
      }
      typedef typename std::conditional<LSE == VBH_LSE_STRICT, double, float>::type sum_t;
      auto finish = [](float m, sum_t sum) -> float {
        if (!(m > -INFINITY)) return -INFINITY;                  // a pdf of gap rows only (gconst = -inf)
        if (LSE == VBH_LSE_STRICT) return (float)((double)m + log((double)sum));
        return m + __logf((float)sum);
      };
      int q = 0;
      float m = so[0], cutoff = m + kMinLogDiff;
      sum_t sum = 0;
#pragma unroll
      for (int r = 0; r < 32; r++) {
        const float x = VBH_ROW(r);
        if (r > 0 && ((mask >> r) & 1u)) {
          so[q] = finish(m, sum);
          q++;
          m = so[q]; cutoff = m + kMinLogDiff; sum = 0;
        }
        if (x >= cutoff) sum += LSE == VBH_LSE_STRICT ? (sum_t)expf(x - m) : (sum_t)__expf(x - m);
      }
      so[q] = finish(m, sum);
    }
#undef VBH_ROW
    if (LSE != VBH_LSE_FAST) {
      // ---- out: runs of np pdfs per frame from the staging area, lanes along (frame, pdf)
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const unsigned inv = (65536u + (unsigned)np - 1u) / (unsigned)np;
      const int total = TM * np;
      float *const ob = out0 + p0;
      for (int idx = lane; idx < total; idx += 64) {
        const int f = (int)(((unsigned)idx * inv) >> 16), q = idx - f * np;
        if (f < fmax) ob[f * ld + q] = stage[f * LDP + q];
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    if (nrb < 0) return false;
    rb = nrb; sb = nsb; rb_end = nend;
    return true;
  };
  while (do_block(sa, sb2) && do_block(sb2, sa)) {}
}

// A pdf with more than TN Gaussians is scored as several parts of <= TN columns (each part is a
// "pdf" of the tile kernel above, written to a scratch [T x n_parts] block); this pass folds the
// parts of every pdf with one more LogSumExp (same cutoff rule, kaldi-vector.cc:757-775).
__global__ __launch_bounds__(256) void gmm_fold_parts_kernel(
    const float *__restrict__ parts, long long T, int n_parts, int P,
    const int *__restrict__ pdf_parts,     // [P+1] first part of every pdf
    float *__restrict__ out, int ld, int strict) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= T * P) return;
  const long long t = idx / P;
  const int p = (int)(idx - t * P);
  const int a = pdf_parts[p], b = pdf_parts[p + 1];
  const float *row = parts + t * n_parts;
  float m = row[a];
  for (int i = a + 1; i < b; i++) m = fmaxf(m, row[i]);
  float res = m;
  if (b - a > 1 && m > -INFINITY) {
    const float cutoff = m + -15.9423847198486328125f;
    if (strict) {
      double sum = 0.0;
      for (int i = a; i < b; i++) { const float v = row[i]; if (v >= cutoff) sum += (double)expf(v - m); }
      res = (float)((double)m + log(sum));
    } else {
      float sum = 0.f;
      for (int i = a; i < b; i++) { const float v = row[i]; if (v >= cutoff) sum += __expf(v - m); }
      res = m + __logf(sum);
    }
  }
  out[t * ld + p] = res;
}

// On-demand form (DecodableAmDiagGmmUnmapped::LogLikelihoodZeroBased, K/gmm/decodable-am-diag-gmm.cc:28-72): only the
// (frame, pdf) pairs a caller asks for -- e.g. the two or three pdfs a lattice holds at a frame (gmm-rescore-lattice).
// One wave per frame, lane = feature dimension; a Gaussian's log-likelihood is one wave reduction, lane k keeps the
// k-th Gaussian's value, LogSumExp across lanes with the reference's cutoff.  Bytes per pair: 2 x 4D per Gaussian of
// the pdf (model rows, L2-resident) + 4 out; the frame's 4D input is read once: HBM/L2-gather bound, AI ~ 0.5 flop/B.
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__global__ __launch_bounds__(256) void gmm_loglikes_pairs_kernel(
    const float *__restrict__ feats, long long T, int D, const long long *__restrict__ pair_off, const int *__restrict__ pair_pdf,
    const int *__restrict__ pdf_offs, const float *__restrict__ gconsts, const float *__restrict__ miv, const float *__restrict__ iv,
    float *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long long t = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= T) return;
  const float x = lane < D ? feats[t * D + lane] : 0.f, x2 = x * x;
  const float kMinLogDiff = -15.9423847198486328125f;
  for (long long q = pair_off[t]; q < pair_off[t + 1]; q++) {
    const int p = pair_pdf[q];
    const int g0 = pdf_offs[p], g1 = pdf_offs[p + 1];
    float res;
    if (g1 - g0 <= 64) {
      float mine = -INFINITY;
      for (int g = g0; g < g1; g++) {
        float v = 0.f;
        if (lane < D) v = miv[(size_t)g * D + lane] * x + -0.5f * (iv[(size_t)g * D + lane] * x2);
        const float ll = gconsts[g] + wsum(v);
        if (lane == g - g0) mine = ll;
      }
      const float mx = wmax(mine);
      const float e = (lane < g1 - g0 && mine >= mx + kMinLogDiff) ? expf(mine - mx) : 0.f;
      res = mx > -INFINITY ? mx + logf(wsum(e)) : mx;
    } else {
      float mx = -INFINITY;
      for (int g = g0; g < g1; g++) {
        float v = 0.f;
        if (lane < D) v = miv[(size_t)g * D + lane] * x + -0.5f * (iv[(size_t)g * D + lane] * x2);
        mx = fmaxf(mx, gconsts[g] + wsum(v));
      }
      float sum = 0.f;
      for (int g = g0; g < g1; g++) {
        float v = 0.f;
        if (lane < D) v = miv[(size_t)g * D + lane] * x + -0.5f * (iv[(size_t)g * D + lane] * x2);
        const float ll = gconsts[g] + wsum(v);
        if (ll >= mx + kMinLogDiff) sum += expf(ll - mx);
      }
      res = mx > -INFINITY ? mx + logf(sum) : mx;
    }
    if (lane == 0) out[q] = res;
  }
}

}  // namespace

struct vbh_gmm {
  vbh_ctx *ctx;
  int D, P, G, Kpad, n_tiles;
  std::vector<float> gconsts;
  std::vector<int> offs;
  DevBuf<float> Bp;
  DevBuf<GmmTile> tiles;
  DevBuf<int> d_offs;
  size_t lds_bytes;
  // pdfs with more than TN Gaussians: n_parts > P column groups, folded by gmm_fold_parts_kernel
  int n_parts;
  std::vector<int> part_offs;     // [n_parts+1] Gaussian column starts of the parts
  DevBuf<int> d_pdf_parts;        // [P+1] first part of every pdf
  DevBuf<float> part_scratch;     // [part_chunk x n_parts]
  long long part_chunk;
  // gmm_rows_kernel's form of the model (row-blocks of 32 Gaussians, see there); its own parts (a pdf of more than 32 Gaussians)
  bool rows = false;
  int n_parts2 = 0;
  struct RowsPack { DevBuf<f32x4> Mp; DevBuf<RbMeta> meta; DevBuf<int> sbs; int n_rb = 0, n_sb = 0; };
  RowsPack pack_dense, pack_local;        // the two packings of gmm_rows_kernel (see there)
  DevBuf<int> d_pdf_parts2; DevBuf<float> part_scratch2; long long part_chunk2 = 0;
  size_t lds_bytes2 = 0;
  int lse_mode = VBH_LSE_FAST;            // vbh_gmm_set_lse_mode
  // plain row-major copies for the kernels that walk single pdfs (fMLLR statistics, vbh_fmllr.hip)
  DevBuf<float> d_miv, d_iv, d_gconsts;   // [G x D], [G x D], [G]
  DevBuf<int> d_pdf_offs;                 // [P+1]
  // work buffers of vbh_gmm_loglikes_pairs, kept across calls (they only grow): a call is one pinned-staged upload of the pair lists,
  // the kernel, one download -- no allocation once the largest job has been seen
  DevBuf<long long> pr_off; DevBuf<int> pr_pdf; DevBuf<float> pr_out;
  char *pr_pin = nullptr; size_t pr_pin_bytes = 0;
  std::mutex pr_mu;
  ~vbh_gmm() { if (pr_pin) (void)hipHostFree(pr_pin); }
};

void vbh_gmm_view_get(const vbh_gmm *g, vbh_gmm_view *v) {
  v->D = g->D; v->P = g->P; v->G = g->G;
  v->pdf_offs = g->d_pdf_offs.p; v->gconsts = g->d_gconsts.p; v->miv = g->d_miv.p; v->iv = g->d_iv.p;
  v->ctx = g->ctx;
}

extern "C" {

int vbh_gmm_create(vbh_ctx *ctx, int32_t dim, int32_t num_pdfs, const int32_t *h_pdf_offsets,
                   const float *h_weights, const float *h_miv, const float *h_iv, vbh_gmm **out) {
  VBH_CHECK(ctx && h_pdf_offsets && h_weights && h_miv && h_iv && out, "vbh_gmm_create: NULL argument");
  VBH_CHECK(dim >= 1 && dim <= 63, "vbh_gmm_create: dim %d out of range [1,63]", dim);
  VBH_CHECK(num_pdfs >= 1 && num_pdfs < (1 << 24), "vbh_gmm_create: num_pdfs %d out of range [1, 2^24)", num_pdfs);
  VBH_CHECK(h_pdf_offsets[0] == 0, "vbh_gmm_create: pdf_offsets[0] != 0");
  if (vbh_use_device(ctx)) return -1;
  const int D = dim, P = num_pdfs, G = h_pdf_offsets[P];
  for (int p = 0; p < P; p++) {
    const int n = h_pdf_offsets[p + 1] - h_pdf_offsets[p];
    VBH_CHECK(n >= 1, "vbh_gmm_create: pdf %d has %d Gaussians", p, n);
  }
  vbh_gmm *g = new vbh_gmm();
  g->ctx = ctx; g->D = D; g->P = P; g->G = G;
  g->offs.assign(h_pdf_offsets, h_pdf_offsets + P + 1);
  // parts: a pdf of n Gaussians is ceil(n/TN) column groups of near-equal size
  std::vector<int> pdf_parts(P + 1, 0);
  g->part_offs.assign(1, 0);
  for (int p = 0; p < P; p++) {
    const int n = h_pdf_offsets[p + 1] - h_pdf_offsets[p], k = (n + TN - 1) / TN;
    for (int i = 1; i <= k; i++) g->part_offs.push_back(h_pdf_offsets[p] + (int)((long long)n * i / k));
    pdf_parts[p + 1] = pdf_parts[p] + k;
  }
  g->n_parts = pdf_parts[P];
  const int *poffs = g->part_offs.data();
  const int NP = g->n_parts;
  // ---- ComputeGconsts (diag-gmm.cc:114-152): float accumulator, double right-hand sides
  g->gconsts.resize(G);
  const float offset = (float)(-0.5 * 1.8378770664093454835606594728112 * D);
  for (int i = 0; i < G; i++) {
    if (!(h_weights[i] >= 0.0f)) { vbh_set_error("vbh_gmm_create: negative weight at Gaussian %d", i); delete g; return -1; }
    float gc = logf(h_weights[i]) + offset;
    for (int d = 0; d < D; d++) {
      const float iv = h_iv[(size_t)i * D + d], miv = h_miv[(size_t)i * D + d];
      gc = (float)(gc + (0.5 * logf(iv) - 0.5 * miv * miv / iv));
    }
    if (gc != gc) { vbh_set_error("vbh_gmm_create: NaN gconst at Gaussian %d", i); delete g; return -1; }
    if (std::isinf(gc) && gc > 0) gc = -gc;
    g->gconsts[i] = gc;
  }
  // ---- tiles: whole pdfs, greedily filled to TN columns
  const int K = 2 * D + 1;
  g->Kpad = (K + 7) & ~7;
  std::vector<GmmTile> tiles;
  for (int p = 0; p < NP;) {
    GmmTile t{poffs[p], 0, p, 0};
    while (p < NP && t.ng + (poffs[p + 1] - poffs[p]) <= TN) {
      t.ng += poffs[p + 1] - poffs[p];
      t.np++; p++;
    }
    tiles.push_back(t);
  }
  g->n_tiles = (int)tiles.size();
  tiles.push_back(GmmTile{0, 0, 0, 0});     // spare: the kernel reads one descriptor ahead
  std::vector<float> Bp((size_t)(g->n_tiles + 1) * g->Kpad * TN, 0.0f);   // + one spare tile (prefetch overrun)
  for (int j = 0; j < g->n_tiles; j++) {
    float *B = Bp.data() + (size_t)j * g->Kpad * TN;
    for (int c = 0; c < tiles[j].ng; c++) {
      const int gi = tiles[j].g0 + c;
      for (int d = 0; d < D; d++) {
        B[(size_t)d * TN + c] = h_miv[(size_t)gi * D + d];
        B[(size_t)(D + d) * TN + c] = -0.5f * h_iv[(size_t)gi * D + d];
      }
      B[(size_t)(2 * D) * TN + c] = g->gconsts[gi];
    }
  }
  hipStream_t s = ctx->stream;
  if (g->Bp.upload(Bp.data(), Bp.size(), s) || g->tiles.upload(tiles.data(), tiles.size(), s) ||
      g->d_offs.upload(g->part_offs.data(), g->part_offs.size(), s)) { delete g; return -1; }
  if (g->d_miv.upload(h_miv, (size_t)G * D, s) || g->d_iv.upload(h_iv, (size_t)G * D, s) || g->d_gconsts.upload(g->gconsts.data(), G, s) ||
      g->d_pdf_offs.upload(g->offs.data(), g->offs.size(), s)) { delete g; return -1; }
  g->part_chunk = 0;
  if (NP > P) {
    // scratch for the parts of a chunk of frames (<= 256 MB), a multiple of the frame tile
    long long chunk = (256LL << 20) / (4LL * NP);
    chunk = chunk / TM * TM;
    if (chunk < TM) chunk = TM;
    if (chunk > 65536) chunk = 65536;
    g->part_chunk = chunk;
    if (g->d_pdf_parts.upload(pdf_parts.data(), pdf_parts.size(), s) ||
        g->part_scratch.alloc((size_t)chunk * NP)) { delete g; return -1; }
  }
  // ---- gmm_rows_kernel's two packings of the model
  g->rows = !getenv("VBH_GMM_V1");
  if (g->rows) {
    const int NC = g->Kpad / 8;
    // rows (row_gauss[i] = Gaussian of packed row i, -1 = gap row) -> the MFMA A-operand strips, with the block table
    auto upload_pack = [&](vbh_gmm::RowsPack &pk, const std::vector<int> &row_gauss, const std::vector<RbMeta> &meta, const std::vector<int> &sbs) -> int {
      pk.n_rb = (int)meta.size();
      pk.n_sb = (int)sbs.size() / 2;
      std::vector<f32x4> Mp((size_t)pk.n_rb * NC * 64);
      for (int rb = 0; rb < pk.n_rb; rb++)
        for (int c = 0; c < NC; c++)
          for (int lane = 0; lane < 64; lane++) {
            const int gi = row_gauss[(size_t)rb * 32 + (lane & 31)];
            f32x4 v;
            for (int j = 0; j < 4; j++) {
              const int k = 2 * (4 * c + j) + (lane >> 5);
              float x = 0.0f;
              if (gi < 0) x = k == 2 * D ? -INFINITY : 0.0f;
              else if (k < D) x = h_miv[(size_t)gi * D + k];
              else if (k < 2 * D) x = -0.5f * h_iv[(size_t)gi * D + (k - D)];
              else if (k == 2 * D) x = g->gconsts[gi];
              v[j] = x;
            }
            Mp[((size_t)rb * NC + c) * 64 + lane] = v;
          }
      return pk.Mp.upload(Mp.data(), Mp.size(), s) || pk.meta.upload(meta.data(), meta.size(), s) || pk.sbs.upload(sbs.data(), sbs.size(), s);
    };
    {
      // dense: the Gaussians in order, cut into ~G/256 superblocks (a multiple of 4) at pdfs whose index is a multiple of 16 (the output
      // runs of a superblock then start on a 64-byte boundary of the row); of the multiples of 16 near a superblock's share of the rows
      // the cut is the one that leaves the fewest unused rows in its last row-block; a row-block in which 16 pdfs end already takes no
      // further pdf end (the staging area's row): gap rows to its end.  The superblocks are then dealt to the four waves longest first,
      // each to the wave with the least so far, so that the waves of a workgroup get the same number of row-blocks (table entry i
      // belongs to wave i % 4).
      std::vector<RbMeta> meta;
      std::vector<int> row_gauss;
      int want_sb = 4 * (int)((G + 4 * 256 / 2) / (4 * 256));
      if (want_sb < 4) want_sb = 4;
      if (want_sb > P) want_sb = P;
      // (never at a pdf that is no multiple of 4: the staging row, LDP, holds 15 waiting pdfs + a block's 16 only if a run's first pdf is)
      int step = P >= 32 * want_sb ? 16 : P >= 16 * want_sb ? 8 : 4;
      if (const char *e = getenv("VBH_GMM_CUT_STEP")) step = std::max(1, atoi(e));
      std::vector<int> cut(1, 0);            // first pdf of every superblock
      for (int i = 1; i < want_sb; i++) {
        const long long target = (long long)G * i / want_sb;
        const int pt = (int)(std::lower_bound(h_pdf_offsets, h_pdf_offsets + P + 1, (int)target) - h_pdf_offsets);
        const int base = (pt + step / 2) / step * step;
        int best = -1, best_pad = 0; long long best_dist = 0;
        for (int k = -2; k <= 2; k++) {
          const int c = base + k * step;
          if (c <= cut.back() || c >= P) continue;
          const int rows = h_pdf_offsets[c] - h_pdf_offsets[cut.back()], pad = (32 - rows % 32) % 32;
          const long long dist = std::llabs((long long)h_pdf_offsets[c] - target);
          if (best < 0 || pad < best_pad || (pad == best_pad && dist < best_dist)) { best = c; best_pad = pad; best_dist = dist; }
        }
        if (best > 0) cut.push_back(best);
      }
      cut.push_back(P);
      const int n_sb = (int)cut.size() - 1;
      std::vector<int> sb_first(n_sb), sb_end(n_sb);
      std::vector<int> ends;             // pdf ends per row-block so far
      std::vector<unsigned> endmask;
      std::vector<int> first_end;
      for (int i = 0; i < n_sb; i++) {
        sb_first[i] = (int)(row_gauss.size() / 32);
        for (int p = cut[i]; p < cut[i + 1]; p++) {
          const int n = h_pdf_offsets[p + 1] - h_pdf_offsets[p];
          const size_t last = row_gauss.size() + n - 1;
          if (last / 32 < ends.size() && ends[last / 32] >= 16) while (row_gauss.size() % 32) row_gauss.push_back(-1);
          for (int gi = h_pdf_offsets[p]; gi < h_pdf_offsets[p + 1]; gi++) row_gauss.push_back(gi);
          const size_t row = row_gauss.size() - 1, b = row / 32;
          if (ends.size() <= b) { ends.resize(b + 1, 0); endmask.resize(b + 1, 0u); first_end.resize(b + 1, 0); }
          if (ends[b] == 0) first_end[b] = p;
          ends[b]++;
          endmask[b] |= 1u << (row % 32);
        }
        while (row_gauss.size() % 32) row_gauss.push_back(-1);
        sb_end[i] = (int)(row_gauss.size() / 32);
      }
      const int n_rb = (int)(row_gauss.size() / 32);
      ends.resize(n_rb, 0); endmask.resize(n_rb, 0u); first_end.resize(n_rb, 0);
      for (int b = 0; b < n_rb; b++) meta.push_back(RbMeta{endmask[b], first_end[b], ends[b], 0});
      std::vector<int> order(n_sb), sbs(2 * (size_t)n_sb);
      for (int i = 0; i < n_sb; i++) order[i] = i;
      if (n_sb % 4 == 0) {
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return sb_end[a] - sb_first[a] > sb_end[b] - sb_first[b]; });
        std::vector<int> dealt(n_sb);
        int load[4] = {0, 0, 0, 0}, cnt[4] = {0, 0, 0, 0};
        for (int i = 0; i < n_sb; i++) {          // longest first, each to the wave with the least so far (of those not yet full)
          int w = -1;
          for (int k = 0; k < 4; k++) if (cnt[k] < n_sb / 4 && (w < 0 || load[k] < load[w])) w = k;
          dealt[4 * cnt[w] + w] = order[i];
          cnt[w]++; load[w] += sb_end[order[i]] - sb_first[order[i]];
        }
        order = dealt;
      }
      for (int i = 0; i < n_sb; i++) { sbs[2 * i] = sb_first[order[i]]; sbs[2 * i + 1] = sb_end[order[i]]; }
      if (getenv("VBH_DEBUG_GMM_PACK")) {
        int per_wave[4] = {0, 0, 0, 0};
        for (int i = 0; i < n_sb; i++) per_wave[i % 4] += sbs[2 * i + 1] - sbs[2 * i];
        fprintf(stderr, "vbh_gmm dense packing: %d Gaussians in %d row-blocks (%.1f %% unused rows), %d superblocks, row-blocks per wave %d %d %d %d\n",
                G, n_rb, 100.0 * (32.0 * n_rb - G) / (32.0 * n_rb), n_sb, per_wave[0], per_wave[1], per_wave[2], per_wave[3]);
      }
      if (upload_pack(g->pack_dense, row_gauss, meta, sbs)) { delete g; return -1; }
    }
    // block-local: parts of <= 32 Gaussians, row-blocks of whole parts (<= 16 of them)
    std::vector<int> part2(1, 0), pdf_parts2(P + 1, 0);
    for (int p = 0; p < P; p++) {
      const int n = h_pdf_offsets[p + 1] - h_pdf_offsets[p], k = (n + 31) / 32;
      for (int i = 1; i <= k; i++) part2.push_back(h_pdf_offsets[p] + (int)((long long)n * i / k));
      pdf_parts2[p + 1] = pdf_parts2[p] + k;
    }
    const int NP2 = pdf_parts2[P];
    g->n_parts2 = NP2;
    {
      std::vector<RbMeta> meta;
      std::vector<int> row_gauss, sbs;
      for (int p = 0; p < NP2;) {
        RbMeta m{0u, p, 0, 0};
        int rows_used = 0;
        while (p < NP2 && m.np < 16 && rows_used + (part2[p + 1] - part2[p]) <= 32) {
          m.mask |= 1u << rows_used;
          for (int gi = part2[p]; gi < part2[p + 1]; gi++) row_gauss.push_back(gi);
          rows_used += part2[p + 1] - part2[p];
          m.np++; p++;
        }
        for (; rows_used < 32; rows_used++) row_gauss.push_back(-1);
        meta.push_back(m);
      }
      for (int i = 0; i < (int)meta.size(); i++) { sbs.push_back(i); sbs.push_back(i + 1); }
      if (upload_pack(g->pack_local, row_gauss, meta, sbs)) { delete g; return -1; }
    }
    if (NP2 > P) {
      long long chunk = (256LL << 20) / (4LL * NP2);
      chunk = chunk / TM * TM;
      if (chunk < TM) chunk = TM;
      if (chunk > 65536) chunk = 65536;
      g->part_chunk2 = chunk;
      if (g->d_pdf_parts2.upload(pdf_parts2.data(), pdf_parts2.size(), s) || g->part_scratch2.alloc((size_t)chunk * NP2)) { delete g; return -1; }
    }
    g->lds_bytes2 = sizeof(float) * ((size_t)g->Kpad * LDA + ((4 - (g->Kpad * LDA) % 4) % 4) + 4 * (size_t)TM * LDP);
    if (const char *e = getenv("VBH_GMM_LDS_PAD")) g->lds_bytes2 += (size_t)atoi(e) * 1024;       // diagnostic: keeps the kernel off CUs whose LDS is mostly taken
  }
  if (const char *e = getenv("VBH_GMM_LSE")) {
    if (g->rows && !strcmp(e, "strict")) g->lse_mode = VBH_LSE_STRICT;
    else if (g->rows && !strcmp(e, "twopass")) g->lse_mode = VBH_LSE_TWOPASS;
  }
  VBH_HIP(hipStreamSynchronize(s));
  const size_t a_floats = (size_t)g->Kpad * LDA + ((4 - (g->Kpad * LDA) % 4) % 4);
  const size_t bc_floats = std::max((size_t)g->Kpad * TN, (size_t)TM * LDC);
  g->lds_bytes = sizeof(float) * (a_floats + bc_floats) + sizeof(int) * (TN + 1);
  *out = g;
  return 0;
}

void vbh_gmm_destroy(vbh_gmm *g) {
  if (!g) return;
  (void)hipSetDevice(g->ctx->device);
  (void)hipStreamSynchronize(g->ctx->stream);
  delete g;
}
int vbh_gmm_set_lse_mode(vbh_gmm *g, int32_t mode) {
  VBH_CHECK(g, "vbh_gmm_set_lse_mode: NULL model");
  VBH_CHECK(mode == VBH_LSE_FAST || mode == VBH_LSE_STRICT || mode == VBH_LSE_TWOPASS, "vbh_gmm_set_lse_mode: unknown mode %d", mode);
  VBH_CHECK(g->rows || mode == VBH_LSE_FAST, "vbh_gmm_set_lse_mode: the tile kernel (VBH_GMM_V1) has the default mode only");
  g->lse_mode = mode;
  return 0;
}
int32_t vbh_gmm_num_pdfs(const vbh_gmm *g) { return g->P; }
int32_t vbh_gmm_dim(const vbh_gmm *g) { return g->D; }
int vbh_gmm_get_gconsts(const vbh_gmm *g, float *h) {
  memcpy(h, g->gconsts.data(), sizeof(float) * g->G);
  return 0;
}

int vbh_gmm_loglikes_pairs(vbh_gmm *g, const float *d_feats, int64_t T, const int64_t *h_pair_offsets, const int32_t *h_pair_pdf, float *h_out) {
  VBH_CHECK(g && h_pair_offsets, "vbh_gmm_loglikes_pairs: NULL argument");
  if (T == 0) return 0;
  VBH_CHECK(T > 0 && d_feats, "vbh_gmm_loglikes_pairs: bad arguments");
  const long long n = h_pair_offsets[T];
  if (n == 0) return 0;
  VBH_CHECK(h_pair_pdf && h_out, "vbh_gmm_loglikes_pairs: NULL argument");
  for (long long q = 0; q < n; q++) VBH_CHECK(h_pair_pdf[q] >= 0 && h_pair_pdf[q] < g->P, "vbh_gmm_loglikes_pairs: pdf-id out of range");
  if (vbh_use_device(g->ctx)) return -1;
  hipStream_t st = g->ctx->stream;
  static_assert(sizeof(long long) == sizeof(int64_t), "int64");
  std::lock_guard<std::mutex> lk(g->pr_mu);
  // device buffers and the pinned staging area grow to the largest call seen and stay
  if (g->pr_off.n < (size_t)T + 1 && g->pr_off.alloc(((size_t)T + 1) * 5 / 4 + 16)) return -1;
  if (g->pr_pdf.n < (size_t)n && g->pr_pdf.alloc((size_t)n * 5 / 4 + 16)) return -1;
  if (g->pr_out.n < (size_t)n && g->pr_out.alloc((size_t)n * 5 / 4 + 16)) return -1;
  const size_t off_b = sizeof(long long) * ((size_t)T + 1), pdf_b = sizeof(int) * (size_t)n, out_b = sizeof(float) * (size_t)n;
  const size_t need = off_b + pdf_b + out_b;
  if (g->pr_pin_bytes < need) {
    if (g->pr_pin) (void)hipHostFree(g->pr_pin);
    g->pr_pin = nullptr; g->pr_pin_bytes = 0;
    VBH_HIP(hipHostMalloc(reinterpret_cast<void **>(&g->pr_pin), need * 5 / 4 + 4096, hipHostMallocDefault));
    g->pr_pin_bytes = need * 5 / 4 + 4096;
  }
  memcpy(g->pr_pin, h_pair_offsets, off_b);
  memcpy(g->pr_pin + off_b, h_pair_pdf, pdf_b);
  VBH_HIP(hipMemcpyAsync(g->pr_off.p, g->pr_pin, off_b, hipMemcpyHostToDevice, st));
  VBH_HIP(hipMemcpyAsync(g->pr_pdf.p, g->pr_pin + off_b, pdf_b, hipMemcpyHostToDevice, st));
  const long long blocks = (T + 3) / 4;
  VBH_CHECK(blocks < 2147483647LL, "vbh_gmm_loglikes_pairs: T too large");
  hipLaunchKernelGGL(gmm_loglikes_pairs_kernel, dim3((unsigned)blocks), dim3(256), 0, st, d_feats, (long long)T, g->D, g->pr_off.p, g->pr_pdf.p, g->d_pdf_offs.p,
                     g->d_gconsts.p, g->d_miv.p, g->d_iv.p, g->pr_out.p);
  VBH_HIP(hipGetLastError());
  VBH_HIP(hipMemcpyAsync(g->pr_pin + off_b + pdf_b, g->pr_out.p, out_b, hipMemcpyDeviceToHost, st));
  VBH_HIP(hipStreamSynchronize(st));      // (the scores are the call's result: the one synchronisation it needs)
  memcpy(h_out, g->pr_pin + off_b + pdf_b, out_b);
  return 0;
}

int vbh_gmm_work(const vbh_gmm *g, int64_t T, double *bytes, double *flops) {
  const double G = g->G, D = g->D, P = g->P, t = (double)T;
  if (bytes) *bytes = 4.0 * (2 * G * D + G) + 4.0 * t * D + 4.0 * t * P;
  if (flops) *flops = t * G * (4 * D + 2) + t * G + t * P;
  return 0;
}

static int gmm_launch_tiles(vbh_gmm *g, const float *d_feats, long long T, int n_cols, float *d_out);
static int gmm_launch_rows(vbh_gmm *g, const float *d_feats, long long T, int n_cols, float *d_out, int ld);

int vbh_gmm_loglikes(vbh_gmm *g, const float *d_feats, int64_t T, float *d_loglikes) {
  VBH_CHECK(g, "vbh_gmm_loglikes: NULL model");
  return vbh_gmm_loglikes_ld(g, d_feats, T, d_loglikes, g->P);
}

int32_t vbh_gmm_padded_pdfs(const vbh_gmm *g) { return (g->P + 31) & ~31; }

int vbh_gmm_loglikes_ld(vbh_gmm *g, const float *d_feats, int64_t T, float *d_loglikes, int32_t ld) {
  VBH_CHECK(g, "vbh_gmm_loglikes: NULL model");
  VBH_CHECK(ld >= g->P && ld < (1 << 24), "vbh_gmm_loglikes: row length %d for %d pdfs", ld, g->P);
  if (T == 0) return 0;
  VBH_CHECK(T > 0 && d_feats && d_loglikes, "vbh_gmm_loglikes: bad arguments");
  if (vbh_use_device(g->ctx)) return -1;
  if (g->rows) {
    if (g->lse_mode == VBH_LSE_FAST || g->n_parts2 == g->P) return gmm_launch_rows(g, d_feats, T, g->P, d_loglikes, ld);
    for (long long t0 = 0; t0 < T; t0 += g->part_chunk2) {
      const long long n = std::min<long long>(g->part_chunk2, T - t0);
      if (gmm_launch_rows(g, d_feats + t0 * g->D, n, g->n_parts2, g->part_scratch2.p, g->n_parts2)) return -1;
      const long long blocks = (n * g->P + 255) / 256;
      VBH_CHECK(blocks < 2147483647LL, "vbh_gmm_loglikes: chunk too large");
      hipLaunchKernelGGL(gmm_fold_parts_kernel, dim3((unsigned)blocks), dim3(256), 0, g->ctx->stream,
                         g->part_scratch2.p, n, g->n_parts2, g->P, g->d_pdf_parts2.p, d_loglikes + t0 * ld, (int)ld, (int)(g->lse_mode == VBH_LSE_STRICT));
      VBH_HIP(hipGetLastError());
    }
    return 0;
  }
  VBH_CHECK(ld == g->P, "vbh_gmm_loglikes: the tile kernel (VBH_GMM_V1) writes dense rows only");
  if (g->n_parts == g->P) return gmm_launch_tiles(g, d_feats, T, g->P, d_loglikes);
  // some pdf spans several column groups: score the parts of a chunk of frames, fold them
  for (long long t0 = 0; t0 < T; t0 += g->part_chunk) {
    const long long n = std::min<long long>(g->part_chunk, T - t0);
    if (gmm_launch_tiles(g, d_feats + t0 * g->D, n, g->n_parts, g->part_scratch.p)) return -1;
    const long long blocks = (n * g->P + 255) / 256;
    VBH_CHECK(blocks < 2147483647LL, "vbh_gmm_loglikes: chunk too large");
    hipLaunchKernelGGL(gmm_fold_parts_kernel, dim3((unsigned)blocks), dim3(256), 0, g->ctx->stream,
                       g->part_scratch.p, n, g->n_parts, g->P, g->d_pdf_parts.p, d_loglikes + t0 * g->P, g->P, 0);
    VBH_HIP(hipGetLastError());
  }
  return 0;
}

}  // extern "C"

static int gmm_launch_tiles(vbh_gmm *g, const float *d_feats, long long T, int n_cols, float *d_out) {
  const long long ftiles = (T + TM - 1) / TM;
  VBH_CHECK(ftiles < 2147483647LL, "vbh_gmm_loglikes: T too large");
  // split the Gaussian tiles over blockIdx.y only when there are too few frame tiles to fill the chip
  int groups = 1;
  const long long want = 4LL * g->ctx->num_cus;
  if (ftiles < want) {
    groups = (int)((want + ftiles - 1) / ftiles);
    if (groups > g->n_tiles) groups = g->n_tiles;
  }
  const int per = (g->n_tiles + groups - 1) / groups;
  groups = (g->n_tiles + per - 1) / per;
#define VBH_GMM_LAUNCH(N)                                                                              \
  case N:                                                                                              \
    VBH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(gmm_loglikes_kernel<N>),               \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)g->lds_bytes));       \
    hipLaunchKernelGGL(gmm_loglikes_kernel<N>, dim3((unsigned)ftiles, groups), dim3(256), g->lds_bytes, \
                       g->ctx->stream, d_feats, (long long)T, g->D, n_cols, g->Bp.p, g->tiles.p,         \
                       g->n_tiles, per, g->d_offs.p, d_out);                                           \
    break;
  switch (g->Kpad / 8) {
    VBH_GMM_LAUNCH(1) VBH_GMM_LAUNCH(2) VBH_GMM_LAUNCH(3) VBH_GMM_LAUNCH(4) VBH_GMM_LAUNCH(5) VBH_GMM_LAUNCH(6)
    VBH_GMM_LAUNCH(7) VBH_GMM_LAUNCH(8) VBH_GMM_LAUNCH(9) VBH_GMM_LAUNCH(10) VBH_GMM_LAUNCH(11) VBH_GMM_LAUNCH(12)
    VBH_GMM_LAUNCH(13) VBH_GMM_LAUNCH(14) VBH_GMM_LAUNCH(15) VBH_GMM_LAUNCH(16)
    default: vbh_set_error("vbh_gmm_loglikes: unsupported K %d", g->Kpad); return -1;
  }
#undef VBH_GMM_LAUNCH
  VBH_HIP(hipGetLastError());
  return 0;
}

static int gmm_launch_rows(vbh_gmm *g, const float *d_feats, long long T, int n_cols, float *d_out, int ld) {
  const long long ftiles = (T + TM - 1) / TM;
  VBH_CHECK(ftiles < 2147483647LL, "vbh_gmm_loglikes: T too large");
  const vbh_gmm::RowsPack &pk = g->lse_mode == VBH_LSE_FAST ? g->pack_dense : g->pack_local;
  // split the superblocks over blockIdx.y only when there are too few frame tiles to fill the chip
  int groups = 1;
  const long long want = 6LL * g->ctx->num_cus;
  if (ftiles < want) {
    groups = (int)((want + ftiles - 1) / ftiles);
    if (groups > (pk.n_sb + 3) / 4) groups = (pk.n_sb + 3) / 4;
  }
  int per = (pk.n_sb + groups - 1) / groups;
  per = (per + 3) & ~3;
  groups = (pk.n_sb + per - 1) / per;
#define VBH_GMM_LAUNCH_M(N, M)                                                                         \
    {                                                                                                  \
      VBH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(gmm_rows_kernel<N, M>),              \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)g->lds_bytes2));    \
      hipLaunchKernelGGL((gmm_rows_kernel<N, M>), dim3((unsigned)ftiles, groups), dim3(256), g->lds_bytes2, \
                         g->ctx->stream, d_feats, (long long)T, g->D, n_cols, pk.Mp.p, pk.meta.p,      \
                         pk.sbs.p, pk.n_sb, per, d_out, ld);                                            \
    }
#define VBH_GMM_LAUNCH(N)                                                                              \
  case N:                                                                                              \
    if (g->lse_mode == VBH_LSE_STRICT) VBH_GMM_LAUNCH_M(N, VBH_LSE_STRICT)                             \
    else if (g->lse_mode == VBH_LSE_TWOPASS) VBH_GMM_LAUNCH_M(N, VBH_LSE_TWOPASS)                      \
    else VBH_GMM_LAUNCH_M(N, VBH_LSE_FAST)                                                             \
    break;
  switch (g->Kpad / 8) {
    VBH_GMM_LAUNCH(1) VBH_GMM_LAUNCH(2) VBH_GMM_LAUNCH(3) VBH_GMM_LAUNCH(4) VBH_GMM_LAUNCH(5) VBH_GMM_LAUNCH(6)
    VBH_GMM_LAUNCH(7) VBH_GMM_LAUNCH(8) VBH_GMM_LAUNCH(9) VBH_GMM_LAUNCH(10) VBH_GMM_LAUNCH(11) VBH_GMM_LAUNCH(12)
    VBH_GMM_LAUNCH(13) VBH_GMM_LAUNCH(14) VBH_GMM_LAUNCH(15) VBH_GMM_LAUNCH(16)
    default: vbh_set_error("vbh_gmm_loglikes: unsupported K %d", g->Kpad); return -1;
  }
#undef VBH_GMM_LAUNCH
#undef VBH_GMM_LAUNCH_M
  VBH_HIP(hipGetLastError());
  return 0;
}
