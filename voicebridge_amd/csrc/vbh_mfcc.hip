// vbh_mfcc.hip -- frame-batched MFCC on gfx950: one 64-lane wavefront per frame, four frames
// per 256-thread workgroup, window / twiddle / mel / DCT tables staged once per workgroup in
// LDS, the N-point real FFT done as an N/2-point complex Stockham FFT in per-wave LDS
// ping-pong buffers.
//
// Replaces (paths relative to /root/reference/kaldi-master/src/):
//   feat/feature-common-inl.h:75-98   OfflineFeatureTpl<MfccComputer>::Compute (frame loop)
//   feat/feature-window.cc:133-220    ExtractWindow + ProcessWindow
//   matrix/srfft.cc:355-425           SplitRadixRealFft::Compute (any exact real FFT; only
//                                     the power spectrum is consumed)
//   feat/feature-functions.cc:29-51   ComputePowerSpectrum
//   feat/mel-computations.cc:228-253  MelBanks::Compute
//   feat/feature-mfcc.cc:28-80        MfccComputer::Compute (floor/log/DCT/lifter/energy)
// Tables follow feature-window.cc:109-131, mel-computations.cc:33-144,255-261 and
// matrix/matrix-functions.cc:592-608 (computed on the host in the same float/double mix).
//
// Two kernels.  mfcc512_kernel is the path of the reference's own configuration (25 ms windows at
// 16 kHz -> N = 512; up to 32 mel bins): one wavefront per frame with NO workgroup barrier in the
// frame loop -- a lane keeps four complex points of the 256-point FFT in registers, the four
// radix-4 Stockham passes exchange them through a per-wave LDS buffer (in-order LDS, wave-level
// fences only), sums are DPP reductions.  mfcc_kernel is the general one (any power-of-two
// window up to 4096, radix-2).
//
// Dither (feature-window.cc:90-98, applied to the extracted window before DC removal): the
// reference adds dither * N(0,1) drawn from libc rand(), which no parallel code can replay.
// Here every window sample gets an independent standard normal from a counter-based generator
// (Philox2x32-10 keyed by a seed, counter = utterance, frame, sample pair; Box-Muller): the same
// distribution, reproducible for a given seed and independent of batching, not the reference's
// bit stream.
//
// Algorithmic traffic per frame: 160 new int16 samples in, num_ceps floats
// out = 372 B at the defaults; ~15 kflop.
#include "vbh_internal.h"

#include <cfloat>
#include <cmath>

namespace {

constexpr int kWaves = 4;           // frames in flight per workgroup
constexpr int kChunkFrames = 64;    // frames of one utterance handled by one workgroup

struct MfccParams {
  int flen, padded, log2m;          // window length, FFT size N, log2(N/2)
  int shift;
  int nbins, nceps;
  int snip_edges, remove_dc, use_energy, raw_energy, htk_compat, htk_mode, use_lifter;
  float preemph, energy_floor, log_energy_floor;
  int melw_total;                   // total packed mel weights
  float dither;                     // 0 = off
  unsigned seed_lo, seed_hi;        // dither stream
  long long utt_base;               // job-wide index of the call's first utterance (noise independent of batching)
  const long long *utt_ids;         // or: the stream index of every utterance of the call (vbh_mfcc_set_dither_ids), NULL = utt_base + position
};

struct vbh_mfcc_tables {
  const float *window;   // [flen]
  const float *twiddle;  // [N/2] x (cos, sin) of 2*pi*k/N
  const int *mel_first;  // [nbins]
  const int *mel_len;    // [nbins]
  const int *mel_off;    // [nbins] offset into mel_w
  const float *mel_w;    // packed weights
  const float *dct;      // [nceps x nbins]
  const float *lifter;   // [nceps]
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <bool kInt16>
__device__ __forceinline__ float load_sample(const void *wave, long long i) {
  if (kInt16) return (float)reinterpret_cast<const short *>(wave)[i];
  return reinterpret_cast<const float *>(wave)[i];
}


// ---- dither noise: two independent N(0,1) for window samples (2p, 2p+1) of frame f of utterance u
__device__ __forceinline__ void dither_pair(const MfccParams &p, long long utt, int frame, int pair, float &n0, float &n1) {
  unsigned c0 = (unsigned)frame, c1 = ((unsigned)pair & 0xFFFu) | ((unsigned)utt << 12);
  unsigned key = p.seed_lo ^ (unsigned)((unsigned long long)utt >> 20) * 0x9E3779B9u ^ p.seed_hi * 0x85EBCA6Bu;
#pragma unroll
  for (int r = 0; r < 10; r++) {      // Philox2x32-10
    const unsigned hi = __umulhi(0xD256D193u, c0), lo = 0xD256D193u * c0;
    c0 = hi ^ key ^ c1; c1 = lo;
    key += 0x9E3779B9u;
  }
  const float u1 = (float)((c0 >> 8) + 1u) * (1.0f / 16777216.0f);      // (0, 1]
  const float u2 = (float)(c1 >> 8) * (1.0f / 16777216.0f);             // [0, 1)
  const float r = sqrtf(-2.0f * __logf(u1));
  float sn, cs;
  __sincosf(6.283185307179586f * u2, &sn, &cs);
  n0 = r * cs; n1 = r * sn;
}

// sum over the wave on the VALU (DPP row shifts + row broadcasts), result on every lane
__device__ __forceinline__ float wave_sum_dpp(float v) {
#define VBH_DPPF(x, ctrl, rm, bc) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, rm, 0xf, bc))
  v += VBH_DPPF(v, 0x111, 0xf, true);
  v += VBH_DPPF(v, 0x112, 0xf, true);
  v += VBH_DPPF(v, 0x114, 0xf, true);
  v += VBH_DPPF(v, 0x118, 0xf, true);
  v += VBH_DPPF(v, 0x142, 0xa, false);
  v += VBH_DPPF(v, 0x143, 0xc, false);
#undef VBH_DPPF
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ void wave_lds_fence() {
  // the LDS queue of a wave is in order; this only keeps the compiler from moving LDS accesses of
  // different lanes' data across each other
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
// padded index of the per-wave FFT buffer (one extra slot per 16: spreads the stride-4 writes over the banks)
__device__ __forceinline__ int fpad(int i) { return i + (i >> 4); }

constexpr int kFastN = 512, kFastM = 256;
constexpr int kFastBuf = kFastM + (kFastM >> 4);    // float2 slots

// dynamic LDS layout (floats): t256[2*256] | per wave: buf[2*kFastBuf] mel[32]
// Per-lane constants live in registers for the whole kernel: the window values of the lane's eight samples, the split
// twiddles and mel weights of its five spectrum bins, its eight DCT coefficients.
//   mel: a spectrum bin lies under at most two (adjacent) triangular filters, so bin k carries {lower filter, weight in it,
//   weight in the next one} (binw, built at create time) and the filter energies are accumulated with LDS float adds straight
//   from the registers that hold the power spectrum -- no per-filter loop, no spectrum in LDS;
//   DCT: lane = (cepstral index c = lane & 15, quarter q = lane >> 4) sums the mel bins q, q + 4, ... and two shuffles add
//   the quarters.
struct FastTables {
  const float *t256;      // [256] x (cos, -sin)(2 pi m / 256)
  const float *t512;      // [129] x (cos, -sin)(2 pi k / 512)
  const float *binw;      // [257] x {lower filter (as float), weight there, weight in the next filter, 0}
};

template <bool kInt16, bool kDither>
__global__ __launch_bounds__(256) void mfcc512_kernel(MfccParams p, vbh_mfcc_tables t, FastTables ft, const void *wave,
                                                      const long long *wave_offsets, const long long *frame_offsets, const int *chunk_utt,
                                                      const int *chunk_f0, int n_chunks, float *feats) {
  extern __shared__ __align__(16) float lds[];
  float2 *s_t256 = reinterpret_cast<float2 *>(lds);
  float *s_wave0 = lds + 512;
  constexpr int per_wave = 2 * kFastBuf + 32;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int i = tid; i < 256; i += 256) s_t256[i] = make_float2(ft.t256[2 * i], ft.t256[2 * i + 1]);
  __syncthreads();
  float2 *buf = reinterpret_cast<float2 *>(s_wave0 + wid * per_wave);
  float *s_mel = reinterpret_cast<float *>(buf + kFastBuf);
  const int flen = p.flen;
  // ---- per-lane constants
  float2 win[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int i0 = 2 * (lane + 64 * k);
    win[k] = make_float2(i0 < flen ? t.window[i0] : 0.0f, i0 + 1 < flen ? t.window[i0 + 1] : 0.0f);
  }
  // spectrum bins of this lane: lane, lane + 64 (split index kk) and their mirrors 256 - kk; lane 0 also owns bin 128
  float2 tw5[2];
  float4 bw[5];
#pragma unroll
  for (int m = 0; m < 2; m++) {
    const int kk = lane + 64 * m;
    tw5[m] = make_float2(ft.t512[2 * kk], ft.t512[2 * kk + 1]);
    bw[2 * m] = reinterpret_cast<const float4 *>(ft.binw)[kk];
    bw[2 * m + 1] = reinterpret_cast<const float4 *>(ft.binw)[kFastM - kk];
  }
  bw[4] = reinterpret_cast<const float4 *>(ft.binw)[128];
  const float2 tw128 = make_float2(ft.t512[256], ft.t512[257]);
  const int dc = lane & 15, dq = lane >> 4;
  float dctc[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int b = dq + 4 * j;
    dctc[j] = (dc < p.nceps && b < p.nbins) ? t.dct[dc * p.nbins + b] : 0.0f;
  }
  const float lift = (p.use_lifter && dc < p.nceps) ? t.lifter[dc] : 1.0f;

  for (int chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    const int utt = chunk_utt[chunk], f0 = chunk_f0[chunk];
    const long long w0 = wave_offsets[utt], nsamp = wave_offsets[utt + 1] - w0;
    const long long row0 = frame_offsets[utt];
    const int T = (int)(frame_offsets[utt + 1] - row0);
    int f_end = f0 + kChunkFrames;
    if (f_end > T) f_end = T;
    for (int f = f0 + wid; f < f_end; f += kWaves) {
      // ---- ExtractWindow (feature-window.cc:162-220): lane l holds the sample pairs p = l + 64 k, k < 4,
      //      i.e. the complex points z[l + 64 k] of the N/2-point FFT -- the inputs of radix-4 butterfly l
      long long start;
      if (p.snip_edges) start = (long long)f * p.shift;
      else start = (long long)p.shift * f + p.shift / 2 - flen / 2;
      const bool edge = start < 0 || start + flen > nsamp;
      float xe[4], xo[4];
      if (!edge && kInt16 && (((w0 + start) & 1) == 0)) {
        const unsigned *wp = reinterpret_cast<const unsigned *>(reinterpret_cast<const short *>(wave) + (w0 + start));
        unsigned v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int i0 = 2 * (lane + 64 * k);
          v[k] = (i0 + 1 < flen) ? wp[lane + 64 * k] : 0u;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int i0 = 2 * (lane + 64 * k);
          xe[k] = (float)(short)(v[k] & 0xFFFFu); xo[k] = (float)(short)(v[k] >> 16);
          if (i0 + 1 >= flen && i0 < flen) { xe[k] = load_sample<kInt16>(wave, w0 + start + i0); xo[k] = 0.0f; }
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int i0 = 2 * (lane + 64 * k);
          float a = 0.0f, b = 0.0f;
          if (i0 < flen) {
            long long s0 = start + i0;
            while (s0 < 0 || s0 >= nsamp) s0 = (s0 < 0) ? (-s0 - 1) : (2 * nsamp - 1 - s0);
            a = load_sample<kInt16>(wave, w0 + s0);
          }
          if (i0 + 1 < flen) {
            long long s1 = start + i0 + 1;
            while (s1 < 0 || s1 >= nsamp) s1 = (s1 < 0) ? (-s1 - 1) : (2 * nsamp - 1 - s1);
            b = load_sample<kInt16>(wave, w0 + s1);
          }
          xe[k] = a; xo[k] = b;
        }
      }
      // ---- ProcessWindow (feature-window.cc:133-156)
      if (kDither) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int pr = lane + 64 * k;
          if (2 * pr < flen) {
            float n0, n1;
            dither_pair(p, (p.utt_ids ? p.utt_ids[utt] : p.utt_base + utt), f, pr, n0, n1);
            xe[k] += p.dither * n0;
            if (2 * pr + 1 < flen) xo[k] += p.dither * n1;
          }
        }
      }
      if (p.remove_dc) {
        const float mean = wave_sum_dpp((xe[0] + xo[0]) + (xe[1] + xo[1]) + (xe[2] + xo[2]) + (xe[3] + xo[3])) / (float)flen;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int i0 = 2 * (lane + 64 * k);
          if (i0 < flen) xe[k] -= mean;
          if (i0 + 1 < flen) xo[k] -= mean;
        }
      }
      float log_energy = 0.f;
      if (p.use_energy && p.raw_energy) {
        float e = 0.f;
#pragma unroll
        for (int k = 0; k < 4; k++) e += xe[k] * xe[k] + xo[k] * xo[k];
        log_energy = logf(fmaxf(wave_sum_dpp(e), FLT_EPSILON));
      }
      float2 a[4];
      {
        // pre-emphasis reads the un-emphasised left neighbour; sample 0 uses itself
        float prev[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          float up = __shfl_up(xo[k], 1, 64);
          if (lane == 0) up = k > 0 ? 0.0f : xe[0];
          prev[k] = up;
        }
#pragma unroll
        for (int k = 1; k < 4; k++) {
          const float last = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xo[k - 1]), 63));
          if (lane == 0) prev[k] = last;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
          float ye = xe[k], yo = xo[k];
          if (p.preemph != 0.f) { ye -= p.preemph * prev[k]; yo -= p.preemph * xe[k]; }
          a[k] = make_float2(ye * win[k].x, yo * win[k].y);
        }
      }
      if (p.use_energy && !p.raw_energy) {  // feature-mfcc.cc:39-41
        float e2 = 0.f;
#pragma unroll
        for (int k = 0; k < 4; k++) e2 += a[k].x * a[k].x + a[k].y * a[k].y;
        log_energy = logf(fmaxf(wave_sum_dpp(e2), FLT_MIN));
      }
      if (lane < 32) s_mel[lane] = 0.0f;
      // ---- 256-point complex FFT: four radix-4 Stockham passes, butterfly j = lane
#pragma unroll
      for (int s = 0; s < 4; s++) {
        const int Ns = 1 << (2 * s);
        const int k = lane & (Ns - 1);
        if (s > 0) {
          wave_lds_fence();
          const int m = k * (64 >> (2 * s));
          const float2 w1 = s_t256[m], w2 = s_t256[2 * m];
#pragma unroll
          for (int q = 0; q < 4; q++) a[q] = buf[fpad(lane + 64 * q)];
          const float2 w3 = cmul(w1, w2);
          a[1] = cmul(a[1], w1); a[2] = cmul(a[2], w2); a[3] = cmul(a[3], w3);
          wave_lds_fence();
        }
        const float2 t0 = make_float2(a[0].x + a[2].x, a[0].y + a[2].y), t1 = make_float2(a[0].x - a[2].x, a[0].y - a[2].y);
        const float2 t2 = make_float2(a[1].x + a[3].x, a[1].y + a[3].y), d = make_float2(a[1].x - a[3].x, a[1].y - a[3].y);
        const float2 t3 = make_float2(d.y, -d.x);      // -i (b1 - b3)
        const int o = ((lane - k) << 2) + k;
        buf[fpad(o)] = make_float2(t0.x + t2.x, t0.y + t2.y);
        buf[fpad(o + Ns)] = make_float2(t1.x + t3.x, t1.y + t3.y);
        buf[fpad(o + 2 * Ns)] = make_float2(t0.x - t2.x, t0.y - t2.y);
        buf[fpad(o + 3 * Ns)] = make_float2(t1.x - t3.x, t1.y - t3.y);
      }
      wave_lds_fence();
      // ---- real-FFT split + ComputePowerSpectrum (feature-functions.cc:29-51): bins kk and N/2 - kk from one pair of
      //      points, then MelBanks::Compute (mel-computations.cc:228-253) as float adds into the filters' LDS cells
      {
        float pw[5];
        const float2 z0 = buf[fpad(lane)], z0m = buf[fpad((kFastM - lane) & (kFastM - 1))];
        const float2 z1 = buf[fpad(lane + 64)], z1m = buf[fpad(kFastM - lane - 64)];
        const float2 z128 = buf[fpad(128)];
        auto split = [](float2 zk, float2 zm, float2 c, float &p_lo, float &p_hi) {
          const float er = 0.5f * (zk.x + zm.x), ei = 0.5f * (zk.y - zm.y);
          const float orr = 0.5f * (zk.y + zm.y), oi = -0.5f * (zk.x - zm.x);
          const float tr = orr * c.x - oi * c.y, ti = orr * c.y + oi * c.x;
          p_lo = (er + tr) * (er + tr) + (ei + ti) * (ei + ti);
          p_hi = (er - tr) * (er - tr) + (ei - ti) * (ei - ti);
        };
        split(z0, z0m, tw5[0], pw[0], pw[1]);
        split(z1, z1m, tw5[1], pw[2], pw[3]);
        float dummy;
        split(z128, z128, tw128, pw[4], dummy);
#pragma unroll
        for (int j = 0; j < 5; j++) {
          if (j == 4 && lane != 0) break;
          const int fl = (int)bw[j].x;
          if (bw[j].y != 0.0f) atomicAdd(&s_mel[fl], bw[j].y * pw[j]);
          if (bw[j].z != 0.0f) atomicAdd(&s_mel[fl + 1], bw[j].z * pw[j]);
        }
      }
      wave_lds_fence();
      // floor + log (feature-mfcc.cc:54-55)
      if (lane < p.nbins) {
        float e = s_mel[lane];
        if (p.htk_mode && e < 1.0f) e = 1.0f;
        s_mel[lane] = logf(fmaxf(e, FLT_EPSILON));
      }
      wave_lds_fence();
      // ---- DCT, lifter, energy / C0, htk reorder (feature-mfcc.cc:57-79)
      {
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) v = fmaf(dctc[j], s_mel[(dq + 4 * j) & 31], v);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (lane < p.nceps) {
          v *= lift;
          if (p.use_energy && lane == 0) {
            float le = log_energy;
            if (p.energy_floor > 0.f && le < p.log_energy_floor) le = p.log_energy_floor;
            v = le;
          }
          int col = lane;
          if (p.htk_compat) {
            if (lane == 0) {
              col = p.nceps - 1;
              if (!p.use_energy) v *= 1.41421356237309504880f;
            } else col = lane - 1;
          }
          feats[(row0 + f) * p.nceps + col] = v;
        }
      }
      wave_lds_fence();
    }
  }
}

// dynamic LDS layout (floats):
//   window[flen] | tw[N] | dct[nceps*nbins] | lifter[nceps] | melw[melw_total] |
//   mel meta ints [3*nbins] | per wave: bufA[N] bufB[N] pspec[N/2+2] mel[nbins(+pad)]
template <bool kInt16>
__global__ __launch_bounds__(256) void mfcc_kernel(MfccParams p, vbh_mfcc_tables t, const void *wave,
                                                   const long long *wave_offsets,  // [n_utt+1]
                                                   const long long *frame_offsets, // [n_utt+1]
                                                   const int *chunk_utt, const int *chunk_f0,
                                                   int n_chunks, float *feats) {
  extern __shared__ __align__(16) float lds[];
  const int N = p.padded, M = N >> 1;
  float *s_window = lds;
  float *s_tw = s_window + p.flen;
  float *s_dct = s_tw + N;
  float *s_lifter = s_dct + p.nceps * p.nbins;
  float *s_melw = s_lifter + p.nceps;
  int *s_melmeta = reinterpret_cast<int *>(s_melw + p.melw_total);
  float *s_wave0 = reinterpret_cast<float *>(s_melmeta + 3 * p.nbins);
  const int mel_pad = (p.nbins + 3) & ~3;
  const int per_wave = 2 * N + (M + 4) + mel_pad;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int i = tid; i < p.flen; i += 256) s_window[i] = t.window[i];
  for (int i = tid; i < N; i += 256) s_tw[i] = t.twiddle[i];
  for (int i = tid; i < p.nceps * p.nbins; i += 256) s_dct[i] = t.dct[i];
  for (int i = tid; i < p.nceps; i += 256) s_lifter[i] = t.lifter[i];
  for (int i = tid; i < p.melw_total; i += 256) s_melw[i] = t.mel_w[i];
  for (int i = tid; i < p.nbins; i += 256) {
    s_melmeta[i] = t.mel_first[i];
    s_melmeta[p.nbins + i] = t.mel_len[i];
    s_melmeta[2 * p.nbins + i] = t.mel_off[i];
  }
  __syncthreads();

  float *bufA = s_wave0 + wid * per_wave;
  float *bufB = bufA + N;
  float *s_ps = bufB + N;
  float *s_mel = s_ps + (M + 4);

  for (int chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    const int utt = chunk_utt[chunk], f0 = chunk_f0[chunk];
    const long long w0 = wave_offsets[utt], nsamp = wave_offsets[utt + 1] - w0;
    const long long row0 = frame_offsets[utt];
    const int T = (int)(frame_offsets[utt + 1] - row0);
    int f_end = f0 + kChunkFrames;
    if (f_end > T) f_end = T;
    // every wave runs the same number of iterations (barriers inside); surplus ones are masked
    for (int fb = f0; fb < f_end; fb += kWaves) {
      const int f = fb + wid;
      const bool valid = f < f_end;
      // ---- ExtractWindow (feature-window.cc:162-220)
      long long start;
      if (p.snip_edges) start = (long long)f * p.shift;
      else start = (long long)p.shift * f + p.shift / 2 - p.flen / 2;
      float part = 0.f;
      if (valid) {
        for (int i = lane; i < N; i += 64) {
          float v = 0.f;
          if (i < p.flen) {
            long long s = start + i;
            if (s < 0 || s >= nsamp) {  // reflection at the edges (snip_edges == false)
              while (s < 0 || s >= nsamp) s = (s < 0) ? (-s - 1) : (2 * nsamp - 1 - s);
            }
            v = load_sample<kInt16>(wave, w0 + s);
            if (p.dither != 0.0f) {
              float n0, n1;
              dither_pair(p, (p.utt_ids ? p.utt_ids[utt] : p.utt_base + utt), f, i >> 1, n0, n1);
              v += p.dither * ((i & 1) ? n1 : n0);
            }
            part += v;
          }
          bufA[i] = v;
        }
      }
      // ---- ProcessWindow (feature-window.cc:133-156)
      float mean = 0.f;
      if (p.remove_dc) mean = wave_sum(part) / (float)p.flen;
      float e_part = 0.f;
      if (valid) {
        for (int i = lane; i < p.flen; i += 64) {
          float v = bufA[i] - mean;   // Add(-Sum()/frame_length)
          e_part += v * v;
          bufA[i] = v;
        }
      }
      float log_energy = 0.f;
      if (p.use_energy && p.raw_energy) log_energy = logf(fmaxf(wave_sum(e_part), FLT_EPSILON));
      __syncthreads();
      // pre-emphasis reads the un-emphasised left neighbour (the reference walks back to front)
      if (valid) {
        for (int i = lane; i < p.flen; i += 64) {
          float v = bufA[i];
          float prev = bufA[i > 0 ? i - 1 : 0];
          if (p.preemph != 0.f) v -= p.preemph * prev;
          bufB[i] = v * s_window[i];
        }
        for (int i = p.flen + lane; i < N; i += 64) bufB[i] = 0.f;
      }
      __syncthreads();
      if (p.use_energy && !p.raw_energy) {  // feature-mfcc.cc:39-41
        float e2 = 0.f;
        if (valid) for (int i = lane; i < N; i += 64) e2 += bufB[i] * bufB[i];
        log_energy = logf(fmaxf(wave_sum(e2), FLT_MIN));
      }
      // ---- N/2-point complex Stockham radix-2 FFT of z[n] = x[2n] + i x[2n+1]
      float2 *src = reinterpret_cast<float2 *>(bufB), *dst = reinterpret_cast<float2 *>(bufA);
      for (int s = 0; s < p.log2m; s++) {
        const int Ns = 1 << s;
        if (valid) {
          for (int j = lane; j < (M >> 1); j += 64) {
            const int k = j & (Ns - 1);
            const float2 a = src[j], b = src[j + (M >> 1)];
            // twiddle e^{-2 pi i k / (2 Ns)} = table entry q = k * N / (2 Ns) of (cos, sin)(2 pi q / N)
            const int q = k * (N >> (s + 1));
            const float cr = s_tw[2 * q], ci = -s_tw[2 * q + 1];
            const float2 wb = make_float2(b.x * cr - b.y * ci, b.x * ci + b.y * cr);
            const int idx = ((j - k) << 1) + k;
            dst[idx] = make_float2(a.x + wb.x, a.y + wb.y);
            dst[idx + Ns] = make_float2(a.x - wb.x, a.y - wb.y);
          }
        }
        __syncthreads();
        float2 *tmp = src; src = dst; dst = tmp;
      }
      // src now holds Z[0..M)
      // ---- real-FFT split + ComputePowerSpectrum (feature-functions.cc:29-51)
      if (valid) {
        for (int k = lane; k <= M; k += 64) {
          float pw;
          if (k == 0 || k == M) {
            const float2 z0 = src[0];
            const float r = (k == 0) ? (z0.x + z0.y) : (z0.x - z0.y);
            pw = r * r;
          } else {
            const float2 zk = src[k], zm = src[M - k];
            const float er = 0.5f * (zk.x + zm.x), ei = 0.5f * (zk.y - zm.y);   // even part
            const float orr = 0.5f * (zk.y + zm.y), oi = -0.5f * (zk.x - zm.x); // odd part
            const float cr = s_tw[2 * k], ci = -s_tw[2 * k + 1];               // e^{-2 pi i k/N}
            const float xr = er + (orr * cr - oi * ci), xi = ei + (orr * ci + oi * cr);
            pw = xr * xr + xi * xi;
          }
          s_ps[k] = pw;
        }
      }
      __syncthreads();
      // ---- MelBanks::Compute + floor + log (mel-computations.cc:228-253, feature-mfcc.cc:54-55)
      if (valid) {
        for (int b = lane; b < p.nbins; b += 64) {
          const int first = s_melmeta[b], len = s_melmeta[p.nbins + b], off = s_melmeta[2 * p.nbins + b];
          float e = 0.f;
          for (int i = 0; i < len; i++) e = fmaf(s_melw[off + i], s_ps[first + i], e);
          if (p.htk_mode && e < 1.0f) e = 1.0f;
          s_mel[b] = logf(fmaxf(e, FLT_EPSILON));
        }
      }
      __syncthreads();
      // ---- DCT, lifter, energy / C0, htk reorder (feature-mfcc.cc:57-79)
      if (valid && lane < p.nceps) {
        float v = 0.f;
        for (int b = 0; b < p.nbins; b++) v = fmaf(s_dct[lane * p.nbins + b], s_mel[b], v);
        if (p.use_lifter) v *= s_lifter[lane];
        if (p.use_energy && lane == 0) {
          float le = log_energy;
          if (p.energy_floor > 0.f && le < p.log_energy_floor) le = p.log_energy_floor;
          v = le;
        }
        int col = lane;
        if (p.htk_compat) {
          if (lane == 0) {
            col = p.nceps - 1;
            if (!p.use_energy) v *= 1.41421356237309504880f;
          } else col = lane - 1;
        }
        feats[(row0 + f) * p.nceps + col] = v;
      }
      __syncthreads();
    }
  }
}

}  // namespace

struct vbh_mfcc {
  vbh_ctx *ctx;
  vbh_mfcc_opts opts;
  MfccParams p;
  DevBuf<float> window, twiddle, mel_w, dct, lifter;
  DevBuf<float> t256, t512, binw;  // fast kernel: exp(-2 pi i m / 256), m < 256; exp(-2 pi i k / 512), k <= 128; per-bin mel weights
  bool fast = false; size_t fast_lds = 0;
  DevBuf<int> mel_first, mel_len, mel_off;
  DevBuf<long long> d_wave_off, d_frame_off;
  DevBuf<int> d_chunk_utt, d_chunk_f0;
  size_t lds_bytes;
  std::vector<int64_t> dither_ids;   // vbh_mfcc_set_dither_ids: stream indices of the next call's utterances
};

// ---- option helpers (feature-window.h:92-101, feature-window.cc:41-87)
static int32_t win_shift(const vbh_mfcc_opts *o) { return (int32_t)(o->samp_freq * 0.001 * o->frame_shift_ms); }
static int32_t win_size(const vbh_mfcc_opts *o) { return (int32_t)(o->samp_freq * 0.001 * o->frame_length_ms); }
static int32_t padded_size(const vbh_mfcc_opts *o) {
  int32_t n = win_size(o);
  if (!o->round_to_power_of_two) return n;
  int32_t p = 1;
  while (p < n) p <<= 1;
  return p;
}
static float mel_scale(float f) { return 1127.0f * logf(1.0f + f / 700.0f); }
static float inv_mel_scale(float m) { return 700.0f * (expf(m / 1127.0f) - 1.0f); }

// mel-computations.cc:152-224
static float vtln_warp_freq(float vl, float vh, float low, float high, float warp, float freq) {
  if (freq < low || freq > high) return freq;
  float l = vl * std::max(1.0f, warp), h = vh * std::min(1.0f, warp);
  float scale = 1.0f / warp, Fl = scale * l, Fh = scale * h;
  float sl = (Fl - low) / (l - low), sr = (high - Fh) / (high - h);
  if (freq < l) return low + sl * (freq - low);
  if (freq < h) return scale * freq;
  return high + sr * (freq - high);
}
static float vtln_warp_mel(float vl, float vh, float low, float high, float warp, float mel) {
  return mel_scale(vtln_warp_freq(vl, vh, low, high, warp, inv_mel_scale(mel)));
}

extern "C" {

void vbh_mfcc_opts_default(vbh_mfcc_opts *o) {
  o->samp_freq = 16000.0f; o->frame_shift_ms = 10.0f; o->frame_length_ms = 25.0f;
  o->dither = 0.0f; o->preemph_coeff = 0.97f; o->remove_dc_offset = 1;
  o->window_type = VBH_WIN_POVEY; o->round_to_power_of_two = 1; o->blackman_coeff = 0.42f;
  o->snip_edges = 1; o->num_bins = 23; o->low_freq = 20.0f; o->high_freq = 0.0f;
  o->vtln_low = 100.0f; o->vtln_high = -500.0f; o->htk_mode = 0; o->num_ceps = 13;
  o->use_energy = 1; o->energy_floor = 0.0f; o->raw_energy = 1; o->cepstral_lifter = 22.0f;
  o->htk_compat = 0;
}

int32_t vbh_mfcc_num_frames(const vbh_mfcc_opts *o, int64_t num_samples) {
  int64_t shift = win_shift(o), len = win_size(o);
  if (o->snip_edges) {
    if (num_samples < len) return 0;
    return (int32_t)(1 + (num_samples - len) / shift);
  }
  return (int32_t)((num_samples + shift / 2) / shift);
}

int vbh_mfcc_create(vbh_ctx *ctx, const vbh_mfcc_opts *o, float vtln_warp, vbh_mfcc **out) {
  VBH_CHECK(ctx && o && out, "vbh_mfcc_create: NULL argument");
  VBH_CHECK(o->dither >= 0.0f, "vbh_mfcc_create: negative dither");
  const int flen = win_size(o), N = padded_size(o), shift = win_shift(o);
  VBH_CHECK(flen > 1 && shift > 0, "vbh_mfcc_create: bad frame length/shift");
  VBH_CHECK((N & (N - 1)) == 0 && N >= 64 && N <= 4096, "vbh_mfcc_create: padded window %d must be a power of two in [64, 4096]", N);
  VBH_CHECK(o->num_bins >= 3 && o->num_bins <= 64, "vbh_mfcc_create: num_bins %d out of range [3,64]", o->num_bins);
  VBH_CHECK(o->num_ceps >= 1 && o->num_ceps <= o->num_bins, "vbh_mfcc_create: num_ceps %d out of range", o->num_ceps);
  if (vbh_use_device(ctx)) return -1;

  // window (feature-window.cc:109-131)
  std::vector<float> window(flen);
  {
    const double a = 6.283185307179586476925286766559005 / (flen - 1);
    for (int i = 0; i < flen; i++) {
      double x = i, w;
      switch (o->window_type) {
        case VBH_WIN_HANNING: w = 0.5 - 0.5 * cos(a * x); break;
        case VBH_WIN_HAMMING: w = 0.54 - 0.46 * cos(a * x); break;
        case VBH_WIN_POVEY: w = pow(0.5 - 0.5 * cos(a * x), 0.85); break;
        case VBH_WIN_RECTANGULAR: w = 1.0; break;
        case VBH_WIN_BLACKMAN:
          w = o->blackman_coeff - 0.5 * cos(a * x) + (0.5 - o->blackman_coeff) * cos(2 * a * x);
          break;
        default: vbh_set_error("vbh_mfcc_create: invalid window type %d", o->window_type); return -1;
      }
      window[i] = (float)w;
    }
  }
  // twiddles (cos, sin)(2 pi k / N), k < N/2
  std::vector<float> tw(N);
  for (int k = 0; k < N / 2; k++) {
    double ang = 6.283185307179586476925286766559005 * k / N;
    tw[2 * k] = (float)cos(ang);
    tw[2 * k + 1] = (float)sin(ang);
  }
  // mel banks (mel-computations.cc:33-144)
  const int nb = o->num_bins, nfft = N / 2;
  std::vector<int> first(nb), len(nb), off(nb);
  std::vector<float> melw;
  {
    const float nyquist = 0.5f * o->samp_freq;
    float low = o->low_freq, high = (o->high_freq > 0.0f) ? o->high_freq : nyquist + o->high_freq;
    VBH_CHECK(!(low < 0.0f || low >= nyquist || high <= 0.0f || high > nyquist || high <= low),
              "vbh_mfcc_create: bad low-freq %g / high-freq %g vs nyquist %g", low, high, nyquist);
    const float bin_width = o->samp_freq / N;
    const float mel_low = mel_scale(low), mel_high = mel_scale(high);
    const float delta = (mel_high - mel_low) / (nb + 1);
    float vl = o->vtln_low, vh = o->vtln_high;
    if (vh < 0.0f) vh += nyquist;
    VBH_CHECK(!(vtln_warp != 1.0f && (vl < 0.0f || vl <= low || vl >= high || vh <= 0.0f || vh >= high || vh <= vl)),
              "vbh_mfcc_create: bad vtln-low %g / vtln-high %g", vl, vh);
    for (int b = 0; b < nb; b++) {
      float left = mel_low + b * delta, center = mel_low + (b + 1) * delta, right = mel_low + (b + 2) * delta;
      if (vtln_warp != 1.0f) {
        left = vtln_warp_mel(vl, vh, low, high, vtln_warp, left);
        center = vtln_warp_mel(vl, vh, low, high, vtln_warp, center);
        right = vtln_warp_mel(vl, vh, low, high, vtln_warp, right);
      }
      std::vector<float> row(nfft, 0.0f);
      int fi = -1, li = -1;
      for (int i = 0; i < nfft; i++) {
        float mel = mel_scale(bin_width * i);
        if (mel > left && mel < right) {
          row[i] = (mel <= center) ? (mel - left) / (center - left) : (right - mel) / (right - center);
          if (fi == -1) fi = i;
          li = i;
        }
      }
      VBH_CHECK(fi != -1, "vbh_mfcc_create: empty mel bin %d (num-mel-bins too large)", b);
      if (o->htk_mode && b == 0 && mel_low != 0.0f) row[fi] = 0.0f;
      first[b] = fi; len[b] = li + 1 - fi; off[b] = (int)melw.size();
      melw.insert(melw.end(), row.begin() + fi, row.begin() + li + 1);
    }
  }
  // DCT rows 0..nceps-1 of the nb x nb DCT-II (matrix-functions.cc:592-608, Real = float)
  std::vector<float> dct((size_t)o->num_ceps * nb);
  {
    float nrm = (float)sqrt(1.0 / (float)nb);
    for (int j = 0; j < nb; j++) dct[j] = nrm;
    nrm = (float)sqrt(2.0 / (float)nb);
    for (int k = 1; k < o->num_ceps; k++)
      for (int n = 0; n < nb; n++) dct[(size_t)k * nb + n] = (float)(nrm * cos(M_PI / nb * (n + 0.5) * k));
  }
  std::vector<float> lifter(o->num_ceps, 1.0f);
  if (o->cepstral_lifter != 0.0f)
    for (int i = 0; i < o->num_ceps; i++) lifter[i] = (float)(1.0 + 0.5 * o->cepstral_lifter * sin(M_PI * i / o->cepstral_lifter));

  vbh_mfcc *m = new vbh_mfcc();
  m->ctx = ctx; m->opts = *o;
  MfccParams &p = m->p;
  p.flen = flen; p.padded = N; p.shift = shift; p.nbins = nb; p.nceps = o->num_ceps;
  p.log2m = 0; while ((1 << p.log2m) < N / 2) p.log2m++;
  p.snip_edges = o->snip_edges; p.remove_dc = o->remove_dc_offset; p.use_energy = o->use_energy;
  p.raw_energy = o->raw_energy; p.htk_compat = o->htk_compat; p.htk_mode = o->htk_mode;
  p.use_lifter = o->cepstral_lifter != 0.0f; p.preemph = o->preemph_coeff;
  p.energy_floor = o->energy_floor; p.log_energy_floor = o->energy_floor > 0.0f ? logf(o->energy_floor) : 0.0f;
  p.melw_total = (int)melw.size();
  p.dither = o->dither; p.seed_lo = 0x2545F491u; p.seed_hi = 0x4F6CDD1Du; p.utt_base = 0; p.utt_ids = nullptr;
  hipStream_t s = ctx->stream;
  m->fast = N == kFastN && flen <= kFastN && nb <= 32 && o->num_ceps <= 16 && !getenv("VBH_MFCC_GENERIC");
  if (m->fast) {
    // per spectrum bin: the (at most two, adjacent) filters it lies under; anything else takes the general kernel
    std::vector<float> binw(4 * 257, 0.0f);
    std::vector<int> cnt(257, 0);
    for (int b = 0; b < nb && m->fast; b++)
      for (int i = 0; i < len[b]; i++) {
        const int k = first[b] + i;
        const float w = melw[off[b] + i];
        if (w == 0.0f) continue;
        if (k > 256) { m->fast = false; break; }
        if (cnt[k] == 0) { binw[4 * k] = (float)b; binw[4 * k + 1] = w; }
        else if (cnt[k] == 1 && (int)binw[4 * k] == b - 1) binw[4 * k + 2] = w;
        else { m->fast = false; break; }
        cnt[k]++;
      }
    std::vector<float> t256(512), t512(2 * 129);
    for (int i = 0; i < 256; i++) { const double ang = 6.283185307179586476925286766559005 * i / 256.0; t256[2 * i] = (float)cos(ang); t256[2 * i + 1] = (float)-sin(ang); }
    for (int i = 0; i <= 128; i++) { const double ang = 6.283185307179586476925286766559005 * i / 512.0; t512[2 * i] = (float)cos(ang); t512[2 * i + 1] = (float)-sin(ang); }
    if (m->fast && (m->t256.upload(t256.data(), t256.size(), s) || m->t512.upload(t512.data(), t512.size(), s) || m->binw.upload(binw.data(), binw.size(), s))) {
      delete m; return -1;
    }
    m->fast_lds = sizeof(float) * (512 + kWaves * (2 * (size_t)kFastBuf + 32));
  }
  if (m->window.upload(window.data(), window.size(), s) || m->twiddle.upload(tw.data(), tw.size(), s) ||
      m->mel_w.upload(melw.data(), melw.size(), s) || m->dct.upload(dct.data(), dct.size(), s) ||
      m->lifter.upload(lifter.data(), lifter.size(), s) || m->mel_first.upload(first.data(), nb, s) ||
      m->mel_len.upload(len.data(), nb, s) || m->mel_off.upload(off.data(), nb, s)) {
    delete m;
    return -1;
  }
  VBH_HIP(hipStreamSynchronize(s));
  const int mel_pad = (nb + 3) & ~3;
  const size_t per_wave = 2 * (size_t)N + (N / 2 + 4) + mel_pad;
  m->lds_bytes = sizeof(float) * (flen + N + (size_t)o->num_ceps * nb + o->num_ceps + melw.size() + 3 * nb + kWaves * per_wave);
  VBH_CHECK(m->lds_bytes <= 160 * 1024, "vbh_mfcc_create: LDS need %zu B exceeds 160 KiB", m->lds_bytes);
  *out = m;
  return 0;
}

void vbh_mfcc_destroy(vbh_mfcc *m) {
  if (!m) return;
  (void)hipSetDevice(m->ctx->device);
  (void)hipStreamSynchronize(m->ctx->stream);
  delete m;
}

int vbh_mfcc_set_dither_stream(vbh_mfcc *m, uint64_t seed, int64_t first_utt_index) {
  VBH_CHECK(m, "vbh_mfcc_set_dither_stream: NULL argument");
  m->p.seed_lo = (unsigned)seed; m->p.seed_hi = (unsigned)(seed >> 32);
  m->p.utt_base = first_utt_index;
  return 0;
}

int vbh_mfcc_set_dither_ids(vbh_mfcc *m, const int64_t *h_ids, int32_t n) {
  VBH_CHECK(m && (h_ids || n == 0) && n >= 0, "vbh_mfcc_set_dither_ids: bad argument");
  m->dither_ids.assign(h_ids, h_ids + n);
  return 0;
}

int vbh_mfcc_compute(vbh_mfcc *m, const void *d_wave, int wave_is_int16, const int64_t *h_wave_offsets,
                     int32_t n_utt, float *d_feats, int64_t *h_frame_offsets) {
  VBH_CHECK(m && h_wave_offsets && h_frame_offsets, "vbh_mfcc_compute: NULL argument");
  VBH_CHECK(n_utt >= 0, "vbh_mfcc_compute: n_utt < 0");
  // the stream indices of vbh_mfcc_set_dither_ids serve the one call they were given for: taken (and dropped) before any
  // early return, so that a call that launches nothing cannot leave them to the next one
  std::vector<long long> ids(m->dither_ids.begin(), m->dither_ids.end());
  m->dither_ids.clear();
  VBH_CHECK(ids.empty() || (int)ids.size() == n_utt, "vbh_mfcc_compute: %zu dither stream indices for %d utterances", ids.size(), n_utt);
  if (vbh_use_device(m->ctx)) return -1;
  std::vector<long long> woff(n_utt + 1), foff(n_utt + 1);
  std::vector<int> cu, cf;
  foff[0] = 0;
  for (int u = 0; u < n_utt; u++) {
    VBH_CHECK(h_wave_offsets[u + 1] >= h_wave_offsets[u], "vbh_mfcc_compute: wave offsets not monotone at %d", u);
    const int T = vbh_mfcc_num_frames(&m->opts, h_wave_offsets[u + 1] - h_wave_offsets[u]);
    foff[u + 1] = foff[u] + T;
    for (int f0 = 0; f0 < T; f0 += kChunkFrames) { cu.push_back(u); cf.push_back(f0); }
  }
  for (int u = 0; u <= n_utt; u++) { woff[u] = h_wave_offsets[u]; h_frame_offsets[u] = foff[u]; }
  if (cu.empty()) return 0;
  VBH_CHECK(d_wave && d_feats, "vbh_mfcc_compute: NULL device buffer");
  hipStream_t s = m->ctx->stream;
  vbh_arena *ar = vbh_arena_begin(m->ctx, vbh_arena_need(woff.size() * 8) + vbh_arena_need(foff.size() * 8) +
                                              vbh_arena_need(cu.size() * 4) + vbh_arena_need(cf.size() * 4) + vbh_arena_need(ids.size() * 8 + 8));
  if (!ar) return -1;
  MfccParams p_call = m->p;
  p_call.utt_ids = ids.empty() ? nullptr : vbh_arena_put(ar, ids.data(), ids.size());
  const long long *d_wave_off = vbh_arena_put(ar, woff.data(), woff.size()), *d_frame_off = vbh_arena_put(ar, foff.data(), foff.size());
  const int *d_chunk_utt = vbh_arena_put(ar, cu.data(), cu.size()), *d_chunk_f0 = vbh_arena_put(ar, cf.data(), cf.size());
  if (vbh_arena_commit(m->ctx, ar)) return -1;
  vbh_mfcc_tables t{m->window.p, m->twiddle.p, m->mel_first.p, m->mel_len.p, m->mel_off.p, m->mel_w.p, m->dct.p, m->lifter.p};
  const int n_chunks = (int)cu.size();
  int grid = n_chunks < 8 * m->ctx->num_cus ? n_chunks : 8 * m->ctx->num_cus;
  if (m->fast) {
#define VBH_LAUNCH_FAST(I16, DIT)                                                                                                       \
    {                                                                                                                                   \
      VBH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(mfcc512_kernel<I16, DIT>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  (int)m->fast_lds));                                                                                   \
      hipLaunchKernelGGL((mfcc512_kernel<I16, DIT>), dim3(grid), dim3(256), m->fast_lds, s, p_call, t, ft, d_wave,                        \
                         d_wave_off, d_frame_off, d_chunk_utt, d_chunk_f0, n_chunks, d_feats);                                          \
    }
    const bool dit = m->p.dither != 0.0f;
    const FastTables ft{m->t256.p, m->t512.p, m->binw.p};
    if (wave_is_int16) { if (dit) VBH_LAUNCH_FAST(true, true) else VBH_LAUNCH_FAST(true, false) }
    else { if (dit) VBH_LAUNCH_FAST(false, true) else VBH_LAUNCH_FAST(false, false) }
#undef VBH_LAUNCH_FAST
  } else if (wave_is_int16) {
    VBH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(mfcc_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)m->lds_bytes));
    hipLaunchKernelGGL(mfcc_kernel<true>, dim3(grid), dim3(256), m->lds_bytes, s, p_call, t, d_wave, d_wave_off,
                       d_frame_off, d_chunk_utt, d_chunk_f0, n_chunks, d_feats);
  } else {
    VBH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(mfcc_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)m->lds_bytes));
    hipLaunchKernelGGL(mfcc_kernel<false>, dim3(grid), dim3(256), m->lds_bytes, s, p_call, t, d_wave, d_wave_off,
                       d_frame_off, d_chunk_utt, d_chunk_f0, n_chunks, d_feats);
  }
  VBH_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
