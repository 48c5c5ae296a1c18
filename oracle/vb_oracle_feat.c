/*
 * vb_oracle_feat.c -- CPU restatement of the feature pipeline (test infrastructure only;
 * see vb_oracle.h).  Reference paths are relative to /root/reference/kaldi-master/src/.
 *
 * Floating-point notes.  The reference works in float32 with BLAS dot products whose
 * summation order is library specific (MKL in the product, OpenBLAS in oracle/_ref) and a
 * float split-radix FFT (matrix/srfft.cc:355-425).  This restatement keeps every stored
 * quantity in float32 exactly where the reference stores one, but evaluates reductions
 * (dot products, the FFT butterflies) in double and rounds once, which is the closest
 * order-independent statement of the same arithmetic.  Agreement with the reference build
 * is checked in tests/test_oracle_feat.py (tolerances stated there).
 */
#include "vb_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.1415926535897932384626433832795
#endif
#define VBO_2PI 6.283185307179586476925286766559005

void vbo_mfcc_opts_default(vbo_mfcc_opts *o) {
  /* feature-window.h:52-62, mel-computations.h:57-59, feature-mfcc.h:50-59 */
  o->samp_freq = 16000.0f; o->frame_shift_ms = 10.0f; o->frame_length_ms = 25.0f;
  o->dither = 1.0f; o->preemph_coeff = 0.97f; o->remove_dc_offset = 1;
  o->window_type = VBO_WIN_POVEY; o->round_to_power_of_two = 1;
  o->blackman_coeff = 0.42f; o->snip_edges = 1;
  o->num_bins = 23; o->low_freq = 20.0f; o->high_freq = 0.0f;
  o->vtln_low = 100.0f; o->vtln_high = -500.0f; o->htk_mode = 0;
  o->num_ceps = 13; o->use_energy = 1; o->energy_floor = 0.0f; o->raw_energy = 1;
  o->cepstral_lifter = 22.0f; o->htk_compat = 0;
}

/* feature-window.h:92-101 */
int32_t vbo_window_shift(const vbo_mfcc_opts *o) {
  return (int32_t)(o->samp_freq * 0.001 * o->frame_shift_ms);
}
int32_t vbo_window_size(const vbo_mfcc_opts *o) {
  return (int32_t)(o->samp_freq * 0.001 * o->frame_length_ms);
}
int32_t vbo_padded_window_size(const vbo_mfcc_opts *o) {
  int32_t n = vbo_window_size(o);
  if (!o->round_to_power_of_two) return n;
  int32_t p = 1;
  while (p < n) p <<= 1;
  return p;
}

/* feature-window.cc:28-39 */
int64_t vbo_first_sample_of_frame(int32_t frame, const vbo_mfcc_opts *o) {
  int64_t shift = vbo_window_shift(o);
  if (o->snip_edges) return (int64_t)frame * shift;
  int64_t mid = shift * frame + shift / 2;
  return mid - vbo_window_size(o) / 2;
}

/* feature-window.cc:41-87 (flush == true, the offline case) */
int32_t vbo_num_frames(int64_t num_samples, const vbo_mfcc_opts *o) {
  int64_t shift = vbo_window_shift(o), len = vbo_window_size(o);
  if (o->snip_edges) {
    if (num_samples < len) return 0;
    return (int32_t)(1 + (num_samples - len) / shift);
  }
  return (int32_t)((num_samples + shift / 2) / shift);
}

/* feature-window.cc:109-131: table in double, stored as float */
void vbo_window_function(const vbo_mfcc_opts *o, float *window) {
  int32_t n = vbo_window_size(o);
  double a = VBO_2PI / (n - 1);
  for (int32_t i = 0; i < n; i++) {
    double x = (double)i, w;
    switch (o->window_type) {
      case VBO_WIN_HANNING: w = 0.5 - 0.5 * cos(a * x); break;
      case VBO_WIN_HAMMING: w = 0.54 - 0.46 * cos(a * x); break;
      case VBO_WIN_POVEY: w = pow(0.5 - 0.5 * cos(a * x), 0.85); break;
      case VBO_WIN_RECTANGULAR: w = 1.0; break;
      default:
        w = o->blackman_coeff - 0.5 * cos(a * x) +
            (0.5 - o->blackman_coeff) * cos(2 * a * x);
    }
    window[i] = (float)w;
  }
}

/* mel-computations.cc:255-261 */
void vbo_lifter_coeffs(float Q, int32_t n, float *coeffs) {
  for (int32_t i = 0; i < n; i++)
    coeffs[i] = (float)(1.0 + 0.5 * Q * sin(M_PI * i / Q));
}

/* matrix-functions.cc:592-608 with Real = float */
void vbo_dct_matrix(int32_t K, int32_t N, float *M) {
  float normalizer = (float)sqrt(1.0 / (float)N);
  for (int32_t j = 0; j < N; j++) M[j] = normalizer;
  normalizer = (float)sqrt(2.0 / (float)N);
  for (int32_t k = 1; k < K; k++)
    for (int32_t n = 0; n < N; n++)
      M[k * N + n] = (float)(normalizer * cos((double)M_PI / N * (n + 0.5) * k));
}

/* mel-computations.h:81-87 */
static float mel_scale(float f) { return 1127.0f * logf(1.0f + f / 700.0f); }
static float inv_mel_scale(float m) { return 700.0f * (expf(m / 1127.0f) - 1.0f); }

/* mel-computations.cc:152-213 */
static float vtln_warp_freq(float vl, float vh, float low, float high, float warp,
                            float freq) {
  if (freq < low || freq > high) return freq;
  float one = 1.0f;
  float l = vl * (warp > one ? warp : one);
  float h = vh * (warp < one ? warp : one);
  float scale = 1.0f / warp;
  float Fl = scale * l, Fh = scale * h;
  float scale_left = (Fl - low) / (l - low);
  float scale_right = (high - Fh) / (high - h);
  if (freq < l) return low + scale_left * (freq - low);
  if (freq < h) return scale * freq;
  return high + scale_right * (freq - high);
}
/* mel-computations.cc:215-224 */
static float vtln_warp_mel(float vl, float vh, float low, float high, float warp,
                           float mel) {
  return mel_scale(vtln_warp_freq(vl, vh, low, high, warp, inv_mel_scale(mel)));
}

/* mel-computations.cc:33-144 */
int vbo_mel_banks(const vbo_mfcc_opts *o, float vtln_warp, int32_t *first,
                  int32_t *len, float *weights) {
  int32_t num_bins = o->num_bins;
  if (num_bins < 3) return -1;
  float sample_freq = o->samp_freq;
  int32_t padded = vbo_padded_window_size(o);
  if (padded % 2 != 0) return -1;
  int32_t num_fft_bins = padded / 2;
  float nyquist = 0.5f * sample_freq;
  float low_freq = o->low_freq, high_freq;
  if (o->high_freq > 0.0f) high_freq = o->high_freq;
  else high_freq = nyquist + o->high_freq;
  if (low_freq < 0.0f || low_freq >= nyquist || high_freq <= 0.0f ||
      high_freq > nyquist || high_freq <= low_freq)
    return -1;
  float fft_bin_width = sample_freq / padded;
  float mel_low = mel_scale(low_freq), mel_high = mel_scale(high_freq);
  float mel_delta = (mel_high - mel_low) / (num_bins + 1);
  float vtln_low = o->vtln_low, vtln_high = o->vtln_high;
  if (vtln_high < 0.0f) vtln_high += nyquist;
  if (vtln_warp != 1.0f &&
      (vtln_low < 0.0f || vtln_low <= low_freq || vtln_low >= high_freq ||
       vtln_high <= 0.0f || vtln_high >= high_freq || vtln_high <= vtln_low))
    return -1;
  memset(weights, 0, sizeof(float) * (size_t)num_bins * num_fft_bins);
  for (int32_t bin = 0; bin < num_bins; bin++) {
    float left = mel_low + bin * mel_delta, center = mel_low + (bin + 1) * mel_delta,
          right = mel_low + (bin + 2) * mel_delta;
    if (vtln_warp != 1.0f) {
      left = vtln_warp_mel(vtln_low, vtln_high, low_freq, high_freq, vtln_warp, left);
      center = vtln_warp_mel(vtln_low, vtln_high, low_freq, high_freq, vtln_warp, center);
      right = vtln_warp_mel(vtln_low, vtln_high, low_freq, high_freq, vtln_warp, right);
    }
    int32_t fi = -1, li = -1;
    float *row = weights + (size_t)bin * num_fft_bins;
    for (int32_t i = 0; i < num_fft_bins; i++) {
      float freq = fft_bin_width * i;
      float mel = mel_scale(freq);
      if (mel > left && mel < right) {
        float w;
        if (mel <= center) w = (mel - left) / (center - left);
        else w = (right - mel) / (right - center);
        row[i] = w;
        if (fi == -1) fi = i;
        li = i;
      }
    }
    if (fi == -1) return -1;
    first[bin] = fi;
    len[bin] = li + 1 - fi;
    if (o->htk_mode && bin == 0 && mel_low != 0.0f) row[fi] = 0.0f;
  }
  return 0;
}

/* Real FFT of n = 2^k points, evaluated in double (see the file header).  Output is
 * the complex half spectrum X[0..n/2]; the reference packs the same values as
 * [re0, re(n/2), re1, im1, ...] (srfft.cc:355-425); only |X|^2 is used downstream
 * (feature-functions.cc:29-51), so the sign convention of the imaginary part is moot. */
static void real_fft_power(const float *x, int32_t n, float *pspec, double *wr,
                           double *wi) {
  /* iterative radix-2 on a complex copy; n <= 4096 in practice */
  int32_t levels = 0;
  while ((1 << levels) < n) levels++;
  for (int32_t i = 0; i < n; i++) {
    int32_t r = 0;
    for (int32_t b = 0; b < levels; b++) r |= ((i >> b) & 1) << (levels - 1 - b);
    wr[r] = x[i];
    wi[r] = 0.0;
  }
  for (int32_t size = 2; size <= n; size <<= 1) {
    int32_t half = size >> 1;
    for (int32_t start = 0; start < n; start += size) {
      for (int32_t j = 0; j < half; j++) {
        double ang = -VBO_2PI * j / size, c = cos(ang), s = sin(ang);
        double ur = wr[start + j], ui = wi[start + j];
        double vr = wr[start + j + half] * c - wi[start + j + half] * s;
        double vi = wr[start + j + half] * s + wi[start + j + half] * c;
        wr[start + j] = ur + vr; wi[start + j] = ui + vi;
        wr[start + j + half] = ur - vr; wi[start + j + half] = ui - vi;
      }
    }
  }
  /* feature-functions.cc:29-51: power spectrum from the float32-stored spectrum */
  for (int32_t k = 0; k <= n / 2; k++) {
    float re = (float)wr[k], im = (float)wi[k];
    if (k == 0 || k == n / 2) pspec[k] = re * re;
    else pspec[k] = re * re + im * im;
  }
}

static float dotf(const float *a, const float *b, int32_t n) {
  double s = 0.0;
  for (int32_t i = 0; i < n; i++) s += (double)a[i] * (double)b[i];
  return (float)s;
}

int vbo_mfcc(const vbo_mfcc_opts *o, const float *wave, int64_t n, float vtln_warp,
             float *out, float *win_dump, float *pspec_dump, float *mel_dump) {
  if (o->dither != 0.0f) return -2; /* libc rand() dependent: not restated */
  int32_t T = vbo_num_frames(n, o);
  int32_t flen = vbo_window_size(o), padded = vbo_padded_window_size(o);
  int32_t nb = o->num_bins, nc = o->num_ceps, nfft = padded / 2;
  if ((padded & (padded - 1)) != 0) return -3; /* non power of two: out of scope */
  if (T == 0) return 0;
  float *window = (float *)malloc(sizeof(float) * flen);
  float *lifter = (float *)malloc(sizeof(float) * nc);
  float *dct_full = (float *)malloc(sizeof(float) * nb * nb);
  int32_t *first = (int32_t *)malloc(sizeof(int32_t) * nb), *len = (int32_t *)malloc(sizeof(int32_t) * nb);
  float *melw = (float *)malloc(sizeof(float) * (size_t)nb * nfft);
  float *frame = (float *)malloc(sizeof(float) * padded);
  float *pspec = (float *)malloc(sizeof(float) * (nfft + 1));
  float *mel = (float *)malloc(sizeof(float) * nb);
  double *wr = (double *)malloc(sizeof(double) * padded), *wi = (double *)malloc(sizeof(double) * padded);
  int rc = 0;
  vbo_window_function(o, window);
  if (o->cepstral_lifter != 0.0f) vbo_lifter_coeffs(o->cepstral_lifter, nc, lifter);
  /* feature-mfcc.cc:86-93: first num_ceps rows of the num_bins x num_bins DCT */
  vbo_dct_matrix(nb, nb, dct_full);
  if (vbo_mel_banks(o, vtln_warp, first, len, melw) != 0) { rc = -1; goto done; }
  float log_energy_floor = 0.0f;
  if (o->energy_floor > 0.0f) log_energy_floor = logf(o->energy_floor);
  int use_raw = o->use_energy && o->raw_energy;

  for (int32_t t = 0; t < T; t++) {
    /* feature-window.cc:162-220 ExtractWindow */
    int64_t start = vbo_first_sample_of_frame(t, o);
    if (start >= 0 && start + flen <= n) {
      memcpy(frame, wave + start, sizeof(float) * flen);
    } else {
      for (int32_t s = 0; s < flen; s++) {
        int64_t si = s + start;
        while (si < 0 || si >= n) {
          if (si < 0) si = -si - 1;
          else si = 2 * n - 1 - si;
        }
        frame[s] = wave[si];
      }
    }
    for (int32_t s = flen; s < padded; s++) frame[s] = 0.0f;
    /* feature-window.cc:133-156 ProcessWindow */
    if (o->remove_dc_offset) {
      double sum = 0.0; /* kaldi-vector.cc Sum(): double accumulator, float result */
      for (int32_t s = 0; s < flen; s++) sum += frame[s];
      float c = -((float)sum) / flen;
      for (int32_t s = 0; s < flen; s++) frame[s] += c;
    }
    float log_energy = 0.0f;
    if (use_raw) {
      float e = dotf(frame, frame, flen);
      if (e < FLT_EPSILON) e = FLT_EPSILON;
      log_energy = logf(e);
    }
    if (o->preemph_coeff != 0.0f) {
      for (int32_t s = flen - 1; s > 0; s--) frame[s] -= o->preemph_coeff * frame[s - 1];
      frame[0] -= o->preemph_coeff * frame[0];
    }
    for (int32_t s = 0; s < flen; s++) frame[s] *= window[s];
    if (win_dump) memcpy(win_dump + (size_t)t * padded, frame, sizeof(float) * padded);
    /* feature-mfcc.cc:39-41 */
    if (o->use_energy && !o->raw_energy) {
      float e = dotf(frame, frame, padded);
      if (e < FLT_MIN) e = FLT_MIN;
      log_energy = logf(e);
    }
    /* feature-mfcc.cc:43-50 */
    real_fft_power(frame, padded, pspec, wr, wi);
    if (pspec_dump) memcpy(pspec_dump + (size_t)t * (nfft + 1), pspec, sizeof(float) * (nfft + 1));
    /* mel-computations.cc:228-253 */
    for (int32_t b = 0; b < nb; b++) {
      float e = dotf(melw + (size_t)b * nfft + first[b], pspec + first[b], len[b]);
      if (o->htk_mode && e < 1.0f) e = 1.0f;
      /* feature-mfcc.cc:54-55 */
      if (e < FLT_EPSILON) e = FLT_EPSILON;
      mel[b] = logf(e);
    }
    if (mel_dump) memcpy(mel_dump + (size_t)t * nb, mel, sizeof(float) * nb);
    /* feature-mfcc.cc:57-62 */
    float *feat = out + (size_t)t * nc;
    for (int32_t k = 0; k < nc; k++) {
      float v = dotf(dct_full + (size_t)k * nb, mel, nb);
      if (o->cepstral_lifter != 0.0f) v *= lifter[k];
      feat[k] = v;
    }
    /* feature-mfcc.cc:64-68 */
    if (o->use_energy) {
      if (o->energy_floor > 0.0f && log_energy < log_energy_floor) log_energy = log_energy_floor;
      feat[0] = log_energy;
    }
    /* feature-mfcc.cc:70-79 */
    if (o->htk_compat) {
      float energy = feat[0];
      for (int32_t i = 0; i < nc - 1; i++) feat[i] = feat[i + 1];
      if (!o->use_energy) energy = (float)(energy * M_SQRT2);
      feat[nc - 1] = energy;
    }
  }
done:
  free(window); free(lifter); free(dct_full); free(first); free(len); free(melw);
  free(frame); free(pspec); free(mel); free(wr); free(wi);
  return rc;
}

/* feature-functions.cc:54-88.  scales row i has dims[i] = 2*i*window+1 entries, stored
 * in rows of width 2*order*window+1. */
void vbo_delta_scales(int32_t order, int32_t window, float *scales, int32_t *dims) {
  int32_t W = 2 * order * window + 1;
  memset(scales, 0, sizeof(float) * (size_t)(order + 1) * W);
  scales[0] = 1.0f;
  dims[0] = 1;
  for (int32_t i = 1; i <= order; i++) {
    const float *prev = scales + (size_t)(i - 1) * W;
    float *cur = scales + (size_t)i * W;
    int32_t prev_offset = (dims[i - 1] - 1) / 2, cur_offset = prev_offset + window;
    dims[i] = dims[i - 1] + 2 * window;
    float normalizer = 0.0f;
    for (int32_t j = -window; j <= window; j++) {
      normalizer += (float)(j * j);
      for (int32_t k = -prev_offset; k <= prev_offset; k++)
        cur[j + k + cur_offset] += (float)j * prev[k + prev_offset];
    }
    float alpha = (float)(1.0 / normalizer);
    for (int32_t k = 0; k < dims[i]; k++) cur[k] *= alpha;
  }
}

/* feature-functions.cc:90-111 + 160-171 */
void vbo_add_deltas(int32_t order, int32_t window, const float *in, int32_t T, int32_t D,
                    float *out) {
  int32_t W = 2 * order * window + 1;
  float *scales = (float *)malloc(sizeof(float) * (size_t)(order + 1) * W);
  int32_t *dims = (int32_t *)malloc(sizeof(int32_t) * (order + 1));
  vbo_delta_scales(order, window, scales, dims);
  int32_t Do = D * (order + 1);
  for (int32_t t = 0; t < T; t++) {
    float *o = out + (size_t)t * Do;
    for (int32_t k = 0; k < Do; k++) o[k] = 0.0f;
    for (int32_t i = 0; i <= order; i++) {
      const float *sc = scales + (size_t)i * W;
      int32_t max_offset = (dims[i] - 1) / 2;
      for (int32_t j = -max_offset; j <= max_offset; j++) {
        int32_t tt = t + j;
        if (tt < 0) tt = 0; else if (tt >= T) tt = T - 1;
        float s = sc[j + max_offset];
        if (s != 0.0f)
          for (int32_t d = 0; d < D; d++) o[i * D + d] += s * in[(size_t)tt * D + d];
      }
    }
  }
  free(scales); free(dims);
}

/* feature-functions.cc:205-226 */
void vbo_splice(const float *in, int32_t T, int32_t D, int32_t left, int32_t right,
                float *out) {
  int32_t N = 1 + left + right;
  for (int32_t t = 0; t < T; t++)
    for (int32_t j = 0; j < N; j++) {
      int32_t t2 = t + j - left;
      if (t2 < 0) t2 = 0;
      if (t2 >= T) t2 = T - 1;
      memcpy(out + ((size_t)t * N + j) * D, in + (size_t)t2 * D, sizeof(float) * D);
    }
}

/* cmvn.cc:30-62 */
void vbo_acc_cmvn_stats(const float *feats, int32_t T, int32_t D, double *stats) {
  for (int32_t t = 0; t < T; t++) {
    for (int32_t d = 0; d < D; d++) {
      float x = feats[(size_t)t * D + d], w = 1.0f;
      stats[d] += x * w;            /* products formed in float, accumulated in double */
      stats[(D + 1) + d] += x * x * w;
    }
    stats[D] += 1.0;
  }
}

/* cmvn.cc:64-113 */
int vbo_apply_cmvn(const double *stats, int32_t norm_vars, float *feats, int32_t T,
                   int32_t D) {
  double count = stats[D];
  if (count < 1.0) return -1;
  for (int32_t d = 0; d < D; d++) {
    double mean = stats[d] / count, offset, scale;
    if (!norm_vars) { scale = 1.0; offset = -mean; }
    else {
      double var = stats[(D + 1) + d] / count - mean * mean, floor_ = 1.0e-20;
      if (var < floor_) var = floor_;
      scale = 1.0 / sqrt(var);
      if (scale != scale || 1 / scale == 0.0) return -1;
      offset = -(mean * scale);
    }
    float fo = (float)offset, fs = (float)scale;
    for (int32_t t = 0; t < T; t++) {
      float *x = feats + (size_t)t * D + d;
      if (norm_vars) *x *= fs;
      *x += fo;
    }
  }
  return 0;
}

/* VB/src/featbin/transform-feats.cpp:89-107 */
int vbo_transform_feats(const float *xform, int32_t R, int32_t C, const float *in,
                        int32_t T, int32_t D, float *out) {
  if (C != D && C != D + 1) return -1;
  for (int32_t t = 0; t < T; t++)
    for (int32_t r = 0; r < R; r++) {
      float v = dotf(xform + (size_t)r * C, in + (size_t)t * D, D);
      if (C == D + 1) v += xform[(size_t)r * C + D];
      out[(size_t)t * R + r] = v;
    }
  return 0;
}

/* ---- CompressedMatrix, kSpeechFeature ("CM") ---- */
typedef struct { float min_value, range; int32_t num_rows, num_cols; } cm_header;

size_t vbo_compress_size(int32_t rows, int32_t cols) {
  return sizeof(cm_header) + (size_t)cols * (8 + (size_t)rows);
}

/* compressed-matrix.cc:347-356 */
static uint16_t float_to_u16(const cm_header *h, float value) {
  float f = (value - h->min_value) / h->range;
  if (f > 1.0f) f = 1.0f;
  if (f < 0.0f) f = 0.0f;
  return (uint16_t)(int)(f * 65535 + 0.499);
}
/* compressed-matrix.cc:371-377 */
static float u16_to_float(const cm_header *h, uint16_t v) {
  return h->min_value + h->range * 1.52590218966964e-05F * v;
}
/* compressed-matrix.cc:457-487 */
static uint8_t float_to_char(float p0, float p25, float p75, float p100, float value) {
  int ans;
  if (value < p25) {
    float f = (value - p0) / (p25 - p0);
    ans = (int)(f * 64 + 0.5);
    if (ans < 0) ans = 0;
    if (ans > 64) ans = 64;
  } else if (value < p75) {
    float f = (value - p25) / (p75 - p25);
    ans = 64 + (int)(f * 128 + 0.5);
    if (ans < 64) ans = 64;
    if (ans > 192) ans = 192;
  } else {
    float f = (value - p75) / (p100 - p75);
    ans = 192 + (int)(f * 63 + 0.5);
    if (ans < 192) ans = 192;
    if (ans > 255) ans = 255;
  }
  return (uint8_t)ans;
}
/* compressed-matrix.cc:490-500 */
static float char_to_float(float p0, float p25, float p75, float p100, uint8_t v) {
  if (v <= 64) return (float)(p0 + (p25 - p0) * v * (1 / 64.0));
  if (v <= 192) return (float)(p25 + (p75 - p25) * (v - 64) * (1 / 128.0));
  return (float)(p75 + (p100 - p75) * (v - 192) * (1 / 63.0));
}
static int cmp_float(const void *a, const void *b) {
  float x = *(const float *)a, y = *(const float *)b;
  return (x > y) - (x < y);
}
static uint16_t u16min(uint16_t a, uint16_t b) { return a < b ? a : b; }
static uint16_t u16max(uint16_t a, uint16_t b) { return a > b ? a : b; }

/* compressed-matrix.cc:60-163 (global header, method kSpeechFeature) + :379-455 (column
 * header; order statistics taken from a full sort, same values as nth_element) + :503-521 */
int vbo_compress(const float *mat, int32_t rows, int32_t cols, uint8_t *blob) {
  if (rows <= 0 || cols <= 0) return -1;
  cm_header h;
  float mn = mat[0], mx = mat[0];
  for (size_t i = 0; i < (size_t)rows * cols; i++) {
    if (mat[i] < mn) mn = mat[i];
    if (mat[i] > mx) mx = mat[i];
  }
  if (mx == mn) mx = (float)(mn + (1.0 + fabs(mn)));
  if (!(mn - mn == 0) || !(mx - mx == 0)) return -1;
  h.min_value = mn; h.range = mx - mn; h.num_rows = rows; h.num_cols = cols;
  memcpy(blob, &h, sizeof(h));
  uint16_t *pch = (uint16_t *)(blob + sizeof(h));
  uint8_t *bytes = blob + sizeof(h) + (size_t)cols * 8;
  float *sdata = (float *)malloc(sizeof(float) * rows);
  for (int32_t c = 0; c < cols; c++) {
    for (int32_t r = 0; r < rows; r++) sdata[r] = mat[(size_t)r * cols + c];
    qsort(sdata, rows, sizeof(float), cmp_float);
    uint16_t q0, q25, q75, q100;
    if (rows >= 5) {
      int q = rows / 4;
      q0 = u16min(float_to_u16(&h, sdata[0]), 65532);
      q25 = u16min(u16max(float_to_u16(&h, sdata[q]), (uint16_t)(q0 + 1)), 65533);
      q75 = u16min(u16max(float_to_u16(&h, sdata[3 * q]), (uint16_t)(q25 + 1)), 65534);
      q100 = u16max(float_to_u16(&h, sdata[rows - 1]), (uint16_t)(q75 + 1));
    } else {
      q0 = u16min(float_to_u16(&h, sdata[0]), 65532);
      q25 = rows > 1 ? u16min(u16max(float_to_u16(&h, sdata[1]), (uint16_t)(q0 + 1)), 65533) : (uint16_t)(q0 + 1);
      q75 = rows > 2 ? u16min(u16max(float_to_u16(&h, sdata[2]), (uint16_t)(q25 + 1)), 65534) : (uint16_t)(q25 + 1);
      q100 = rows > 3 ? u16max(float_to_u16(&h, sdata[3]), (uint16_t)(q75 + 1)) : (uint16_t)(q75 + 1);
    }
    pch[c * 4 + 0] = q0; pch[c * 4 + 1] = q25; pch[c * 4 + 2] = q75; pch[c * 4 + 3] = q100;
    float p0 = u16_to_float(&h, q0), p25 = u16_to_float(&h, q25),
          p75 = u16_to_float(&h, q75), p100 = u16_to_float(&h, q100);
    for (int32_t r = 0; r < rows; r++)
      bytes[(size_t)c * rows + r] = float_to_char(p0, p25, p75, p100, mat[(size_t)r * cols + c]);
  }
  free(sdata);
  return 0;
}

/* compressed-matrix.cc CopyToMat (format 1) */
int vbo_uncompress(const uint8_t *blob, float *mat) {
  cm_header h;
  memcpy(&h, blob, sizeof(h));
  const uint16_t *pch = (const uint16_t *)(blob + sizeof(h));
  const uint8_t *bytes = blob + sizeof(h) + (size_t)h.num_cols * 8;
  for (int32_t c = 0; c < h.num_cols; c++) {
    float p0 = u16_to_float(&h, pch[c * 4 + 0]), p25 = u16_to_float(&h, pch[c * 4 + 1]),
          p75 = u16_to_float(&h, pch[c * 4 + 2]), p100 = u16_to_float(&h, pch[c * 4 + 3]);
    for (int32_t r = 0; r < h.num_rows; r++)
      mat[(size_t)r * h.num_cols + c] = char_to_float(p0, p25, p75, p100, bytes[(size_t)c * h.num_rows + r]);
  }
  return 0;
}
