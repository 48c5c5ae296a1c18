/*
 * vb_oracle_gmm.c -- CPU restatement of diagonal-GMM scoring (test infrastructure only;
 * see vb_oracle.h).  Reference paths relative to /root/reference/kaldi-master/src/.
 */
#include "vb_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>

#define VBO_LOG_2PI 1.8378770664093454835606594728112

/* gmm/diag-gmm.cc:114-152 ComputeGconsts.  Float accumulator, term order as written:
 * gc = log(w) + offset; gc += 0.5*log(iv) - 0.5*miv*miv/iv for d = 0..D-1.
 * Returns the number of "bad" (infinite) gconsts, -1 if any is NaN. */
int vbo_gmm_gconsts(int32_t G, int32_t D, const float *weights, const float *means_invvars,
                    const float *inv_vars, float *gconsts) {
  float offset = (float)(-0.5 * VBO_LOG_2PI * D);
  int bad = 0;
  for (int32_t g = 0; g < G; g++) {
    float gc = logf(weights[g]) + offset;
    for (int32_t d = 0; d < D; d++) {
      float iv = inv_vars[(size_t)g * D + d], miv = means_invvars[(size_t)g * D + d];
      /* reference expression mixes float and double literals: 0.5 is double, so the
       * right-hand side is evaluated in double and added to the float gc */
      gc = (float)(gc + (0.5 * logf(iv) - 0.5 * miv * miv / iv));
    }
    if (gc != gc) return -1;
    if (isinf(gc)) { bad++; if (gc > 0) gc = -gc; }
    gconsts[g] = gc;
  }
  return bad;
}

/* gmm/diag-gmm.cc:528-543 (vector form): loglikes = gconsts + miv.x - 0.5 iv.x^2 */
void vbo_gmm_component_loglikes(int32_t G, int32_t D, const float *gconsts,
                                const float *means_invvars, const float *inv_vars,
                                const float *x, float *out) {
  for (int32_t g = 0; g < G; g++) {
    const float *miv = means_invvars + (size_t)g * D, *iv = inv_vars + (size_t)g * D;
    double a = 0.0, b = 0.0;
    for (int32_t d = 0; d < D; d++) {
      float xs = x[d] * x[d]; /* data_sq is a float vector (ApplyPow(2.0)) */
      a += (double)miv[d] * (double)x[d];
      b += (double)iv[d] * (double)xs;
    }
    /* two sgemv calls with beta = 1: y = 1.0*(M x) + y, then y = -0.5*(V xsq) + y */
    float y = gconsts[g];
    y = (float)a + y;
    y = -0.5f * (float)b + y;
    out[g] = y;
  }
}

/* matrix/kaldi-vector.cc:757-775 LogSumExp (Real = float): float max, cutoff
 * max + log(FLT_EPSILON) (base/kaldi-math.h:124-125), expf terms summed in double,
 * result max + log(sum) rounded to float. */
float vbo_log_sum_exp(const float *v, int32_t n, float prune) {
  float max_elem = -INFINITY;
  for (int32_t i = 0; i < n; i++) if (v[i] > max_elem) max_elem = v[i];
  float cutoff = max_elem + logf(FLT_EPSILON);
  if (prune > 0.0f && max_elem - prune > cutoff) cutoff = max_elem - prune;
  double sum = 0.0;
  for (int32_t i = 0; i < n; i++) {
    float f = v[i];
    if (f >= cutoff) sum += expf(f - max_elem);
  }
  return (float)(max_elem + log(sum));
}

/* gmm/decodable-am-diag-gmm.cc:28-72 evaluated for every (frame, pdf); log_sum_exp_prune
 * is -1 in gmm-latgen-faster (decodable-am-diag-gmm.h:52-54 default). */
void vbo_gmm_loglikes(int32_t P, const int32_t *offs, int32_t D, const float *gconsts,
                      const float *means_invvars, const float *inv_vars,
                      const float *feats, int32_t T, float *out) {
  int32_t maxg = 0;
  for (int32_t p = 0; p < P; p++)
    if (offs[p + 1] - offs[p] > maxg) maxg = offs[p + 1] - offs[p];
  float *tmp = (float *)malloc(sizeof(float) * (maxg > 0 ? maxg : 1));
  for (int32_t t = 0; t < T; t++) {
    const float *x = feats + (size_t)t * D;
    for (int32_t p = 0; p < P; p++) {
      int32_t g0 = offs[p], ng = offs[p + 1] - g0;
      vbo_gmm_component_loglikes(ng, D, gconsts + g0, means_invvars + (size_t)g0 * D,
                                 inv_vars + (size_t)g0 * D, x, tmp);
      out[(size_t)t * P + p] = vbo_log_sum_exp(tmp, ng, -1.0f);
    }
  }
  free(tmp);
}
