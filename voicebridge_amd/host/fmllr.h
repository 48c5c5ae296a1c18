// fmllr.h -- constrained MLLR (fMLLR) estimation for diagonal GMMs, host side.
//
// What each piece follows (paths relative to /root/reference/kaldi-master/src):
//   statistics beta, K [D x (D+1)], G_d [(D+1) x (D+1)], accumulated in double from per-frame
//   linear/quadratic terms a = sum_g gamma_g miv_g, b = sum_g gamma_g iv_g kept in float
//                                             transform/fmllr-diag-gmm.cc:30-44,542-596, transform-common.h (AffineXformStats)
//   AccumulateForGmm (component posteriors by soft-max of the per-Gaussian log-likelihoods)
//                                             transform/fmllr-diag-gmm.cc:100-113, gmm/diag-gmm.cc:528-543,601-615,
//                                             matrix/kaldi-vector.cc:852-859
//   row-by-row update of the full transform, auxiliary function
//                                             transform/fmllr-diag-gmm.cc:195-273, :350-380 (FmllrAuxFuncDiagGmm)
//   ComposeTransforms                          transform/transform-common.cc:132-166
// SURVEY.md 8f rank 2 (the estimation stages of DecodeFmllr).  The matrix inverses go through this
// file's own Gauss-Jordan elimination in double where the reference calls LAPACK, so transforms
// agree with the reference's to rounding (tests state the tolerance), not bit for bit.
#ifndef VOICEBRIDGE_FMLLR_H_
#define VOICEBRIDGE_FMLLR_H_

#include <cstdint>
#include <string>
#include <vector>

namespace vbfmllr {

// one pdf of the acoustic model: rows of length dim
struct PdfView { int num_gauss; const float *gconsts, *means_invvars, *inv_vars; };

struct FmllrOptions { std::string update_type = "full"; float min_count = 500.0f; int num_iters = 40; };

class FmllrAccs {
 public:
  explicit FmllrAccs(int dim);
  int Dim() const { return dim_; }
  // posterior[g] for the Gaussians of pdf (already scaled by the frame's weight)
  void AccumulateFromPosteriors(const PdfView &pdf, const float *x, const float *posterior);
  // soft-max of the pdf's per-Gaussian log-likelihoods times weight; returns the pdf's log-likelihood
  float AccumulateForGmm(const PdfView &pdf, const float *x, float weight);
  // xform [dim x (dim+1)] row-major, must hold a non-singular starting value (unit); returns objective improvement
  float Update(const FmllrOptions &opts, float *xform, float *count);
  double beta() const { return beta_; }
  void Commit();
  // statistics accumulated elsewhere (vbh_fmllr_accumulate): added to this object's; K [dim][dim+1], G [dim][dim+1][dim+1] full symmetric
  void AddStats(double beta, const double *K, const double *G);
 private:
  int dim_;
  double beta_;
  std::vector<double> K_;                 // [dim][dim+1]
  std::vector<std::vector<double>> G_;    // dim full symmetric (dim+1) x (dim+1) matrices
  std::vector<float> x_, a_, b_;          // single-frame statistics
  double count_;
  bool have_x_;
};

// per-Gaussian log-likelihoods of one pdf (diag-gmm.cc:528-543), float
void GaussLogLikes(const PdfView &pdf, int dim, const float *x, std::vector<float> *loglikes);
// c = a composed with b (transform-common.cc:132-166); false on empty input, throws on mismatch
bool ComposeTransforms(const std::vector<float> &a, int a_rows, int a_cols, const std::vector<float> &b, int b_rows, int b_cols,
                       bool b_is_affine, std::vector<float> *c, int *c_rows, int *c_cols);

}  // namespace vbfmllr
#endif
