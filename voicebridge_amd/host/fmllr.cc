// fmllr.cc -- see fmllr.h for what each function follows in the reference.
#include "fmllr.h"

#include <cmath>
#include <cstring>
#include <stdexcept>

namespace vbfmllr {

namespace {

// in-place inverse of a general n x n matrix (row-major, double) by Gauss-Jordan elimination with
// partial pivoting; returns log|det| (the reference: LAPACK dgetrf/dgetri, matrix/kaldi-matrix.cc Invert)
double InvertGeneral(std::vector<double> &m, int n) {
  std::vector<int> perm(n);
  double logdet = 0.0;
  for (int c = 0; c < n; c++) {
    int piv = c;
    double best = std::fabs(m[(size_t)c * n + c]);
    for (int r = c + 1; r < n; r++) if (std::fabs(m[(size_t)r * n + c]) > best) { best = std::fabs(m[(size_t)r * n + c]); piv = r; }
    if (best == 0.0) throw std::runtime_error("fMLLR: singular matrix");
    perm[c] = piv;
    if (piv != c) for (int k = 0; k < n; k++) std::swap(m[(size_t)c * n + k], m[(size_t)piv * n + k]);
    const double d = m[(size_t)c * n + c];
    logdet += std::log(std::fabs(d));
    const double inv = 1.0 / d;
    m[(size_t)c * n + c] = 1.0;
    for (int k = 0; k < n; k++) m[(size_t)c * n + k] *= inv;
    for (int r = 0; r < n; r++) {
      if (r == c) continue;
      const double f = m[(size_t)r * n + c];
      if (f == 0.0) continue;
      m[(size_t)r * n + c] = 0.0;
      for (int k = 0; k < n; k++) m[(size_t)r * n + k] -= f * m[(size_t)c * n + k];
    }
  }
  for (int c = n - 1; c >= 0; c--)
    if (perm[c] != c) for (int r = 0; r < n; r++) std::swap(m[(size_t)r * n + c], m[(size_t)r * n + perm[c]]);
  return logdet;
}

// FmllrAuxFuncDiagGmm (fmllr-diag-gmm.cc:496-508)
double AuxFunc(const std::vector<double> &W, int dim, double beta, const std::vector<double> &K, const std::vector<std::vector<double>> &G) {
  const int n1 = dim + 1;
  std::vector<double> A((size_t)dim * dim);
  for (int r = 0; r < dim; r++) for (int c = 0; c < dim; c++) A[(size_t)r * dim + c] = W[(size_t)r * n1 + c];
  double obj = beta * InvertGeneral(A, dim);
  for (size_t i = 0; i < W.size(); i++) obj += W[i] * K[i];
  std::vector<double> rg(n1);
  for (int d = 0; d < dim; d++) {
    const double *w = &W[(size_t)d * n1];
    double q = 0.0;
    for (int i = 0; i < n1; i++) {
      double s = 0.0;
      for (int j = 0; j < n1; j++) s += G[d][(size_t)i * n1 + j] * w[j];
      q += s * w[i];
    }
    obj -= 0.5 * q;
  }
  return obj;
}

}  // namespace

void GaussLogLikes(const PdfView &pdf, int dim, const float *x, std::vector<float> *ll) {
  ll->resize(pdf.num_gauss);
  for (int g = 0; g < pdf.num_gauss; g++) {
    const float *miv = pdf.means_invvars + (size_t)g * dim, *iv = pdf.inv_vars + (size_t)g * dim;
    float lin = 0.0f, quad = 0.0f;
    for (int d = 0; d < dim; d++) { lin += miv[d] * x[d]; quad += iv[d] * (x[d] * x[d]); }
    (*ll)[g] = (pdf.gconsts[g] + lin) + -0.5f * quad;
  }
}

FmllrAccs::FmllrAccs(int dim)
    : dim_(dim), beta_(0.0), K_((size_t)dim * (dim + 1), 0.0), G_(dim, std::vector<double>((size_t)(dim + 1) * (dim + 1), 0.0)),
      x_(dim, 0.0f), a_(dim, 0.0f), b_(dim, 0.0f), count_(0.0), have_x_(false) {}

void FmllrAccs::Commit() {       // CommitSingleFrameStats (fmllr-diag-gmm.cc:562-596), "full" update type
  if (count_ == 0.0) return;
  const int n1 = dim_ + 1;
  std::vector<double> xp(n1);
  for (int i = 0; i < dim_; i++) xp[i] = x_[i];
  xp[dim_] = 1.0;
  beta_ += count_;
  for (int r = 0; r < dim_; r++) { const double a = a_[r]; for (int c = 0; c < n1; c++) K_[(size_t)r * n1 + c] += a * xp[c]; }
  for (int d = 0; d < dim_; d++) {
    const double b = b_[d];
    std::vector<double> &G = G_[d];
    for (int i = 0; i < n1; i++) { const double bi = b * xp[i]; for (int j = 0; j <= i; j++) G[(size_t)i * n1 + j] += bi * xp[j]; }
  }
  count_ = 0.0;
  std::fill(a_.begin(), a_.end(), 0.0f);
  std::fill(b_.begin(), b_.end(), 0.0f);
}

void FmllrAccs::AddStats(double beta, const double *K, const double *G) {
  Commit();
  const int n1 = dim_ + 1;
  beta_ += beta;
  for (size_t i = 0; i < K_.size(); i++) K_[i] += K[i];
  for (int d = 0; d < dim_; d++) for (size_t i = 0; i < (size_t)n1 * n1; i++) G_[d][i] += G[(size_t)d * n1 * n1 + i];
}

void FmllrAccs::AccumulateFromPosteriors(const PdfView &pdf, const float *x, const float *post) {
  if (!have_x_ || memcmp(x, x_.data(), sizeof(float) * dim_) != 0) {     // DataHasChanged
    Commit();
    memcpy(x_.data(), x, sizeof(float) * dim_);
    have_x_ = true;
    count_ = 0.0;
    std::fill(a_.begin(), a_.end(), 0.0f);
    std::fill(b_.begin(), b_.end(), 0.0f);
  }
  float sum = 0.0f;
  for (int g = 0; g < pdf.num_gauss; g++) sum += post[g];
  count_ += sum;
  for (int g = 0; g < pdf.num_gauss; g++) {
    const float p = post[g];
    if (p == 0.0f) continue;
    const float *miv = pdf.means_invvars + (size_t)g * dim_, *iv = pdf.inv_vars + (size_t)g * dim_;
    for (int d = 0; d < dim_; d++) { a_[d] += p * miv[d]; b_[d] += p * iv[d]; }
  }
}

float FmllrAccs::AccumulateForGmm(const PdfView &pdf, const float *x, float weight) {
  std::vector<float> ll;
  GaussLogLikes(pdf, dim_, x, &ll);
  // ApplySoftMax (kaldi-vector.cc:852-859)
  float mx = ll[0];
  for (float v : ll) if (v > mx) mx = v;
  float sum = 0.0f;
  for (float &v : ll) sum += (v = std::exp(v - mx));
  const float inv = 1.0f / sum;
  for (float &v : ll) v = v * inv * weight;
  AccumulateFromPosteriors(pdf, x, ll.data());
  return mx + std::log(sum);
}

float FmllrAccs::Update(const FmllrOptions &opts, float *xform, float *count) {
  Commit();
  if (count) *count = (float)beta_;
  if (opts.update_type != "full" && opts.update_type != "diag" && opts.update_type != "offset" && opts.update_type != "none")
    throw std::runtime_error("Unknown fMLLR update type " + opts.update_type + ", fmllr-update-type must be one of \"full\"|\"diag\"|\"offset\"|\"none\"");
  if (!(beta_ > opts.min_count)) return 0.0f;        // "Not updating fMLLR since below min-count"
  const int dim = dim_, n1 = dim + 1;
  if (opts.update_type == "none") return 0.0f;
  if (opts.update_type != "full") {
    // ComputeFmllrMatrixDiagGmmDiagonal (fmllr-diag-gmm.cc:275-348) / ...Offset (:350-379): closed forms per row from K(i,i), K(i,dim) and
    // G_i(i,i), G_i(dim,i), G_i(dim,dim).  The reference accumulates only those elements of G for these update types, with the product
    // b_i x_i x_i rounded to float per frame (:583-590); here they are read out of the full statistics (double products): same values to
    // ~1e-7 relative.
    for (int r = 0; r < dim; r++) for (int c = 0; c < dim; c++)
      if (r != c && xform[(size_t)r * n1 + c] != 0.0f) throw std::runtime_error("fMLLR " + opts.update_type + " update: the initial transform must be diagonal");
    auto g = [&](int i, int r, int c) { return G_[i][(size_t)r * n1 + c]; };      // lower triangle: r >= c
    if (opts.update_type == "offset") {
      for (int i = 0; i < dim; i++) if (xform[(size_t)i * n1 + i] != 1.0f) throw std::runtime_error("fMLLR offset update: the initial transform must be the unit transform");
      float impr = 0.0f;
      for (int i = 0; i < dim; i++) {
        const double gdd = g(i, dim, dim), gid = g(i, dim, i), kid = K_[(size_t)i * n1 + dim];
        float b = xform[(size_t)i * n1 + dim];
        const float before = (float)(-0.5 * b * b * gdd - b * gid + b * kid);
        b = (float)((kid - gid) / gdd);
        xform[(size_t)i * n1 + dim] = b;
        const float after = (float)(-0.5 * b * b * gdd - b * gid + b * kid);
        impr += after - before;
      }
      return impr;
    }
    std::vector<std::vector<double>> G = G_;
    for (int d = 0; d < dim; d++) for (int i = 0; i < n1; i++) for (int j = i + 1; j < n1; j++) G[d][(size_t)i * n1 + j] = G[d][(size_t)j * n1 + i];
    std::vector<double> W((size_t)dim * n1);
    for (size_t i = 0; i < W.size(); i++) W[i] = xform[i];
    const float old_objf = (float)AuxFunc(W, dim, beta_, K_, G);
    for (int i = 0; i < dim; i++) {
      const double kii = K_[(size_t)i * n1 + i], kid = K_[(size_t)i * n1 + dim], giii = g(i, i, i), gidd = g(i, dim, dim), gidi = g(i, dim, i);
      const double a = gidi * gidi / gidd - giii, b = kii - gidi * kid / gidd, c = beta_;
      const double sc = (-b - std::sqrt(b * b - 4 * a * c)) / (2 * a);
      if (!(sc > 0.0)) throw std::runtime_error("fMLLR diag update: non-positive scale");
      const double o = (kid - sc * gidi) / gidd;
      xform[(size_t)i * n1 + i] = (float)sc;
      xform[(size_t)i * n1 + dim] = (float)o;
    }
    for (size_t i = 0; i < W.size(); i++) W[i] = xform[i];        // the reference evaluates the float matrix it stored
    return (float)AuxFunc(W, dim, beta_, K_, G) - old_objf;
  }
  // the symmetric halves of G were accumulated in the lower triangle only
  std::vector<std::vector<double>> G = G_, invG(dim);
  for (int d = 0; d < dim; d++) {
    for (int i = 0; i < n1; i++) for (int j = i + 1; j < n1; j++) G[d][(size_t)i * n1 + j] = G[d][(size_t)j * n1 + i];
    invG[d] = G[d];
    InvertGeneral(invG[d], n1);
  }
  std::vector<double> oldW((size_t)dim * n1), W((size_t)dim * n1);
  for (size_t i = 0; i < W.size(); i++) oldW[i] = W[i] = xform[i];
  const float old_objf = (float)AuxFunc(oldW, dim, beta_, K_, G);
  // FmllrInnerUpdate (fmllr-diag-gmm.cc:195-231) takes row d of inverse(A^T) -- column d of inverse(A) -- for every row of every
  // iteration, i.e. 40 x 39 inversions of a 39 x 39 matrix per transform in the reference.  Replacing row d of A is a rank-one
  // change, so inverse(A) is carried along by the Sherman-Morrison formula (O(dim^2) per row) and recomputed from scratch once
  // per iteration, which bounds the drift to 39 updates in double; same transforms (tests/test_host_fmllr.py compares with a
  // restatement that inverts every time).
  std::vector<double> M((size_t)dim * dim), row(n1), row_invg(n1), u(dim), v(dim), delta(dim);
  for (int iter = 0; iter < opts.num_iters; iter++) {
    for (int r = 0; r < dim; r++) for (int c = 0; c < dim; c++) M[(size_t)r * dim + c] = W[(size_t)r * n1 + c];
    InvertGeneral(M, dim);
    for (int d = 0; d < dim; d++) {
      for (int c = 0; c < dim; c++) row[c] = M[(size_t)c * dim + d];
      row[dim] = 0.0;
      const std::vector<double> &ig = invG[d];
      for (int i = 0; i < n1; i++) { double s = 0.0; for (int j = 0; j < n1; j++) s += ig[(size_t)i * n1 + j] * row[j]; row_invg[i] = s; }
      const double *k = &K_[(size_t)d * n1];
      double e1 = 0.0, e2 = 0.0;
      for (int i = 0; i < n1; i++) { e1 += row_invg[i] * row[i]; e2 += row_invg[i] * k[i]; }
      const double discr = std::sqrt(e2 * e2 + 4 * e1 * beta_);
      const double alpha1 = (-e2 + discr) / (2 * e1), alpha2 = (-e2 - discr) / (2 * e1);
      const double auxf1 = beta_ * std::log(std::fabs(alpha1 * e1 + e2)) - 0.5 * alpha1 * alpha1 * e1;
      const double auxf2 = beta_ * std::log(std::fabs(alpha2 * e1 + e2)) - 0.5 * alpha2 * alpha2 * e1;
      const double alpha = auxf1 > auxf2 ? alpha1 : alpha2;
      for (int i = 0; i < n1; i++) row[i] = row[i] * alpha + k[i];
      for (int i = 0; i < n1; i++) {
        double s = 0.0;
        for (int j = 0; j < n1; j++) s += ig[(size_t)i * n1 + j] * row[j];
        if (i < dim) delta[i] = s - W[(size_t)d * n1 + i];
        W[(size_t)d * n1 + i] = s;
      }
      // A' = A + e_d delta^T  ->  inverse(A') = M - (M e_d)(delta^T M) / (1 + delta^T M e_d)
      double denom = 1.0;
      for (int r = 0; r < dim; r++) { u[r] = M[(size_t)r * dim + d]; denom += delta[r] * u[r]; }
      if (denom == 0.0 || !std::isfinite(denom)) throw std::runtime_error("fMLLR: singular matrix");
      for (int c = 0; c < dim; c++) { double sv = 0.0; for (int r = 0; r < dim; r++) sv += delta[r] * M[(size_t)r * dim + c]; v[c] = sv / denom; }
      for (int r = 0; r < dim; r++) { const double ur = u[r]; double *mr = &M[(size_t)r * dim]; for (int c = 0; c < dim; c++) mr[c] -= ur * v[c]; }
    }
  }
  const float new_objf = (float)AuxFunc(W, dim, beta_, K_, G);
  const float impr = new_objf - old_objf;
  // "No applying fMLLR transform change because objective function did not increase."
  if (impr < 0.0f && !(std::fabs(new_objf - old_objf) <= 0.001f * (std::fabs(new_objf) + std::fabs(old_objf)))) return 0.0f;
  for (size_t i = 0; i < W.size(); i++) xform[i] = (float)W[i];
  return impr;
}

bool ComposeTransforms(const std::vector<float> &a, int ar, int ac, const std::vector<float> &b, int br, int bc, bool b_is_affine,
                       std::vector<float> *c, int *cr, int *cc) {
  if (br == 0 || ac == 0) return false;
  std::vector<float> bx;
  int bxr = br, bxc = bc;
  if (ac == br) bx = b;
  else if (ac == br + 1) {
    if (b_is_affine) { bxr = br + 1; bxc = bc; bx.assign((size_t)bxr * bxc, 0.0f); for (int r = 0; r < br; r++) for (int k = 0; k < bc; k++) bx[(size_t)r * bxc + k] = b[(size_t)r * bc + k]; bx[(size_t)br * bxc + bc - 1] = 1.0f; }
    else { bxr = br + 1; bxc = bc + 1; bx.assign((size_t)bxr * bxc, 0.0f); for (int r = 0; r < br; r++) for (int k = 0; k < bc; k++) bx[(size_t)r * bxc + k] = b[(size_t)r * bc + k]; bx[(size_t)br * bxc + bc] = 1.0f; }
  } else throw std::runtime_error("ComposeTransforms: mismatched dimensions, a has " + std::to_string(ac) + " columns and b has " + std::to_string(br) + " rows.");
  *cr = ar; *cc = bxc;
  c->assign((size_t)ar * bxc, 0.0f);
  for (int r = 0; r < ar; r++)
    for (int k = 0; k < bxc; k++) {
      float s = 0.0f;
      for (int m = 0; m < ac; m++) s += a[(size_t)r * ac + m] * bx[(size_t)m * bxc + k];
      (*c)[(size_t)r * bxc + k] = s;
    }
  return true;
}

}  // namespace vbfmllr
