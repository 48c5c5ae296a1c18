/*
 * voicebridge_decodable.h -- the algorithm seam of the reference over GPU scores: C++, header only.
 *
 * The reference's searches (LatticeFasterDecoder::Decode, FasterDecoder::Decode; K/decoder/) pull acoustic scores
 * through DecodableInterface (K/itf/decodable-itf.h:83-119): LogLikelihood(frame, transition-id), IsLastFrame,
 * NumFramesReady, NumIndices.  Its GMM implementation evaluates a pdf when asked (DecodableAmDiagGmmScaled,
 * K/gmm/decodable-am-diag-gmm.h:103-149).  Here the scores of every pdf of every frame of a batch are computed on the GPU
 * first (vbh_mfcc_compute -> vbh_featpost -> vbh_gmm_loglikes) and this class serves them through the same interface, so
 * that the reference's own CPU decoder runs unchanged on GPU scores -- BASELINE.json config 2, "MFCC + GMM-score only,
 * decoder still CPU":
 *
 *     vbh::GpuAcousticScores scores;                                  // host copy of the batch's [T x P] scores
 *     scores.Compute(ctx, gmm, d_feats, frame_offsets, n_utt);         // one GEMM-shaped kernel for the whole batch
 *     vbh::DecodableGpuScores<kaldi::DecodableInterface> decodable(scores.Rows(u), scores.NumFrames(u), scores.NumPdfs(),
 *                                                                  trans_model.id2pdf table, num_tids, acoustic_scale);
 *     decoder.Decode(&decodable);                                     // kaldi::LatticeFasterDecoder, as in gmm-latgen-faster.cc:112-135
 *
 * Base is the interface class to derive from: kaldi::DecodableInterface inside the reference's tree, or the stand-alone
 * vbh::DecodableBase (same virtual signatures) elsewhere.  Semantics follow DecodableAmDiagGmmScaled::LogLikelihood
 * (K/gmm/decodable-am-diag-gmm.h:142-145): scale * loglike(frame, TransitionIdToPdf(tid)); frames are 0-based,
 * transition-ids 1-based; an out-of-range request is the reference's KALDI_ERR, i.e. a std::runtime_error.
 */
#ifndef VOICEBRIDGE_DECODABLE_H_
#define VOICEBRIDGE_DECODABLE_H_

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "voicebridge_hip.h"

namespace vbh {

struct DecodableBase {            /* the shape of K/itf/decodable-itf.h:83-119 */
  virtual float LogLikelihood(int32_t frame, int32_t index) = 0;
  virtual bool IsLastFrame(int32_t frame) const = 0;
  virtual int32_t NumFramesReady() const = 0;
  virtual int32_t NumIndices() const = 0;
  virtual ~DecodableBase() {}
};

template <class Base = DecodableBase>
class DecodableGpuScores : public Base {
 public:
  /* scores: [num_frames x num_pdfs] row-major, unscaled log-likelihoods (rows of vbh_gmm_loglikes' output, on the host);
   * tid2pdf: [num_tids + 1], entry 0 unused (TransitionModel::id2pdf_id_, K/hmm/transition-model.h:324-328).
   * Nothing is copied: both arrays must outlive the object. */
  DecodableGpuScores(const float *scores, int32_t num_frames, int32_t num_pdfs, const int32_t *tid2pdf, int32_t num_tids, float scale)
      : scores_(scores), num_frames_(num_frames), num_pdfs_(num_pdfs), tid2pdf_(tid2pdf), num_tids_(num_tids), scale_(scale) {}
  float LogLikelihood(int32_t frame, int32_t tid) override {
    if (frame < 0 || frame >= num_frames_ || tid < 1 || tid > num_tids_)
      throw std::runtime_error("DecodableGpuScores: frame " + std::to_string(frame) + " / transition-id " + std::to_string(tid) +
                               " out of range (likely graph/model mismatch)");
    return scale_ * scores_[(size_t)frame * num_pdfs_ + tid2pdf_[tid]];
  }
  bool IsLastFrame(int32_t frame) const override { return frame == num_frames_ - 1; }
  int32_t NumFramesReady() const override { return num_frames_; }
  int32_t NumIndices() const override { return num_tids_; }

 private:
  const float *scores_;
  int32_t num_frames_, num_pdfs_;
  const int32_t *tid2pdf_;
  int32_t num_tids_;
  float scale_;
};

/* Scores of a batch of utterances on the host: the hand-off between the GPU front end and a CPU decoder. */
class GpuAcousticScores {
 public:
  /* d_feats: device [total_frames x dim] features of n_utt utterances (rows frame_offsets[u] .. frame_offsets[u+1]);
   * scores all pdfs of all frames on the GPU and brings the matrix to the host.  Throws std::runtime_error on failure. */
  void Compute(vbh_ctx *ctx, vbh_gmm *gmm, const float *d_feats, const int64_t *frame_offsets, int32_t n_utt) {
    offsets_.assign(frame_offsets, frame_offsets + n_utt + 1);
    num_pdfs_ = vbh_gmm_num_pdfs(gmm);
    const int64_t T = offsets_.back() - offsets_.front();
    host_.resize((size_t)T * num_pdfs_);
    if (T == 0) return;
    void *d_ll = nullptr;
    Check(vbh_dev_alloc(ctx, sizeof(float) * host_.size(), &d_ll), "vbh_dev_alloc");
    int rc = vbh_gmm_loglikes(gmm, d_feats + offsets_.front() * (int64_t)vbh_gmm_dim(gmm), T, static_cast<float *>(d_ll));
    if (rc == 0) rc = vbh_copy_d2h(ctx, host_.data(), d_ll, sizeof(float) * host_.size());
    (void)vbh_dev_free(ctx, d_ll);
    Check(rc, "vbh_gmm_loglikes");
  }
  int32_t NumPdfs() const { return num_pdfs_; }
  int32_t NumUtterances() const { return (int32_t)offsets_.size() - 1; }
  int32_t NumFrames(int32_t u) const { return (int32_t)(offsets_[u + 1] - offsets_[u]); }
  const float *Rows(int32_t u) const { return host_.data() + (size_t)(offsets_[u] - offsets_[0]) * num_pdfs_; }

 private:
  static void Check(int rc, const char *what) {
    if (rc != 0) throw std::runtime_error(std::string(what) + ": " + vbh_last_error());
  }
  std::vector<float> host_;
  std::vector<int64_t> offsets_;
  int32_t num_pdfs_ = 0;
};

}  // namespace vbh

#endif  /* VOICEBRIDGE_DECODABLE_H_ */
