// voicebridge_fmllrbin.cc -- the programs between the two searches of DecodeFmllr:
// lattice posteriors, silence weighting, Gaussian-level posteriors, fMLLR estimation, transform
// composition, lattice rescoring.  Declarations and the reference files they mirror:
// voicebridge_api.h.  Host code this round: per frame these touch the two or three pdfs a lattice
// holds at that time, not the whole model; the statistics accumulation is the part marked for the
// GPU (DESIGN.md "next").
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <functional>
#include <map>
#include <memory>
#include <set>
#include <sstream>
#include <thread>
#include <unordered_map>

#include "api_common.h"
#include "device_feats.h"
#include "fmllr.h"
#include "kaldi_io.h"
#include "lattice.h"
#include "voicebridge_api.h"
#include "voicebridge_hip.h"

using namespace vbapi;

namespace {

struct Gmm {       // the acoustic model as the host sees it: gconsts recomputed as DiagGmm::Read does (K/gmm/diag-gmm.cc:114-152)
  vbio::Model m;
  std::vector<float> gconsts;
  std::string path;      // the file it was read from (key of the thread's cached device form, CachedDeviceGmm)
  explicit Gmm(const std::string &path_in) : path(path_in) {
    vbio::ReadModel(path, &m);
    const int D = m.dim, G = m.pdf_offsets.back();
    gconsts.resize(G);
    const float offset = (float)(-0.5 * 1.8378770664093454835606594728112 * D);
    for (int i = 0; i < G; i++) {
      float gc = std::log(m.weights[i]) + offset;
      for (int d = 0; d < D; d++) {
        const float iv = m.inv_vars[(size_t)i * D + d], miv = m.means_invvars[(size_t)i * D + d];
        gc = (float)(gc + (0.5 * std::log(iv) - 0.5 * miv * miv / iv));
      }
      if (gc != gc) throw ApiError("At component " + std::to_string(i) + ", not a number in gconst computation");
      if (std::isinf(gc) && gc > 0) gc = -gc;
      gconsts[i] = gc;
    }
  }
  vbfmllr::PdfView Pdf(int p) const {
    const int a = m.pdf_offsets[p], D = m.dim;
    return vbfmllr::PdfView{m.pdf_offsets[p + 1] - a, gconsts.data() + a, m.means_invvars.data() + (size_t)a * D, m.inv_vars.data() + (size_t)a * D};
  }
  // DecodableAmDiagGmmUnmapped::LogLikelihoodZeroBased (K/gmm/decodable-am-diag-gmm.cc:28-72): float
  // per-Gaussian terms, LogSumExp with a double accumulator (K/matrix/kaldi-vector.cc:757-775)
  float LogLike(int pdf, const float *x) const {
    std::vector<float> ll;
    vbfmllr::GaussLogLikes(Pdf(pdf), m.dim, x, &ll);
    float mx = ll[0];
    for (float v : ll) if (v > mx) mx = v;
    const float cutoff = mx + -15.9423847198486328125f;
    double sum = 0.0;
    for (float v : ll) if (v >= cutoff) sum += std::exp(v - mx);
    return mx + (float)std::log(sum);
  }
};

// ConvertPosteriorToPdfs (K/hmm/posterior.cc:322-346), pdfs of a frame in increasing order
void PosteriorToPdfs(const std::vector<int32_t> &tid2pdf, const vbio::Posterior &in, vbio::Posterior *out) {
  out->assign(in.size(), {});
  for (size_t i = 0; i < in.size(); i++) {
    std::map<int32_t, float> acc;
    for (const auto &pr : in[i]) {
      if (pr.first <= 0 || pr.first >= (int32_t)tid2pdf.size()) throw ApiError("posterior: transition-id out of the model's range");
      acc[tid2pdf[pr.first]] += pr.second;
    }
    for (const auto &kv : acc) if (kv.second != 0.0f) (*out)[i].push_back(kv);
  }
}

vbfmllr::FmllrOptions ReadFmllrOptions(const Options &po) {
  vbfmllr::FmllrOptions o;
  o.update_type = po.Str("fmllr-update-type", "full");
  o.min_count = (float)po.Num("fmllr-min-count", 500.0);
  o.num_iters = (int)po.Num("fmllr-num-iters", 40);
  return o;
}

}  // namespace

// lattice-to-post   (VB/src/latbin/lattice-to-post.cpp; K/latbin/lattice-to-post.cc:28-115)
int LatticeToPost(int argc, char *argv[], std::ofstream &file_log) {
  return Guard("lattice-to-post", &file_log, [&](Log &log) -> int {
    Options po;
    po.Know({"acoustic-scale", "lm-scale"});
    po.Parse(argc, argv);
    if (po.args.size() < 2 || po.args.size() > 3) throw ApiError("Wrong arguments.");
    if (po.args.size() == 3 && !po.args[2].empty()) throw ApiError("the log-likelihood output table is not supported");
    const float acoustic_scale = (float)po.Num("acoustic-scale", 1.0), lm_scale = (float)po.Num("lm-scale", 1.0);
    if (acoustic_scale == 0.0f) throw ApiError("Do not use a zero acoustic scale (cannot be inverted)");
    vbio::LatticeReader reader(po.args[0]);
    int n_done = 0;
    double total_like = 0.0, total_ac = 0.0, total_time = 0.0;
    // the lattices of the table are independent: parsed, scaled, forward-backward on the host's threads; the table is written (a file),
    // or handed on as it is ("ark:mem:"), in table order
    struct Item { std::string key; const char *bytes = nullptr; size_t nbytes = 0; vbio::FlatPost post; std::string err; double like = 0, ac = 0; };
    std::vector<Item> items;
    Laps lap("lattice-to-post");
    { Item it; while (reader.NextBytes(&it.key, &it.bytes, &it.nbytes)) { items.push_back(std::move(it)); it = Item(); } }
    lap("table read");
    {
      std::atomic<size_t> next{0};
      auto work = [&]() {
        for (size_t i = next++; i < items.size(); i = next++) {
          Item &it = items[i];
          try {
            vblat::Lattice lat;
            vbio::ParseLatticeBytes(it.bytes, it.nbytes, &lat);
            if (acoustic_scale != 1.0f || lm_scale != 1.0f) vblat::ScaleLattice((double)lm_scale, 0.0, 0.0, (double)acoustic_scale, &lat);
            if (!vblat::IsTopSorted(lat) && !vblat::TopSort(&lat)) throw ApiError("Cycles detected in lattice.");
            vblat::Posterior post;
            it.like = vblat::LatticeForwardBackward(lat, &post, &it.ac);
            vbio::FlattenPosterior(post, &it.post);
          } catch (const std::exception &e) { it.err = e.what(); if (it.err.empty()) it.err = "error"; }
        }
      };
      const unsigned nt = std::max(1u, std::min<unsigned>(16, std::min<unsigned>((unsigned)items.size(), std::thread::hardware_concurrency())));
      std::vector<std::thread> pool;
      for (unsigned t = 1; t < nt; t++) pool.emplace_back(work);
      work();
      for (std::thread &t : pool) t.join();
    }
    lap("parse + forward-backward");
    auto table = std::make_shared<vbio::FlatPostTable>();
    table->reserve(items.size());
    for (Item &it : items) {
      if (!it.err.empty()) throw ApiError(it.err);
      total_like += it.like; total_time += (double)it.post.frames(); total_ac += it.ac;
      table->emplace_back(std::move(it.key), std::move(it.post));
      n_done++;
    }
    vbio::WritePosteriorTableFlat(po.args[1], std::move(table));
    lap("write");
    log.os() << "LOG (lattice-to-post) Overall average log-like/frame is " << (total_like / total_time) << " over " << total_time
             << " frames.  Average acoustic like/frame is " << (total_ac / total_time) << "\n";
    log.os() << "LOG (lattice-to-post) Done " << n_done << " lattices.\n";
    return n_done != 0 ? 0 : 1;
  });
}

// weight-silence-post   (VB/src/bin/weight-silence-post.cpp; K/bin/weight-silence-post.cc:28-110, K/hmm/posterior.cc:375-395)
int WeightSilencePost(int argc, char *argv[], std::ofstream &file_log) {
  return Guard("weight-silence-post", &file_log, [&](Log &log) -> int {
    Options po;
    po.Know({"distribute"});
    po.Parse(argc, argv);
    if (po.args.size() != 5) throw ApiError("Wrong arguments.");
    if (po.Bool("distribute", false)) throw ApiError("--distribute=true is not supported");
    char *end = nullptr;
    const float w = strtof(po.args[0].c_str(), &end);
    if (end == po.args[0].c_str() || *end) throw ApiError("Invalid silence-weight parameter: expected float, got \"" + po.args[0] + "\"");
    std::set<int32_t> sil;
    { std::string tok; std::istringstream is(po.args[1]); while (std::getline(is, tok, ':')) { if (tok.empty()) throw ApiError("Invalid silence-phones string " + po.args[1]); sil.insert(atoi(tok.c_str())); } }
    vbio::Model model;
    vbio::ReadModel(po.args[2], &model);
    int n = 0;
    // (per transition-id: is its phone a silence phone?)
    std::vector<char> tid_sil(model.trans.phone.size(), 0);
    for (size_t t = 1; t < model.trans.phone.size(); t++) tid_sil[t] = sil.count(model.trans.phone[t]) ? 1 : 0;
    Laps lap("weight-silence-post");
    const std::shared_ptr<const vbio::FlatPostTable> in = vbio::ReadPosteriorTableFlat(po.args[3]);
    lap("model + table read");
    auto out = std::make_shared<vbio::FlatPostTable>();
    out->reserve(in->size());
    for (const auto &kv : *in) {
      const vbio::FlatPost &src = kv.second;
      vbio::FlatPost dst;
      dst.off.reserve(src.off.size()); dst.e.reserve(src.e.size());
      for (size_t t = 0; t < src.frames(); t++) {
        for (int32_t i = src.off[t]; i < src.off[t + 1]; i++) {
          const std::pair<int32_t, float> pr = src.e[(size_t)i];
          if (pr.first <= 0 || pr.first >= (int32_t)tid_sil.size()) throw ApiError("posterior: transition-id out of the model's range");
          if (tid_sil[pr.first]) { if (w != 0.0f) dst.e.emplace_back(pr.first, pr.second * w); }
          else dst.e.push_back(pr);
        }
        dst.EndFrame();
      }
      out->emplace_back(kv.first, std::move(dst));
      n++;
    }
    vbio::WritePosteriorTableFlat(po.args[4], std::move(out));
    lap("weight + write");
    log.os() << "LOG (weight-silence-post) Done " << n << " posteriors.\n";
    return n != 0 ? 0 : 1;
  });
}

// gmm-post-to-gpost   (VB/src/gmmbin/gmm-post-to-gpost.cpp; K/gmmbin/gmm-post-to-gpost.cc:28-140)
int GmmPostToGpost(int argc, char *argv[], std::ofstream &file_log) {
  return Guard("gmm-post-to-gpost", &file_log, [&](Log &log) -> int {
    Options po;
    po.Know({"binary", "rand-prune"});
    po.Parse(argc, argv);
    if (po.args.size() != 4) throw ApiError("Wrong arguments.");
    if (po.Num("rand-prune", 0.0) > 0.0) throw ApiError("--rand-prune is not supported (it draws from rand())");
    const Gmm gmm(po.args[0]);
    std::map<std::string, vbio::Posterior> posts;
    for (auto &kv : vbio::ReadPosteriorTable(po.args[2])) posts.emplace(kv.first, std::move(kv.second));
    std::ofstream out = vbio::OpenArkForWriting(po.args[3]);
    vbio::MatrixReader r(po.args[1]);
    int num_done = 0, num_no_post = 0, num_err = 0;
    double tot_like = 0.0, tot_t = 0.0;
    std::string key; vbio::Matrix mat;
    std::vector<float> ll;
    while (r.Next(&key, &mat)) {
      auto it = posts.find(key);
      if (it == posts.end()) { num_no_post++; continue; }
      const vbio::Posterior &post = it->second;
      if ((int)post.size() != mat.rows) { log.os() << "WARNING (gmm-post-to-gpost) Posterior vector has wrong size " << post.size() << " vs. " << mat.rows << "\n"; num_err++; continue; }
      num_done++;
      vbio::Posterior pdf_post;
      PosteriorToPdfs(gmm.m.tid2pdf, post, &pdf_post);
      vbio::GaussPost gp(post.size());
      for (size_t i = 0; i < post.size(); i++) {
        const float *x = &mat.data[i * (size_t)mat.cols];
        for (const auto &pr : pdf_post[i]) {
          vbfmllr::GaussLogLikes(gmm.Pdf(pr.first), gmm.m.dim, x, &ll);
          float mx = ll[0];
          for (float v : ll) if (v > mx) mx = v;
          float sum = 0.0f;
          for (float &v : ll) sum += (v = std::exp(v - mx));
          const float inv = 1.0f / sum;
          bool any = false;
          for (float &v : ll) { v = v * inv * pr.second; any = any || v != 0.0f; }
          if (any) gp[i].emplace_back(pr.first, ll);
          tot_like += (double)(mx + std::log(sum)) * pr.second;
          tot_t += pr.second;
        }
      }
      vbio::WriteGaussPostEntry(out, key, gp);
    }
    log.os() << "LOG (gmm-post-to-gpost) Done " << num_done << " files, " << num_no_post << " with no posteriors, " << num_err << " with other errors.\n";
    log.os() << "LOG (gmm-post-to-gpost) Overall avg like per frame (Gaussian only) = " << (tot_like / tot_t) << " over " << tot_t << " frames.\n";
    return num_done != 0 ? 0 : 1;
  });
}

// The common body of gmm-est-fmllr and gmm-est-fmllr-gpost.  The statistics come from the GPU (vbh_fmllr_accumulate /
// _gpost: per-frame terms, then the f64-MFMA contraction per speaker); the row-by-row update of each unit's matrix
// stays on the host.  `add_utt(utt, feats, sink)` appends the utterance's posterior pairs to the batch (false: skip it).
struct PairSink {
  std::vector<int64_t> poff{0}, goff{0};
  std::vector<int32_t> ppdf; std::vector<float> pw, gpost;
  void EndFrame() { poff.push_back((int64_t)ppdf.size()); }
  void Clear() { poff.assign(1, 0); goff.assign(1, 0); ppdf.clear(); pw.clear(); gpost.clear(); }
};
template <class AddUtt>
static int EstimateOnGpu(const char *prog, bool use_gpost, const Options &po, const Gmm &gmm, const Gmm *ali_gmm, const std::string &feat_rspec,
                         const std::string &trans_wspec, Log &log, AddUtt add_utt) {
  const vbfmllr::FmllrOptions opts = ReadFmllrOptions(po);
  const int D = gmm.m.dim, n1 = D + 1;
  Laps lap(prog);
  // ---- the features: the table on the device (the thread's cached copy when this job's previous program read it), behind
  //      the fused front end when --fe-* options are given
  FrontEnd fe;
  fe.Parse(po);
  const RawTable &tab = CachedRawTable(feat_rspec);
  const size_t NU = tab.keys.size();
  // ---- the units a transform is estimated for (speakers, or utterances without --spk2utt) and their utterances
  std::vector<std::pair<std::string, std::vector<std::string>>> units;
  if (po.Has("spk2utt")) units = vbio::ReadSpk2Utt(po.Str("spk2utt", ""));
  else for (const std::string &k : tab.keys) units.emplace_back(k, std::vector<std::string>{k});
  vbh_ctx *ctx = SharedCtx();
  // (the device forms of the models are the thread's cached ones: no device allocation, hence no device-wide wait, per call)
  vbh_gmm *dgmm = gmm.path.empty() ? nullptr : CachedDeviceGmm(gmm.path, gmm.m);
  struct GmmGuard { vbh_gmm *g; ~GmmGuard() { if (g) vbh_gmm_destroy(g); } } guard{nullptr}, ali_guard{nullptr};
  if (!dgmm) {
    if (vbh_gmm_create(ctx, D, gmm.m.num_pdfs, gmm.m.pdf_offsets.data(), gmm.m.weights.data(), gmm.m.means_invvars.data(), gmm.m.inv_vars.data(), &dgmm) != 0)
      throw ApiError(std::string("vbh_gmm_create: ") + vbh_last_error());
    guard.g = dgmm;
  }
  // --ali-model (extension): the Gaussian posteriors come from that model, the statistics from `gmm` -- what the reference gets
  // by piping gmm-post-to-gpost into gmm-est-fmllr-gpost (VB/scr/steps/decode_fmllr.cpp first pass), without the gpost table
  vbh_gmm *dali = dgmm;
  if (ali_gmm) {
    if (ali_gmm->m.dim != D || ali_gmm->m.pdf_offsets != gmm.m.pdf_offsets || ali_gmm->m.tid2pdf != gmm.m.tid2pdf) throw ApiError("--ali-model: the alignment model's Gaussians do not match the model's");
    if (!ali_gmm->path.empty()) dali = CachedDeviceGmm(ali_gmm->path, ali_gmm->m);
    else {
      if (vbh_gmm_create(ctx, D, ali_gmm->m.num_pdfs, ali_gmm->m.pdf_offsets.data(), ali_gmm->m.weights.data(), ali_gmm->m.means_invvars.data(),
                         ali_gmm->m.inv_vars.data(), &ali_guard.g) != 0)
        throw ApiError(std::string("vbh_gmm_create: ") + vbh_last_error());
      dali = ali_guard.g;
    }
  }
  lap("feature table + device models");
  std::vector<int32_t> num_gauss(gmm.m.num_pdfs);
  for (int p = 0; p < gmm.m.num_pdfs; p++) num_gauss[p] = gmm.m.pdf_offsets[p + 1] - gmm.m.pdf_offsets[p];
  std::vector<vbfmllr::FmllrAccs> accs(units.size(), vbfmllr::FmllrAccs(D));
  std::vector<uint8_t> unit_has_utt(units.size(), 0);
  int num_done = 0, num_err = 0;
  // ---- the table's utterances in table order: unit of each (-1: not part of any unit, or skipped), its posterior pairs per frame
  std::vector<int32_t> u2s(NU, -1);
  for (size_t s_ = 0; s_ < units.size(); s_++)
    for (const std::string &utt : units[s_].second) {
      auto it = tab.index.find(utt);
      if (it == tab.index.end() || tab.rows((size_t)it->second) == 0) { log.os() << "WARNING (" << prog << ") Did not find features for utterance " << utt << "\n"; num_err++; continue; }
      u2s[(size_t)it->second] = (int32_t)s_;
    }
  PairSink sink;
  for (size_t u = 0; u < NU; u++) {
    const int rows = tab.rows(u);
    if (u2s[u] >= 0) {
      const size_t before = sink.poff.size();
      if (add_utt(tab.keys[u], rows, &sink)) { unit_has_utt[(size_t)u2s[u]] = 1; num_done++; continue; }
      if (sink.poff.size() != before) throw ApiError("internal: a skipped utterance left posterior frames behind");
      num_err++;
      u2s[u] = -1;
    }
    for (int t = 0; t < rows; t++) sink.EndFrame();      // an utterance without statistics: frames without pairs
  }
  lap("posterior pairs");
  // The "full" update (the default) runs on the device from the statistics where the accumulation left them (vbh_fmllr_update_full: one
  // wave per unit; the 128 MB of a 256-speaker batch stay there); the closed-form update types read the statistics on the host.
  const bool device_update = opts.update_type == "full" && !getenv("VB_FMLLR_HOST_UPDATE");
  std::vector<vbio::Matrix> xf(units.size());
  std::vector<float> imprs(units.size(), 0.0f), counts(units.size(), 0.0f);
  std::vector<double> betas(units.size(), 0.0);
  std::vector<std::string> uerr(units.size());
  for (size_t s = 0; s < units.size(); s++) {
    vbio::Matrix &t = xf[s];
    t.rows = D; t.cols = n1; t.data.assign((size_t)D * n1, 0.0f);
    for (int d = 0; d < D; d++) t.data[(size_t)d * n1 + d] = 1.0f;
  }
  bool updated_on_device = false;
  if (num_done > 0) {
    int fdim = 0;
    std::unique_ptr<DevBuffer> hold;
    const float *d_feats = DeviceFeatures(fe, tab, 0, NU, &fdim, &hold);
    lap("features on the device");
    if (fdim != D) throw ApiError("Dim mismatch: data dim = " + std::to_string(fdim) + " vs. model dim = " + std::to_string(D));
    const int S = (int)units.size();
    std::vector<double> beta(S), K, G;
    if (!device_update) { K.resize((size_t)S * D * n1); G.resize((size_t)S * D * n1 * n1); }
    double *pK = device_update ? nullptr : K.data(), *pG = device_update ? nullptr : G.data();
    const int rc = use_gpost
        ? vbh_fmllr_accumulate_gpost(ctx, dgmm, d_feats, (int32_t)NU, tab.offs.data(), u2s.data(), S, sink.poff.data(), sink.ppdf.data(),
                                     sink.goff.data(), sink.gpost.data(), num_gauss.data(), beta.data(), pK, pG, nullptr)
        : vbh_fmllr_accumulate(ctx, dali, dgmm, d_feats, (int32_t)NU, tab.offs.data(), u2s.data(), S, sink.poff.data(), sink.ppdf.data(),
                               sink.pw.data(), beta.data(), pK, pG, nullptr);
    if (rc != 0) throw ApiError(std::string("vbh_fmllr_accumulate: ") + vbh_last_error());
    lap("statistics on the device");
    for (int s_ = 0; s_ < S; s_++) betas[(size_t)s_] = beta[(size_t)s_];
    if (device_update) {
      std::vector<float> x((size_t)S * D * n1), im((size_t)S, 0.0f);
      for (int s_ = 0; s_ < S; s_++) memcpy(&x[(size_t)s_ * D * n1], xf[(size_t)s_].data.data(), sizeof(float) * (size_t)D * n1);
      if (vbh_fmllr_update_full(ctx, D, S, opts.min_count, opts.num_iters, x.data(), im.data(), nullptr) != 0) throw ApiError(std::string("fMLLR update: ") + vbh_last_error());
      for (int s_ = 0; s_ < S; s_++) {
        memcpy(xf[(size_t)s_].data.data(), &x[(size_t)s_ * D * n1], sizeof(float) * (size_t)D * n1);
        imprs[(size_t)s_] = im[(size_t)s_]; counts[(size_t)s_] = (float)beta[(size_t)s_];
      }
      updated_on_device = true;
      lap("row updates on the device");
    } else {
      for (int s_ = 0; s_ < S; s_++) if (beta[s_] != 0.0) accs[s_].AddStats(beta[s_], &K[(size_t)s_ * D * n1], &G[(size_t)s_ * D * n1 * n1]);
      lap("statistics added");
    }
  }
  // ---- the other update types (and units without statistics): from the unit transform on the host's threads, written in table order
  if (!updated_on_device) {
    std::atomic<size_t> next{0};
    auto work = [&]() {
      for (size_t s = next++; s < units.size(); s = next++) {
        if (!po.Has("spk2utt") && !unit_has_utt[s]) continue;
        try { imprs[s] = accs[s].Update(opts, xf[s].data.data(), &counts[s]); betas[s] = accs[s].beta(); } catch (const std::exception &e) { uerr[s] = e.what(); }
      }
    };
    const unsigned nt = std::max(1u, std::min<unsigned>(16, std::min<unsigned>((unsigned)units.size(), std::thread::hardware_concurrency())));
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < nt; t++) pool.emplace_back(work);
    work();
    for (std::thread &t : pool) t.join();
    lap("row updates");
  }
  vbio::MatrixWriter writer(trans_wspec);
  double tot_impr = 0.0, tot_t = 0.0;
  for (size_t s = 0; s < units.size(); s++) {
    if (!po.Has("spk2utt") && !unit_has_utt[s]) continue;     // per-utterance mode writes nothing for skipped utterances
    if (!uerr[s].empty()) throw ApiError("fMLLR update for " + units[s].first + ": " + uerr[s]);
    if (!(betas[s] > opts.min_count)) log.os() << "WARNING (" << prog << ") Not updating fMLLR since below min-count: count is " << betas[s] << "\n";
    writer.Write(units[s].first, xf[s]);
    log.os() << "LOG (" << prog << ") For " << (po.Has("spk2utt") ? "speaker " : "utterance ") << units[s].first << ", auxf-impr from fMLLR is " << (imprs[s] / counts[s])
             << ", over " << counts[s] << " frames.\n";
    tot_impr += imprs[s]; tot_t += counts[s];
  }
  writer.Close();
  log.os() << "LOG (" << prog << ") Done " << num_done << " files, " << num_err << " with errors.\n";
  log.os() << "LOG (" << prog << ") Overall fMLLR auxf impr per frame is " << (tot_impr / tot_t) << " over " << tot_t << " frames.\n";
  return num_done != 0 ? 0 : 1;
}

// gmm-est-fmllr-gpost   (VB/src/gmmbin/gmm-est-fmllr-gpost.cpp; K/gmmbin/gmm-est-fmllr-gpost.cc:28-190)
int GmmEstFmllrGpost(int argc, char *argv[], std::ofstream &file_log) {
  return Guard("gmm-est-fmllr-gpost", &file_log, [&](Log &log) -> int {
    Options po;
    po.Know({"spk2utt", "fmllr-update-type", "fmllr-min-count", "fmllr-num-iters"});
    FrontEnd::Know(&po);
    po.Parse(argc, argv);
    if (po.args.size() != 4) throw ApiError("Wrong arguments.");
    const Gmm gmm(po.args[0]);
    std::map<std::string, vbio::GaussPost> gposts;
    for (auto &kv : vbio::ReadGaussPostTable(po.args[2])) gposts.emplace(kv.first, std::move(kv.second));
    return EstimateOnGpu("gmm-est-fmllr-gpost", true, po, gmm, nullptr, po.args[1], po.args[3], log, [&](const std::string &utt, int feat_rows, PairSink *sink) {
      auto it = gposts.find(utt);
      if (it == gposts.end()) { log.os() << "WARNING (gmm-est-fmllr-gpost) Did not find posteriors for utterance " << utt << "\n"; return false; }
      const vbio::GaussPost &gp = it->second;
      if ((int)gp.size() != feat_rows) { log.os() << "WARNING (gmm-est-fmllr-gpost) GaussPost vector has wrong size " << gp.size() << " vs. " << feat_rows << "\n"; return false; }
      for (const auto &fr : gp)
        for (const auto &pr : fr)
          if (pr.first < 0 || pr.first >= gmm.m.num_pdfs || (int)pr.second.size() != gmm.m.pdf_offsets[pr.first + 1] - gmm.m.pdf_offsets[pr.first])
            throw ApiError("Gaussian posteriors do not match the model for utterance " + utt);
      for (const auto &fr : gp) {
        for (const auto &pr : fr) { sink->ppdf.push_back(pr.first); sink->gpost.insert(sink->gpost.end(), pr.second.begin(), pr.second.end()); sink->goff.push_back((int64_t)sink->gpost.size()); }
        sink->EndFrame();
      }
      return true;
    });
  });
}

// gmm-est-fmllr   (VB/src/gmmbin/gmm-est-fmllr.cpp; K/gmmbin/gmm-est-fmllr.cc:28-200)
int GmmEstFmllr(int argc, char *argv[], std::ofstream &file_log) {
  return Guard("gmm-est-fmllr", &file_log, [&](Log &log) -> int {
    Options po;
    po.Know({"spk2utt", "fmllr-update-type", "fmllr-min-count", "fmllr-num-iters", "ali-model"});
    FrontEnd::Know(&po);
    po.Parse(argc, argv);
    if (po.args.size() != 4) throw ApiError("Wrong arguments.");
    const Gmm gmm(po.args[0]);
    std::unique_ptr<Gmm> ali;
    if (po.Has("ali-model")) ali.reset(new Gmm(po.Str("ali-model", "")));
    Laps lap("gmm-est-fmllr");
    const std::shared_ptr<const vbio::FlatPostTable> table = vbio::ReadPosteriorTableFlat(po.args[2]);
    std::unordered_map<std::string, const vbio::FlatPost *> posts;
    posts.reserve(table->size() * 2);
    for (const auto &kv : *table) posts.emplace(kv.first, &kv.second);       // (a key met twice: the first entry, as std::map::emplace kept it)
    lap("models + posterior table read");
    std::vector<std::pair<int32_t, float>> fr;
    return EstimateOnGpu("gmm-est-fmllr", false, po, gmm, ali.get(), po.args[1], po.args[3], log, [&](const std::string &utt, int feat_rows, PairSink *sink) {
      auto pi = posts.find(utt);
      if (pi == posts.end()) { log.os() << "WARNING (gmm-est-fmllr) Did not find posteriors for utterance " << utt << "\n"; return false; }
      const vbio::FlatPost &post = *pi->second;
      if ((int)post.frames() != feat_rows) { log.os() << "WARNING (gmm-est-fmllr) Posterior vector has wrong size " << post.frames() << " vs. " << feat_rows << "\n"; return false; }
      // ConvertPosteriorToPdfs (K/hmm/posterior.cc:322-346) frame by frame: the weights of a frame's transition-ids that share a pdf added
      // in their order of appearance, pdfs in increasing order, zero totals dropped
      for (int32_t i = 0; i < (int32_t)post.e.size(); i++)
        if (post.e[(size_t)i].first <= 0 || post.e[(size_t)i].first >= (int32_t)gmm.m.tid2pdf.size()) throw ApiError("posterior: transition-id out of the model's range");
      for (size_t t = 0; t < post.frames(); t++) {
        fr.clear();
        for (int32_t i = post.off[t]; i < post.off[t + 1]; i++) {
          const int32_t pdf = gmm.m.tid2pdf[post.e[(size_t)i].first];
          size_t k = 0;
          while (k < fr.size() && fr[k].first < pdf) k++;
          if (k < fr.size() && fr[k].first == pdf) fr[k].second += post.e[(size_t)i].second;
          else fr.insert(fr.begin() + (std::ptrdiff_t)k, std::make_pair(pdf, 0.0f + post.e[(size_t)i].second));
        }
        for (const auto &pr : fr) if (pr.second != 0.0f) { sink->ppdf.push_back(pr.first); sink->pw.push_back(pr.second); }
        sink->EndFrame();
      }
      return true;
    });
  });
}

// compose-transforms   (VB/src/featbin/compose-transforms.cpp; K/featbin/compose-transforms.cc:28-165): both tables
int ComposeTransforms(int argc, char *argv[], std::ofstream &file_log) {
  return Guard("compose-transforms", &file_log, [&](Log &log) -> int {
    Options po;
    po.Know({"utt2spk", "b-is-affine", "binary"});
    po.Parse(argc, argv);
    if (po.args.size() != 3) throw ApiError("Wrong arguments.");
    if (po.Has("utt2spk")) throw ApiError("--utt2spk is not supported");
    const bool b_is_affine = po.Bool("b-is-affine", false);
    for (int i = 0; i < 3; i++)
      if (po.args[i].compare(0, 4, "ark:") != 0 && po.args[i].compare(0, 4, "scp:") != 0 && po.args[i].compare(0, 4, "ark,") != 0)
        throw ApiError("only the table form (both inputs rspecifiers, output a wspecifier) is supported");
    const std::map<std::string, vbio::Matrix> b = vbio::ReadMatrixTable(po.args[1]);
    vbio::MatrixReader a_reader(po.args[0]);
    vbio::MatrixWriter writer(po.args[2]);
    std::string key; vbio::Matrix a;
    while (a_reader.Next(&key, &a)) {
      auto it = b.find(key);
      if (it == b.end()) { log.os() << "WARNING (compose-transforms) Second table does not have key " << key << "\n"; continue; }
      vbio::Matrix c;
      if (!vbfmllr::ComposeTransforms(a.data, a.rows, a.cols, it->second.data, it->second.rows, it->second.cols, b_is_affine, &c.data, &c.rows, &c.cols)) {
        log.os() << "WARNING (compose-transforms) Empty matrix in ComposeTransforms\n";
        continue;
      }
      writer.Write(key, c);
    }
    writer.Close();
    return 0;
  });
}

// gmm-rescore-lattice   (VB/src/gmmbin/gmm-rescore-lattice.cpp; K/gmmbin/gmm-rescore-lattice.cc:28-115).  The scores come
// from the GPU in the on-demand form (vbh_gmm_loglikes_pairs): for a batch of utterances, the (frame, pdf) pairs their
// lattices hold are collected, scored in one call and added to the lattices' acoustic costs.
int GmmRescoreLattice(int argc, char *argv[], std::ofstream &file_log) {
  return Guard("gmm-rescore-lattice", &file_log, [&](Log &log) -> int {
    Options po;
    po.Know({"old-acoustic-scale"});
    FrontEnd::Know(&po);
    po.Parse(argc, argv);
    if (po.args.size() != 4) throw ApiError("Wrong arguments.");
    Laps lap("gmm-rescore-lattice");
    const float old_acoustic_scale = (float)po.Num("old-acoustic-scale", 0.0);
    vbio::Model model;
    vbio::ReadModel(po.args[0], &model);
    const int D = model.dim;
    FrontEnd fe;
    fe.Parse(po);
    const RawTable &tab = CachedRawTable(po.args[2]);
    vbh_ctx *ctx = SharedCtx();
    (void)ctx;
    vbh_gmm *dgmm = CachedDeviceGmm(po.args[0], model);      // (the thread's cached device form: no device allocation per call)
    lap("model + table + device model");
    vbio::LatticeReader reader(po.args[1]);
    vbio::CompactLatticeWriter writer;
    writer.Open(po.args[3]);
    int num_done = 0, num_err = 0;
    long long num_frames = 0;
    // Per lattice: conversion to the compact form, state times, the pdfs it holds at every frame (host threads); then ONE call of
    // the on-demand kernel for the (frame, pdf) pairs of all lattices over the table's features on the device; then the scores are
    // added (host threads) and the lattices written in table order.
    struct Item {
      std::string key; const char *bytes = nullptr; size_t nbytes = 0; std::string entry; int utt = -1, rows = 0;
      std::vector<int64_t> off; std::vector<int32_t> pdfs;      // per frame [off[t], off[t+1]) into pdfs (sorted, unique)
      int64_t base = 0; std::string warn, err; bool ok = false;
    };
    std::vector<Item> items;
    { Item it; while (reader.NextBytes(&it.key, &it.bytes, &it.nbytes)) { items.push_back(std::move(it)); it = Item(); } }
    lap("lattice table read");
    auto parallel_for = [&](size_t n, const std::function<void(size_t)> &fn) {
      std::atomic<size_t> next{0};
      auto work = [&]() { for (size_t i = next++; i < n; i = next++) fn(i); };
      const unsigned nt = std::max(1u, std::min<unsigned>(16, std::min<unsigned>((unsigned)n, std::thread::hardware_concurrency())));
      std::vector<std::thread> pool;
      for (unsigned t = 1; t < nt; t++) pool.emplace_back(work);
      work();
      for (std::thread &t : pool) t.join();
    };
    // ---- pass 1 (host threads): which (frame, pdf) pairs a lattice holds.  For the state-level lattices DecodeFmllr hands over (the main
    //      search's raw lattices: 9 k states / 15 k arcs per 30 s) these are read off the arcs as parsed -- an arc's transition-id at its
    //      source state's time -- the same set the compact form's strings give (RescoreCompactLattice's time_to_state walk,
    //      lattice-functions.cc:1236-1270); nothing of the lattice is kept: pass 2 parses the entry again (0.2 ms) and converts it there.
    //      Keeping a thousand converted lattices alive between the passes grew sixteen threads' heaps page by page at once: 3.4 s of
    //      system time in 0.3 s of wall clock, ten times what the conversion itself takes (round 5, VBH_DEBUG_TIMES laps).
    parallel_for(items.size(), [&](size_t i) {
      Item &it = items[i];
      try {
        auto fi = tab.index.find(it.key);
        if (fi == tab.index.end() || tab.rows((size_t)fi->second) == 0) { it.warn = "No feature found for utterance " + it.key + ". Skipping"; return; }
        it.utt = fi->second; it.rows = tab.rows((size_t)it.utt);
        std::vector<uint64_t> keys;      // (frame << 32 | pdf) of every transition-id, then sorted and made unique
        bool ok = true;
        auto add1 = [&](int32_t t, int32_t tid) {
          if (tid <= 0 || tid >= (int32_t)model.tid2pdf.size()) throw ApiError("lattice: transition-id out of the model's range");
          if (t >= it.rows) { ok = false; return; }
          keys.push_back(((uint64_t)(uint32_t)t << 32) | (uint32_t)model.tid2pdf[tid]);
        };
        int32_t utt_len = 0;
        vblat::Lattice lat; vblat::CompactLattice clat; bool compact = false;
        vbio::ParseLatticeBytes(it.bytes, it.nbytes, &lat, &clat, &compact);
        if (!compact) {
          if (lat.start < 0 || lat.st.empty()) { it.warn = "Rescoring empty lattice for utterance " + it.key; return; }
          // state times by a walk from the start state (every path to a state holds the same number of transition-ids)
          const int32_t n = (int32_t)lat.st.size();
          std::vector<int32_t> times((size_t)n, -1), stack(1, lat.start);
          times[(size_t)lat.start] = 0;
          keys.reserve(lat.NumArcs() + 16);
          while (!stack.empty()) {
            const int32_t s = stack.back(); stack.pop_back();
            const int32_t t = times[(size_t)s];
            for (const vblat::Arc &a : lat.st[(size_t)s].arcs) {
              if (a.ilabel != 0) add1(t, a.ilabel);
              if (times[(size_t)a.next] < 0) { times[(size_t)a.next] = t + (a.ilabel != 0 ? 1 : 0); stack.push_back(a.next); }
            }
            if (!vblat::IsZero(lat.st[(size_t)s].final)) utt_len = std::max(utt_len, t);
            if (!ok) break;
          }
        } else {
          if (clat.st.empty()) { it.warn = "Rescoring empty lattice for utterance " + it.key; return; }
          if (!vblat::TopSort(&clat)) { it.warn = "Cycles detected in lattice " + it.key; return; }
          std::vector<int32_t> times;
          utt_len = vblat::CompactLatticeStateTimes(clat, &times);
          keys.reserve(2 * clat.NumArcs() + 16);
          auto add = [&](int32_t t0, const vblat::LabelStr &str) { for (size_t o = 0; o < str.size() && ok; o++) add1(t0 + (int32_t)o, str[o]); };
          for (size_t s = 0; s < clat.st.size() && ok; s++) {
            for (const vblat::CArc &a : clat.st[s].arcs) add(times[s], a.w.str);
            add(times[s], clat.st[s].final.str);
          }
        }
        if (utt_len > it.rows) { it.warn = "Features are too short for lattice: utt-len is " + std::to_string(utt_len) + " for utterance " + it.key; return; }
        if (!ok) { it.warn = "lattice/feature mismatch for utterance " + it.key; return; }
        std::sort(keys.begin(), keys.end());
        keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
        it.off.assign((size_t)it.rows + 1, 0);
        it.pdfs.resize(keys.size());
        for (size_t k = 0; k < keys.size(); k++) { it.pdfs[k] = (int32_t)(uint32_t)keys[k]; it.off[(size_t)(keys[k] >> 32) + 1]++; }
        for (int t = 0; t < it.rows; t++) it.off[(size_t)t + 1] += it.off[(size_t)t];
        it.ok = true;
      } catch (const std::exception &e) { it.err = e.what(); if (it.err.empty()) it.err = "error"; }
    });
    for (const Item &it : items) if (!it.err.empty()) throw ApiError(it.err);
    lap("pass 1: parse + pdf lists");
    // ---- the pairs of every frame of the table, in table order (frames of utterances without a usable lattice: none)
    std::vector<int> item_of(tab.keys.size(), -1);
    for (size_t i = 0; i < items.size(); i++) if (items[i].ok && item_of[(size_t)items[i].utt] < 0) item_of[(size_t)items[i].utt] = (int)i;
    std::vector<int64_t> poff{0}; std::vector<int32_t> ppdf;
    poff.reserve((size_t)tab.frames() + 1);
    for (size_t u = 0; u < tab.keys.size(); u++) {
      const int rows = tab.rows(u);
      if (item_of[u] < 0) { for (int t = 0; t < rows; t++) poff.push_back((int64_t)ppdf.size()); continue; }
      Item &it = items[(size_t)item_of[u]];
      it.base = (int64_t)ppdf.size();
      for (int t = 0; t < rows; t++) poff.push_back(it.base + it.off[t + 1]);
      ppdf.insert(ppdf.end(), it.pdfs.begin(), it.pdfs.end());
    }
    std::vector<float> scores(ppdf.size());
    lap("pair table");
    if (!ppdf.empty()) {
      int fdim = 0;
      std::unique_ptr<DevBuffer> hold;
      const float *d_feats = DeviceFeatures(fe, tab, 0, tab.keys.size(), &fdim, &hold);
      if (fdim != D) throw ApiError("Dim mismatch: data dim = " + std::to_string(fdim) + " vs. model dim = " + std::to_string(D));
      if (vbh_gmm_loglikes_pairs(dgmm, d_feats, tab.frames(), poff.data(), ppdf.data(), scores.data()) != 0)
        throw ApiError(std::string("vbh_gmm_loglikes_pairs: ") + vbh_last_error());
    }
    lap("features + scores on the device");
    // ---- pass 2 (host threads): the entry parsed again, converted to the compact form (CompactLatticeHolder::Read of a state-level
    //      lattice), old scores scaled, sorted, rescored, serialised -- every temporary freed before the thread's next lattice
    parallel_for(items.size(), [&](size_t k) {
      Item &it = items[k];
      if (!it.ok) return;
      const Item &src = items[(size_t)item_of[(size_t)it.utt]];       // (a key met twice in the lattice table shares the first one's scores)
      auto loglike = [&](int32_t t, int32_t tid) -> float {
        const int32_t pdf = model.tid2pdf[tid];
        const int32_t *a = src.pdfs.data() + src.off[t], *b = src.pdfs.data() + src.off[t + 1];
        const int32_t *f = std::lower_bound(a, b, pdf);
        if (f == b || *f != pdf) throw ApiError("internal: pdf of a lattice arc was not scored");
        return scores[src.base + (f - src.pdfs.data())];
      };
      try {
        vblat::CompactLattice clat;
        {
          vblat::Lattice lat; bool compact = false;
          vbio::ParseLatticeBytes(it.bytes, it.nbytes, &lat, &clat, &compact);
          if (!compact) vblat::ConvertLattice(lat, &clat);
        }
        if (old_acoustic_scale != 1.0f) vblat::ScaleLattice(1.0, 0.0, 0.0, (double)old_acoustic_scale, &clat);
        if (clat.st.empty()) { it.ok = false; it.warn = "Rescoring empty lattice for utterance " + it.key; return; }
        if (!vblat::TopSort(&clat)) { it.ok = false; it.warn = "Cycles detected in lattice " + it.key; return; }
        if (!vblat::RescoreCompactLattice(loglike, it.rows, &clat)) { it.ok = false; it.warn = "lattice/feature mismatch for utterance " + it.key; return; }
        vbio::SerializeLattice(clat, &it.entry);
      } catch (const std::exception &e) { it.err = e.what(); }
    });
    lap("pass 2: parse + convert + rescore + serialize");
    for (Item &it : items) {
      if (!it.err.empty()) throw ApiError(it.err);
      if (!it.warn.empty()) { log.os() << "WARNING (gmm-rescore-lattice) " << it.warn << "\n"; num_err++; continue; }
      writer.WriteSerialized(it.key, std::move(it.entry));
      num_done++; num_frames += it.rows;
      std::string().swap(it.entry);
    }
    lap("write");
    log.os() << "LOG (gmm-rescore-lattice) Done " << num_done << " lattices with errors on " << num_err << ", #frames is " << num_frames << "\n";
    return num_done != 0 ? 0 : 1;
  });
}

extern "C" {
int vb_LatticeToPost(int argc, char *argv[], const char *lp) { return WithLog(lp, [&](std::ofstream &l) { return LatticeToPost(argc, argv, l); }); }
int vb_WeightSilencePost(int argc, char *argv[], const char *lp) { return WithLog(lp, [&](std::ofstream &l) { return WeightSilencePost(argc, argv, l); }); }
int vb_GmmPostToGpost(int argc, char *argv[], const char *lp) { return WithLog(lp, [&](std::ofstream &l) { return GmmPostToGpost(argc, argv, l); }); }
int vb_GmmEstFmllrGpost(int argc, char *argv[], const char *lp) { return WithLog(lp, [&](std::ofstream &l) { return GmmEstFmllrGpost(argc, argv, l); }); }
int vb_GmmEstFmllr(int argc, char *argv[], const char *lp) { return WithLog(lp, [&](std::ofstream &l) { return GmmEstFmllr(argc, argv, l); }); }
int vb_ComposeTransforms(int argc, char *argv[], const char *lp) { return WithLog(lp, [&](std::ofstream &l) { return ComposeTransforms(argc, argv, l); }); }
int vb_GmmRescoreLattice(int argc, char *argv[], const char *lp) { return WithLog(lp, [&](std::ofstream &l) { return GmmRescoreLattice(argc, argv, l); }); }
}
