// voicebridge_alignbin.cc -- GmmAlignCompiled: Viterbi alignment of utterances against their own compiled graphs.
//
// Replaces (paths relative to /root/reference/):
//   VoiceBridge/VoiceBridge/kaldi-win/src/gmmbin/gmm-align-compiled.cpp:24-160   the program (argv grammar, tables, log lines)
//   kaldi-master/src/decoder/decoder-wrappers.cc:424-516                        AlignUtteranceWrapper: FasterDecoder with
//       `beam`, a second attempt with `retry_beam` when no final state was reached, best path -> alignment, score
//   kaldi-master/src/hmm/hmm-utils.cc:1019-1073                                 AddTransitionProbs on the utterance's graph
//   kaldi-master/src/lat/lattice-functions.cc GetPerFrameAcousticCosts           (--write-per-frame-acoustic-loglikes)
// This is where the VoiceBridge DLL runs its 1-best search (FasterDecoder): step AlignSi (scr/kaldi_scr.h:226) -> gmm-align-compiled.
//
// MI355X mapping.  The reference decodes one utterance after the other, each against a graph of a few thousand states.  Here a
// batch of utterances is ONE launch of the persistent FasterDecoder grid (csrc/vbh_faster.hip): the batch's graphs are laid side
// by side as one union graph on the device (state ids shifted, one start state per utterance), the all-pdf GMM kernel scores the
// batch's frames once, and every workgroup searches its utterance inside its own part of the union.  Utterances that do not reach
// a final state are searched again, together, with retry_beam.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <limits>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "api_common.h"
#include "kaldi_io.h"
#include "voicebridge_api.h"
#include "voicebridge_hip.h"

using namespace vbapi;

namespace {

void Check(int rc, const char *what) {
  if (rc != 0) throw ApiError(std::string(what) + ": " + vbh_last_error());
}

struct DevMem {   // a per-call work buffer: scratch slot `slot` of the context (vbh_ctx_scratch: nothing is allocated or freed per call once it has its size)
  vbh_ctx *ctx; void *p = nullptr;
  DevMem(vbh_ctx *c, size_t bytes, int slot) : ctx(c) { Check(vbh_ctx_scratch(c, slot, bytes, &p), "vbh_ctx_scratch"); }
  DevMem(const DevMem &) = delete;
  DevMem &operator=(const DevMem &) = delete;
};

struct AlignUtt {
  std::string key;
  vbio::Graph graph;          // transition probabilities added
  const vbio::Matrix *feats;
};

struct AlignResult {
  bool reached_final = false;
  std::vector<int32_t> ali;
  float graph = 0.0f, acoustic = 0.0f;
  std::vector<float> per_frame_acoustic;      // acoustic cost of every arc with a transition-id, in path order
};

// one launch of the batched FasterDecoder over `utts` (indices into `all`) with `beam`
void AlignBatch(vbh_ctx *ctx, vbh_gmm *gmm, const vbio::Model &model, const std::vector<AlignUtt> &all, const std::vector<int> &utts, float beam,
                float acoustic_scale, std::vector<AlignResult> *out) {
  // ---- union graph: the utterances' graphs side by side
  std::vector<float> fin; std::vector<int64_t> aoff(1, 0); std::vector<vbh_arc> arcs; std::vector<int32_t> starts; std::vector<int64_t> foff(1, 0);
  std::vector<float> data;
  int32_t max_states = 0, max_T = 0;
  for (int u : utts) {
    const vbio::Graph &g = all[u].graph;
    const int32_t base = (int32_t)fin.size(), S = (int32_t)g.final_cost.size();
    if ((int64_t)base + S > 0x3FFFFFFFLL) throw ApiError("alignment batch: the union of the graphs has too many states; use a smaller --decode-batch");
    starts.push_back(base + g.start);
    fin.insert(fin.end(), g.final_cost.begin(), g.final_cost.end());
    const int64_t abase = (int64_t)arcs.size();
    for (const vbio::Arc &a : g.arcs) arcs.push_back(vbh_arc{a.ilabel, a.olabel, a.weight, a.nextstate + base});
    for (int32_t s = 1; s <= S; s++) aoff.push_back(abase + g.arc_offset[s]);
    max_states = std::max(max_states, S);
    const vbio::Matrix &m = *all[u].feats;
    data.insert(data.end(), m.data.begin(), m.data.end());
    foff.push_back(foff.back() + m.rows);
    max_T = std::max(max_T, m.rows);
  }
  const int64_t T = foff.back();
  const int32_t ld = vbh_gmm_padded_pdfs(gmm);       // score rows on 128-byte lines; the search takes ld as its row length
  DevMem d_in(ctx, sizeof(float) * data.size(), 1), d_ll(ctx, sizeof(float) * (size_t)T * ld, 5);
  Check(vbh_copy_h2d(ctx, d_in.p, data.data(), sizeof(float) * data.size()), "vbh_copy_h2d");
  Check(vbh_gmm_loglikes_ld(gmm, (const float *)d_in.p, T, (float *)d_ll.p, ld), "vbh_gmm_loglikes");
  vbh_graph *g = nullptr;
  Check(vbh_graph_create(ctx, (int32_t)fin.size(), starts[0], fin.data(), aoff.data(), arcs.data(), &g), "vbh_graph_create");
  vbh_faster_cfg cfg;
  vbh_faster_cfg_default(&cfg);          // FasterDecoderOptions defaults (faster-decoder.h:38-43); AlignUtteranceWrapper sets the beam only (:457-459)
  cfg.beam = beam;
  cfg.acoustic_scale = acoustic_scale;
  vbh_faster_decoder *dec = nullptr;
  if (vbh_faster_decoder_create_ex(ctx, g, &cfg, (int32_t)utts.size(), max_T, max_states, &dec) != 0) {
    const std::string e = vbh_last_error(); vbh_graph_destroy(g); throw ApiError("vbh_faster_decoder_create: " + e);
  }
  const int rc = vbh_faster_decode_batch_ex(dec, (const float *)d_ll.p, ld, foff.data(), (int32_t)utts.size(), model.tid2pdf.data(),
                                            (int32_t)model.tid2pdf.size() - 1, starts.data(), 0.0f);
  if (rc != 0) { const std::string e = vbh_last_error(); vbh_faster_decoder_destroy(dec); vbh_graph_destroy(g); throw ApiError("vbh_faster_decode_batch: " + e); }
  for (size_t i = 0; i < utts.size(); i++) {
    vbh_best_path bp;
    Check(vbh_faster_decoder_result(dec, (int32_t)i, &bp), "vbh_faster_decoder_result");
    AlignResult &r = (*out)[utts[i]];
    r = AlignResult();
    r.reached_final = bp.ok && bp.reached_final;       // decoder.ReachedFinal(): "consider only final states" (:463)
    if (!r.reached_final) continue;
    r.ali.assign(bp.ali, bp.ali + bp.num_ali);
    r.graph = bp.best_graph; r.acoustic = bp.best_acoustic;
    for (int a = 0; a < bp.num_arcs; a++) if (bp.arc_ilabel[a] != 0) r.per_frame_acoustic.push_back(bp.arc_acoustic[a]);
  }
  vbh_faster_decoder_destroy(dec);
  vbh_graph_destroy(g);
}

}  // namespace

int GmmAlignCompiled(int argc, char *argv[], std::ofstream &file_log) {
  return Guard("gmm-align-compiled", &file_log, [&](Log &log) -> int {
    Options po;
    po.Know({"beam", "retry-beam", "careful", "transition-scale", "acoustic-scale", "self-loop-scale", "write-per-frame-acoustic-loglikes", "decode-batch"});
    po.Parse(argc, argv);
    if (po.args.size() < 4 || po.args.size() > 5) throw ApiError("wrong arguments.");
    // AlignConfig (K/decoder/decoder-wrappers.h:36-52) and the program's own options (:47-58)
    const float beam = (float)po.Num("beam", 200.0), retry_beam = (float)po.Num("retry-beam", 0.0);
    const bool careful = po.Bool("careful", false);
    const float transition_scale = (float)po.Num("transition-scale", 1.0), acoustic_scale = (float)po.Num("acoustic-scale", 1.0),
                self_loop_scale = (float)po.Num("self-loop-scale", 1.0);
    if ((retry_beam != 0 && retry_beam <= beam) || beam <= 0.0)
      throw ApiError("Beams do not make sense: beam " + std::to_string(beam) + ", retry-beam " + std::to_string(retry_beam));
    if (careful) throw ApiError("--careful=true (ModifyGraphForCarefulAlignment) is not part of this build");
    vbio::Model model;
    vbio::ReadModel(po.args[0], &model);
    const int32_t num_tids = (int32_t)model.tid2pdf.size() - 1;
    vbio::FstTableReader fst_reader(po.args[1]);
    const std::map<std::string, vbio::Matrix> features = vbio::ReadMatrixTable(po.args[2]);      // (the reference reads them by key: RandomAccessBaseFloatMatrixReader)
    vbio::IntVectorWriter alignment_writer(po.args[3]);
    vbio::FloatWriter scores_writer;
    if (po.args.size() == 5 && !po.args[4].empty()) scores_writer.Open(po.args[4]);
    vbio::FloatVectorWriter per_frame_writer;
    if (po.Has("write-per-frame-acoustic-loglikes") && !po.Str("write-per-frame-acoustic-loglikes", "").empty())
      per_frame_writer.Open(po.Str("write-per-frame-acoustic-loglikes", ""));
    vbh_ctx *ctx = SharedCtx();
    vbh_gmm *gmm = nullptr;
    Check(vbh_gmm_create(ctx, model.dim, model.num_pdfs, model.pdf_offsets.data(), model.weights.data(), model.means_invvars.data(),
                         model.inv_vars.data(), &gmm), "vbh_gmm_create");
    struct GmmGuard { vbh_gmm *g; ~GmmGuard() { vbh_gmm_destroy(g); } } gmm_guard{gmm};
    const int batch_utts = std::max(1, (int)po.Num("decode-batch", 256));
    int num_done = 0, num_err = 0, num_retry = 0;
    double tot_like = 0.0;
    int64_t frame_count = 0;
    bool more = true;
    while (more) {
      std::vector<AlignUtt> batch;
      while ((int)batch.size() < batch_utts) {
        AlignUtt au;
        if (!(more = fst_reader.Next(&au.key, &au.graph))) break;
        const auto it = features.find(au.key);
        if (it == features.end()) { num_err++; log.os() << "WARNING (gmm-align-compiled) No features for utterance " << au.key << "\n"; continue; }
        if (it->second.rows == 0) { num_err++; log.os() << "WARNING (gmm-align-compiled) Zero-length utterance: " << au.key << "\n"; continue; }
        if (it->second.cols != model.dim) throw ApiError("Dim mismatch: data dim = " + std::to_string(it->second.cols) + " vs. model dim = " + std::to_string(model.dim));
        if (au.graph.start < 0) {       // fst->Start() == kNoStateId (:445-449)
          num_err++; log.os() << "WARNING (gmm-align-compiled) Empty decoding graph for " << au.key << "\n"; continue;
        }
        // AddTransitionProbs (hmm-utils.cc:1042-1073), no disambiguation symbols
        for (vbio::Arc &a : au.graph.arcs) {
          if (a.ilabel >= 1 && a.ilabel <= num_tids) a.weight = a.weight + -vbio::ScaledTransitionLogProb(model, a.ilabel, transition_scale, self_loop_scale);
          else if (a.ilabel != 0) throw ApiError("AddTransitionProbs: invalid symbol " + std::to_string(a.ilabel) + " on graph input side.");
        }
        au.feats = &it->second;
        batch.push_back(std::move(au));
      }
      if (batch.empty()) continue;
      std::vector<AlignResult> res(batch.size());
      std::vector<int> idx(batch.size());
      for (size_t i = 0; i < batch.size(); i++) idx[i] = (int)i;
      AlignBatch(ctx, gmm, model, batch, idx, beam, acoustic_scale, &res);
      if (retry_beam != 0.0f) {            // (:466-473)
        std::vector<int> again;
        for (size_t i = 0; i < batch.size(); i++) if (!res[i].reached_final) again.push_back((int)i);
        for (int i : again) {
          num_retry++;
          log.os() << "WARNING (gmm-align-compiled) Retrying utterance " << batch[i].key << " with beam " << retry_beam << "\n";
        }
        if (!again.empty()) AlignBatch(ctx, gmm, model, batch, again, retry_beam, acoustic_scale, &res);
      }
      for (size_t i = 0; i < batch.size(); i++) {
        const AlignUtt &au = batch[i];
        const AlignResult &r = res[i];
        log.os() << "LOG (gmm-align-compiled) " << au.key << "\n";
        if (!r.reached_final) {
          num_err++;
          log.os() << "WARNING (gmm-align-compiled) Did not successfully decode file " << au.key << ", len = " << au.feats->rows << "\n";
          continue;
        }
        const float like = -(r.graph + r.acoustic) / acoustic_scale;
        num_done++;
        tot_like += like;
        frame_count += au.feats->rows;
        alignment_writer.Write(au.key, r.ali);
        if (scores_writer.IsOpen()) scores_writer.Write(au.key, -(r.graph + r.acoustic));
        if (per_frame_writer.IsOpen()) {
          std::vector<float> v(r.per_frame_acoustic);
          for (float &x : v) x *= -1 / acoustic_scale;
          per_frame_writer.Write(au.key, v);
        }
      }
    }
    log.os() << "Overall log-likelihood per frame is " << (tot_like / frame_count) << " over " << frame_count << " frames." << "\n";
    log.os() << "Retried " << num_retry << " out of " << (num_done + num_err) << " utterances." << "\n";
    log.os() << "Done " << num_done << ", errors on " << num_err << "\n";
    return num_done != 0 ? 0 : 1;
  });
}

extern "C" int vb_GmmAlignCompiled(int argc, char *argv[], const char *lp) {
  return WithLog(lp, [&](std::ofstream &l) { return GmmAlignCompiled(argc, argv, l); });
}
