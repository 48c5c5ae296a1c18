// voicebridge_latbin.cc -- the lattice programs that follow the search in the reference's Decode
// (SURVEY.md 8f rank 1 and 3): determinization, scaling, word insertion penalty, best path, WER.
// Declarations and the reference files they mirror: voicebridge_api.h.  Host code: these run per
// utterance on lattices of a few thousand arcs, between the GPU search and the files.
#include <atomic>
#include <cmath>
#include <map>
#include <sstream>
#include <thread>

#include "api_common.h"
#include "kaldi_io.h"
#include "lattice.h"
#include "voicebridge_api.h"

using namespace vbapi;

namespace {

vblat::DetOptions ReadDetOptions(const Options &po, int default_max_mem, int default_max_loop) {
  vblat::DetOptions o;
  o.delta = (float)po.Num("delta", vblat::kDelta);
  o.max_mem = (int)po.Num("max-mem", default_max_mem);
  o.max_loop = (int)po.Num("max-loop", default_max_loop);
  o.max_states = (int)po.Num("max-states", -1);
  o.max_arcs = (int)po.Num("max-arcs", -1);
  o.retry_cutoff = (float)po.Num("retry-cutoff", 0.5);
  return o;
}

// CompactLatticeDepth (lat/lattice-functions.cc): arcs' and final strings' transition-ids per frame
double CompactLatticeDepth(const vblat::CompactLattice &clat, int32_t *num_frames) {
  if (clat.start < 0) { *num_frames = 0; return 1.0; }
  size_t num_arc_frames = 0;
  // frames = length of any path: follow first arcs from the start (the lattice is top-sorted)
  int32_t t = 0;
  {
    int32_t s = clat.start;
    for (;;) {
      const vblat::CState &st = clat.st[s];
      if (!vblat::IsZero(st.final.w) && st.arcs.empty()) { t += (int32_t)st.final.str.size(); break; }
      if (st.arcs.empty()) break;
      t += (int32_t)st.arcs[0].w.str.size();
      s = st.arcs[0].next;
    }
  }
  for (const vblat::CState &st : clat.st) {
    for (const vblat::CArc &a : st.arcs) num_arc_frames += a.w.str.size();
    num_arc_frames += st.final.str.size();
  }
  *num_frames = t;
  return t == 0 ? 1.0 : (double)num_arc_frames / (double)t;
}

void TopSortCompactIfNeeded(vblat::CompactLattice *clat) {
  bool sorted = true;
  for (size_t s = 0; s < clat->st.size() && sorted; s++) for (const vblat::CArc &a : clat->st[s].arcs) if (a.next <= (int32_t)s) { sorted = false; break; }
  if (!sorted && !vblat::TopSort(clat)) throw ApiError("Topological sorting failed");
}

}  // namespace

// lattice-determinize-pruned   (VB/src/latbin/lattice-determinize-pruned.cpp:33-160)
int LatticeDeterminizePruned(int argc, char *argv[], std::ofstream &file_log) {
  return Guard("lattice-determinize-pruned", &file_log, [&](Log &log) -> int {
    Options po;
    po.Know({"acoustic-scale", "beam", "minimize", "delta", "max-mem", "max-arcs", "max-states", "max-loop", "retry-cutoff"});
    po.Parse(argc, argv);
    if (po.args.size() != 2) throw ApiError("Wrong arguments.");
    const float acoustic_scale = (float)po.Num("acoustic-scale", 1.0), beam = (float)po.Num("beam", 10.0);
    const bool minimize = po.Bool("minimize", false);
    const vblat::DetOptions opts = ReadDetOptions(po, 50000000, 0);
    if (acoustic_scale == 0.0f) throw ApiError("Do not use a zero acoustic scale (cannot be inverted)");
    vbio::LatticeReader reader(po.args[0]);
    vbio::CompactLatticeWriter writer;
    writer.Open(po.args[1]);
    int n_done = 0, n_warn = 0;
    double sum_depth_in = 0.0, sum_depth_out = 0.0, sum_t = 0.0;
    // the lattices of the table are independent: read, determinized on the host's threads, written in table order
    struct Item { std::string key; const char *bytes = nullptr; size_t nbytes = 0; vblat::Lattice lat; vblat::CompactLattice det; std::string entry;
                  bool ok = true, empty = false, unsorted = false; double depth = 0; int32_t t = 0; size_t in_states = 0; std::string err; };
    std::vector<Item> items;
    Laps lap("lattice-determinize-pruned");
    { Item it; while (reader.NextBytes(&it.key, &it.bytes, &it.nbytes)) { items.push_back(std::move(it)); it = Item(); } }
    lap("table read");
    {
      std::atomic<size_t> next{0};
      auto work = [&]() {
        for (size_t i = next++; i < items.size(); i = next++) {
          Item &it = items[i];
          try {
            vbio::ParseLatticeBytes(it.bytes, it.nbytes, &it.lat);
            vblat::Invert(&it.lat);
            vblat::ScaleLattice(1.0, 0.0, 0.0, (double)acoustic_scale, &it.lat);
            it.unsorted = !vblat::TopSort(&it.lat);
            vblat::ArcSortILabel(&it.lat);
            it.ok = vblat::DeterminizeLatticePruned(it.lat, beam, &it.det, opts);
            vblat::Connect(&it.det);
            it.empty = it.det.st.empty();
            if (minimize) {             // K/latbin/lattice-determinize-pruned.cc:111-115
              vblat::PushCompactLatticeStrings(&it.det);
              vblat::PushCompactLatticeWeights(&it.det);
              vblat::MinimizeCompactLattice(&it.det);
            }
            TopSortCompactIfNeeded(&it.det);
            it.depth = CompactLatticeDepth(it.det, &it.t);
            it.in_states = it.lat.st.size();
            vblat::ScaleLattice(1.0, 0.0, 0.0, 1.0 / (double)acoustic_scale, &it.det);
            vblat::Lattice().st.swap(it.lat.st);
            vbio::SerializeLattice(it.det, &it.entry);
            vblat::CompactLattice().st.swap(it.det.st);
          } catch (const std::exception &e) { it.err = e.what(); if (it.err.empty()) it.err = "error"; }
        }
      };
      const unsigned nt = std::max(1u, std::min<unsigned>(16, std::min<unsigned>((unsigned)items.size(), std::thread::hardware_concurrency())));
      std::vector<std::thread> pool;
      for (unsigned t = 1; t < nt; t++) pool.emplace_back(work);
      work();
      for (std::thread &t : pool) t.join();
    }
    lap("parse + determinize + serialize");
    for (Item &it : items) {
      if (!it.err.empty()) throw ApiError("key " + it.key + ": " + it.err);
      if (it.unsorted) log.os() << "WARNING (lattice-determinize-pruned) Could not topologically sort lattice: this probably means it has bad properties e.g. epsilon cycles.\n";
      if (!it.ok) { log.os() << "WARNING (lattice-determinize-pruned) For key " << it.key << ", determinization did not succeed(partial output will be pruned tighter than the specified beam.)\n"; n_warn++; }
      if (it.empty) { log.os() << "WARNING (lattice-determinize-pruned) For key " << it.key << ", determinized and trimmed lattice was empty.\n"; n_warn++; }
      sum_depth_in += (double)it.in_states; sum_depth_out += it.depth * it.t; sum_t += it.t;
      writer.WriteSerialized(it.key, std::move(it.entry));
      std::string().swap(it.entry);
      n_done++;
    }
    if (sum_t != 0.0)
      log.os() << "LOG (lattice-determinize-pruned) Average input-lattice depth (measured at at state level) is " << (sum_depth_in / sum_t)
               << ", output depth is " << (sum_depth_out / sum_t) << ", over " << sum_t << " frames  (average num-frames = " << (sum_t / n_done) << ").\n";
    log.os() << "LOG (lattice-determinize-pruned) Done " << n_done << " lattices, determinization finished earlier than specified by the beam (or output was empty) on "
             << n_warn << " of these.\n";
    return n_done != 0 ? 0 : 1;
  });
}

// lattice-determinize-phone-pruned   (upstream K/latbin/lattice-determinize-phone-pruned.cc:30-130;
// the same two-pass determinization gmm-latgen-faster applies to every utterance)
int LatticeDeterminizePhonePruned(int argc, char *argv[], std::ofstream &file_log) {
  return Guard("lattice-determinize-phone-pruned", &file_log, [&](Log &log) -> int {
    Options po;
    po.Know({"acoustic-scale", "beam", "delta", "max-mem", "phone-determinize", "word-determinize", "minimize"});
    po.Parse(argc, argv);
    if (po.args.size() != 3) throw ApiError("Wrong arguments.");
    const float acoustic_scale = (float)po.Num("acoustic-scale", 1.0), beam = (float)po.Num("beam", 10.0);
    vblat::PhoneDetOptions opts;
    opts.delta = (float)po.Num("delta", vblat::kDelta);
    opts.max_mem = (int)po.Num("max-mem", 50000000);
    opts.phone_determinize = po.Bool("phone-determinize", true);
    opts.word_determinize = po.Bool("word-determinize", true);
    opts.minimize = po.Bool("minimize", false);
    if (acoustic_scale == 0.0f) throw ApiError("Do not use a zero acoustic scale (cannot be inverted)");
    vbio::Model model;
    vbio::ReadModel(po.args[0], &model);
    vbio::LatticeReader reader(po.args[1]);
    vbio::CompactLatticeWriter writer;
    writer.Open(po.args[2]);
    int n_done = 0, n_warn = 0;
    std::string key; vblat::Lattice lat;
    while (reader.NextLattice(&key, &lat)) {
      vblat::ScaleLattice(1.0, 0.0, 0.0, (double)acoustic_scale, &lat);
      vblat::CompactLattice det;
      if (!vblat::DeterminizeLatticePhonePrunedWrapper(model.trans, &lat, beam, &det, opts)) {
        log.os() << "WARNING (lattice-determinize-phone-pruned) For key " << key << ", determinization did not succeed(partial output will be pruned tighter than the specified beam.)\n";
        n_warn++;
      }
      TopSortCompactIfNeeded(&det);
      vblat::ScaleLattice(1.0, 0.0, 0.0, 1.0 / (double)acoustic_scale, &det);
      writer.Write(key, det);
      n_done++;
    }
    log.os() << "LOG (lattice-determinize-phone-pruned) Done " << n_done << " lattices, determinization finished earlier than specified by the beam on "
             << n_warn << " of these.\n";
    return n_done != 0 ? 0 : 1;
  });
}

// lattice-scale   (VB/src/latbin/lattice-scale.cpp; K/latbin/lattice-scale.cc:28-100)
int LatticeScale(int argc, char *argv[], std::ofstream &file_log) {
  return Guard("lattice-scale", &file_log, [&](Log &log) -> int {
    Options po;
    po.Know({"acoustic-scale", "inv-acoustic-scale", "lm-scale", "acoustic2lm-scale", "lm2acoustic-scale"});
    po.Parse(argc, argv);
    if (po.args.size() != 2) throw ApiError("Wrong arguments.");
    float acoustic_scale = (float)po.Num("acoustic-scale", 1.0);
    const float inv_acoustic_scale = (float)po.Num("inv-acoustic-scale", 1.0), lm_scale = (float)po.Num("lm-scale", 1.0),
                a2l = (float)po.Num("acoustic2lm-scale", 0.0), l2a = (float)po.Num("lm2acoustic-scale", 0.0);
    if (!(acoustic_scale == 1.0f || inv_acoustic_scale == 1.0f)) throw ApiError("only one of --acoustic-scale and --inv-acoustic-scale may be set");
    if (inv_acoustic_scale != 1.0f) acoustic_scale = (float)(1.0 / inv_acoustic_scale);
    vbio::LatticeReader reader(po.args[0]);
    vbio::CompactLatticeWriter writer;
    writer.Open(po.args[1]);
    int n_done = 0;
    std::string key; vblat::CompactLattice clat;
    while (reader.NextCompact(&key, &clat)) {
      vblat::ScaleLattice((double)lm_scale, (double)a2l, (double)l2a, (double)acoustic_scale, &clat);
      writer.Write(key, clat);
      n_done++;
    }
    log.os() << "LOG (lattice-scale) Done " << n_done << " lattices.\n";
    return n_done != 0 ? 0 : 1;
  });
}

// lattice-add-penalty   (VB/src/latbin/lattice-add-penalty.cpp; K/latbin/lattice-add-penalty.cc:28-75)
int LatticeAddPenalty(int argc, char *argv[], std::ofstream &file_log) {
  return Guard("lattice-add-penalty", &file_log, [&](Log &log) -> int {
    Options po;
    po.Know({"word-ins-penalty"});
    po.Parse(argc, argv);
    if (po.args.size() != 2) throw ApiError("Wrong arguments.");
    const float pen = (float)po.Num("word-ins-penalty", 0.0);
    vbio::LatticeReader reader(po.args[0]);
    vbio::CompactLatticeWriter writer;
    writer.Open(po.args[1]);
    long long n_done = 0;
    std::string key; vblat::CompactLattice clat;
    while (reader.NextCompact(&key, &clat)) {
      vblat::AddWordInsPen(pen, &clat);
      writer.Write(key, clat);
      n_done++;
    }
    log.os() << "LOG (lattice-add-penalty) Done adding word insertion penalty to " << n_done << " lattices.\n";
    return n_done != 0 ? 0 : 1;
  });
}

// lattice-best-path   (VB/src/latbin/lattice-best-path.cpp; K/latbin/lattice-best-path.cc:28-140)
int LatticeBestPath(int argc, char *argv[], std::ofstream &file_log) {
  return Guard("lattice-best-path", &file_log, [&](Log &log) -> int {
    Options po;
    po.Know({"acoustic-scale", "lm-scale", "word-symbol-table"});
    po.Parse(argc, argv);
    if (po.args.size() < 1 || po.args.size() > 3) throw ApiError("Wrong arguments.");
    const float acoustic_scale = (float)po.Num("acoustic-scale", 1.0), lm_scale = (float)po.Num("lm-scale", 1.0);
    vbio::LatticeReader reader(po.args[0]);
    vbio::IntVectorWriter words_writer, ali_writer;
    if (po.args.size() >= 2 && !po.args[1].empty()) words_writer.Open(po.args[1]);
    if (po.args.size() >= 3 && !po.args[2].empty()) ali_writer.Open(po.args[2]);
    int n_done = 0, n_fail = 0;
    long long n_frame = 0;
    vblat::Weight tot = vblat::One();
    std::string key; vblat::CompactLattice clat;
    while (reader.NextCompact(&key, &clat)) {
      vblat::ScaleLattice((double)lm_scale, 0.0, 0.0, (double)acoustic_scale, &clat);
      vblat::CompactLattice best;
      vblat::CompactLatticeShortestPath(clat, &best);
      if (best.start < 0) { log.os() << "WARNING (lattice-best-path) Best-path failed for key " << key << "\n"; n_fail++; continue; }
      std::vector<int32_t> ali, words;
      vblat::Weight w;
      if (!vblat::GetLinearSymbolSequence(best, &ali, &words, &w)) throw ApiError("best path of " + key + " is not linear");
      log.os() << "LOG (lattice-best-path) For utterance " << key << ", best cost " << w.v1 << " + " << w.v2 << " = " << (w.v1 + w.v2)
               << " over " << ali.size() << " frames.\n";
      if (words_writer.IsOpen()) words_writer.Write(key, words);
      if (ali_writer.IsOpen()) ali_writer.Write(key, ali);
      n_done++;
      n_frame += (long long)ali.size();
      tot = vblat::Times(tot, w);
    }
    const float tot_f = tot.v1 + tot.v2;
    log.os() << "LOG (lattice-best-path) Overall cost per frame is " << (tot_f / n_frame) << " = " << (tot.v1 / n_frame) << " [graph] + "
             << (tot.v2 / n_frame) << " [acoustic] over " << n_frame << " frames.\n";
    log.os() << "LOG (lattice-best-path) Done " << n_done << " lattices, failed for " << n_fail << "\n";
    return n_done != 0 ? 0 : 1;
  });
}

// LevenshteinEditDistance with error types (K/util/edit-distance-inl.h:78-127): ties prefer
// substitution/match only when strictly cheapest, then deletion, then insertion
static int EditDistance(const std::vector<std::string> &ref, const std::vector<std::string> &hyp, int *ins, int *del, int *sub) {
  struct E { int ins, del, sub, cost; };
  std::vector<E> e(ref.size() + 1), cur(ref.size() + 1);
  for (size_t i = 0; i < e.size(); i++) e[i] = E{0, (int)i, 0, (int)i};
  for (size_t h = 1; h <= hyp.size(); h++) {
    cur[0] = e[0]; cur[0].ins++; cur[0].cost++;
    for (size_t r = 1; r <= ref.size(); r++) {
      const int ins_err = e[r].cost + 1, del_err = cur[r - 1].cost + 1;
      int sub_err = e[r - 1].cost;
      const bool diff = hyp[h - 1] != ref[r - 1];
      if (diff) sub_err++;
      if (sub_err < ins_err && sub_err < del_err) { cur[r] = e[r - 1]; if (diff) cur[r].sub++; cur[r].cost = sub_err; }
      else if (del_err < ins_err) { cur[r] = cur[r - 1]; cur[r].cost = del_err; cur[r].del++; }
      else { cur[r] = e[r]; cur[r].cost = ins_err; cur[r].ins++; }
    }
    e = cur;
  }
  *ins = e.back().ins; *del = e.back().del; *sub = e.back().sub;
  return e.back().cost;
}

// compute-wer   (VB/src/bin/compute-wer.cpp; K/bin/compute-wer.cc:28-140).  The reference prints
// to stdout; here the three lines go to the log stream and the counters can be fetched with
// vb_ComputeWerCounts (the numbers the multi-GPU gather sums).
namespace { thread_local long long g_wer[8]; }
int ComputeWer(int argc, char *argv[], std::ofstream &file_log) {
  return Guard("compute-wer", &file_log, [&](Log &log) -> int {
    Options po;
    po.Know({"mode", "text"});
    po.Parse(argc, argv);
    // upstream: <ref> <hyp>; the reference's wrapper adds a third argument, the file the counts are
    // written to (VB/src/bin/compute-wer.cpp:65,141-157), which BestWer reads back
    if (po.args.size() != 2 && po.args.size() != 3) throw ApiError("Wrong arguments.");
    const std::string mode = po.Str("mode", "strict");
    if (mode != "strict" && mode != "present" && mode != "all")
      throw ApiError("--mode option invalid: expected \"present\"|\"all\"|\"strict\", got " + mode);
    const auto ref = vbio::ReadTokenTable(po.args[0]);
    std::map<std::string, std::vector<std::string>> hyp;
    for (auto &kv : vbio::ReadTokenTable(po.args[1])) hyp[kv.first] = std::move(kv.second);
    int num_words = 0, word_errs = 0, num_sent = 0, sent_errs = 0, num_ins = 0, num_del = 0, num_sub = 0, num_absent = 0;
    for (const auto &kv : ref) {
      std::vector<std::string> hyp_sent;
      auto it = hyp.find(kv.first);
      if (it == hyp.end()) {
        if (mode == "strict") throw ApiError("No hypothesis for key " + kv.first + " and strict mode specifier.");
        num_absent++;
        if (mode == "present") continue;
      } else hyp_sent = it->second;
      num_words += (int)kv.second.size();
      int ins, del, sub;
      word_errs += EditDistance(kv.second, hyp_sent, &ins, &del, &sub);
      num_ins += ins; num_del += del; num_sub += sub;
      num_sent++;
      sent_errs += kv.second != hyp_sent;
    }
    const float wer = 100.0f * (float)word_errs / (float)num_words, ser = 100.0f * (float)sent_errs / (float)num_sent;
    std::ostringstream o;
    o.precision(2);
    o << "%WER " << std::fixed << wer << " [ " << word_errs << " / " << num_words << ", " << num_ins << " ins, " << num_del << " del, " << num_sub
      << " sub ]" << (num_absent != 0 ? " [PARTIAL]" : "") << '\n';
    o << "%SER " << std::fixed << ser << " [ " << sent_errs << " / " << num_sent << " ]\n";
    o << "Scored " << num_sent << " sentences, " << num_absent << " not present in hyp.\n";
    log.os() << o.str();
    if (po.args.size() == 3 && !po.args[2].empty()) {
      std::ofstream f(po.args[2], std::ios::binary);
      if (!f) throw ApiError("Output file is not accessible " + po.args[2] + ".");
      f << std::fixed;
      f.precision(2);
      f << wer << " " << word_errs << " " << num_words << " " << num_ins << " " << num_del << " " << num_sub << " " << ser << " " << sent_errs << " "
        << num_sent << '\n';
    }
    const long long c[8] = {word_errs, num_words, num_ins, num_del, num_sub, num_sent, sent_errs, num_absent};
    for (int i = 0; i < 8; i++) g_wer[i] = c[i];
    return 0;
  });
}

extern "C" {
int vb_LatticeDeterminizePruned(int argc, char *argv[], const char *lp) { return WithLog(lp, [&](std::ofstream &l) { return LatticeDeterminizePruned(argc, argv, l); }); }
int vb_LatticeDeterminizePhonePruned(int argc, char *argv[], const char *lp) { return WithLog(lp, [&](std::ofstream &l) { return LatticeDeterminizePhonePruned(argc, argv, l); }); }
int vb_LatticeScale(int argc, char *argv[], const char *lp) { return WithLog(lp, [&](std::ofstream &l) { return LatticeScale(argc, argv, l); }); }
int vb_LatticeAddPenalty(int argc, char *argv[], const char *lp) { return WithLog(lp, [&](std::ofstream &l) { return LatticeAddPenalty(argc, argv, l); }); }
int vb_LatticeBestPath(int argc, char *argv[], const char *lp) { return WithLog(lp, [&](std::ofstream &l) { return LatticeBestPath(argc, argv, l); }); }
int vb_ComputeWer(int argc, char *argv[], const char *lp) { return WithLog(lp, [&](std::ofstream &l) { return ComputeWer(argc, argv, l); }); }
// {word_errs, num_words, ins, del, sub, num_sent, sent_errs, absent} of the calling thread's last ComputeWer
void vb_ComputeWerCounts(long long *out8) { for (int i = 0; i < 8; i++) out8[i] = g_wer[i]; }
}
