// voicebridge_api.cc -- argv-style bin functions of the hot path on top of the C ABI.
// See voicebridge_api.h for the reference declarations each one mirrors.  The bodies follow the
// control flow of the reference wrappers (table in -> per-utterance op -> table out, counters,
// return codes) but batch the utterances of a table through the GPU.
#include "voicebridge_api.h"

#include <sys/stat.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <limits>
#include <map>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <sstream>
#include <string>
#include <future>
#include <thread>
#include <vector>

#include "api_common.h"
#include "device_feats.h"
#include "kaldi_io.h"
#include "lattice.h"
#include "voicebridge_hip.h"

using namespace vbapi;

namespace {

thread_local int g_device = 0;
thread_local vbh_ctx *g_ctx = nullptr;
thread_local int g_ctx_device = -1;

void Check(int rc, const char *what) {
  if (rc != 0) throw ApiError(std::string(what) + ": " + vbh_last_error());
}

vbh_ctx *Ctx() {
  if (g_ctx && g_ctx_device == g_device) return g_ctx;
  if (g_ctx) { vbh_ctx_destroy(g_ctx); g_ctx = nullptr; }
  Check(vbh_ctx_create(g_device, &g_ctx), "vbh_ctx_create");
  g_ctx_device = g_device;
  return g_ctx;
}

struct DevMem {   // a per-call work buffer: scratch slot `slot` of the context (vbh_ctx_scratch: nothing is allocated or freed per call once it has its size)
  vbh_ctx *ctx; void *p = nullptr;
  DevMem(vbh_ctx *c, size_t bytes, int slot) : ctx(c) { Check(vbh_ctx_scratch(c, slot, bytes, &p), "vbh_ctx_scratch"); }
  DevMem(const DevMem &) = delete;
  DevMem &operator=(const DevMem &) = delete;
};

struct Batch {   // utterances of a feature table packed row-wise
  std::vector<std::string> keys;
  std::vector<int64_t> offs{0};
  std::vector<float> data;
  int dim = -1;
  void Add(const std::string &k, const vbio::Matrix &m) {
    keys.push_back(k);
    data.insert(data.end(), m.data.begin(), m.data.end());
    offs.push_back(offs.back() + m.rows);
  }
  int64_t frames() const { return offs.back(); }
  void Clear() { keys.clear(); offs.assign(1, 0); data.clear(); }
};

constexpr int64_t kBatchFrames = 1 << 20;   // rows per device pass of the table-to-table tools

// runs one vbh_featpost configuration over a whole feature table
template <class PerBatch>
int TableToTable(const std::string &rspec, const std::string &wspec, Log &log, const char *prog, int expect_dim_multiple,
                 PerBatch per_batch) {
  vbio::MatrixReader reader(rspec);
  vbio::MatrixWriter writer(wspec);
  Batch b;
  std::string key; vbio::Matrix m;
  int done = 0, err = 0;
  auto flush = [&]() {
    if (b.keys.empty()) return;
    std::vector<float> out; int out_dim = 0;
    per_batch(b, &out, &out_dim);
    for (size_t u = 0; u < b.keys.size(); u++) {
      vbio::Matrix o;
      o.rows = (int32_t)(b.offs[u + 1] - b.offs[u]); o.cols = out_dim;
      o.data.assign(out.begin() + b.offs[u] * out_dim, out.begin() + b.offs[u + 1] * out_dim);
      writer.Write(b.keys[u], o);
      done++;
    }
    b.Clear();
  };
  while (reader.Next(&key, &m)) {
    if (m.rows == 0) { log.os() << "WARNING (" << prog << ") Empty feature matrix for key " << key << "\n"; err++; continue; }
    if (b.dim < 0) b.dim = m.cols;
    if (m.cols != b.dim) { log.os() << "WARNING (" << prog << ") dimension mismatch for key " << key << "\n"; err++; continue; }
    (void)expect_dim_multiple;
    b.Add(key, m);
    if (b.frames() >= kBatchFrames) flush();
  }
  flush();
  writer.Close();
  log.os() << "LOG (" << prog << ") Processed " << done << " feature matrices; " << err << " with errors.\n";
  return done != 0 ? 0 : 1;
}

void RunFeatpost(const vbh_featpost_cfg &cfg, const Batch &b, const std::vector<int32_t> &utt2spk, int n_spk,
                 const double *stats, const float *lda, const float *spk, std::vector<float> *out, int *out_dim) {
  vbh_ctx *ctx = Ctx();
  *out_dim = vbh_featpost_out_dim(&cfg);
  const int64_t T = b.frames();
  DevMem d_in(ctx, sizeof(float) * b.data.size(), 1), d_out(ctx, sizeof(float) * (size_t)T * *out_dim, 2);
  Check(vbh_copy_h2d(ctx, d_in.p, b.data.data(), sizeof(float) * b.data.size()), "vbh_copy_h2d");
  Check(vbh_featpost(ctx, &cfg, (const float *)d_in.p, b.offs.data(), (int32_t)b.keys.size(),
                     utt2spk.empty() ? nullptr : utt2spk.data(), n_spk, stats, lda, spk, (float *)d_out.p), "vbh_featpost");
  out->resize((size_t)T * *out_dim);
  Check(vbh_copy_d2h(ctx, out->data(), d_out.p, sizeof(float) * out->size()), "vbh_copy_d2h");
}

// Model, graph and search workspace of the calling thread's last gmm-latgen-faster call, kept on the device: the nj jobs of a Decode
// step (and the passes of DecodeFmllr) name the same files again and again, and reading a 100 MB graph, building its device form and
// allocating tens of GB of workspace per job cost more than the search of a job itself.  Keyed by path, size and modification time of
// the two files; released by vb_release_device_cache(), by a call with other files, or when the thread's device changes.
// The decoding graph on a device, shared by the threads of the process (the jobs of a step run on threads of their own when
// several are allowed): the file is read and its device form built once per (file, device); searches of different threads
// use it at the same time (they must name the same transition model, which the jobs of a step do).
struct GraphHolder {
  vbh_graph *g = nullptr;
  int32_t start = 0;
  ~GraphHolder() { if (g) vbh_graph_destroy(g); }
};
std::mutex g_graph_mu;
std::map<std::string, std::weak_ptr<GraphHolder>> g_graphs;
template <class Lap>
std::shared_ptr<GraphHolder> SharedGraph(vbh_ctx *ctx, int device, const std::string &path, long long sig, Lap lap) {
  const std::string key = std::to_string(device) + "|" + std::to_string(sig) + "|" + path;
  std::lock_guard<std::mutex> lk(g_graph_mu);          // (a second thread asking for the same graph waits for the first one's load)
  auto it = g_graphs.find(key);
  if (it != g_graphs.end()) { if (auto sp = it->second.lock()) return sp; }
  vbio::Graph graph;
  vbio::ReadFst(path, &graph);
  lap("read graph");
  static_assert(sizeof(vbio::Arc) == sizeof(vbh_arc), "arc layout");
  auto sp = std::make_shared<GraphHolder>();
  Check(vbh_graph_create(ctx, (int32_t)graph.final_cost.size(), graph.start, graph.final_cost.data(), graph.arc_offset.data(),
                         reinterpret_cast<const vbh_arc *>(graph.arcs.data()), &sp->g), "vbh_graph_create");
  sp->start = graph.start;
  g_graphs[key] = sp;
  lap("device graph");
  return sp;
}

struct LatgenCache {
  std::string model_path, graph_path;
  long long model_sig = 0, graph_sig = 0;
  vbh_ctx *ctx = nullptr;
  vbio::Model model;
  int32_t graph_start = 0;
  vbh_gmm *gmm = nullptr;
  std::shared_ptr<GraphHolder> gh;
  // one search workspace per layout of the search kernel (voicebridge_hip.h vbh_decode_layout_for): DecodeFmllr alternates between a
  // light first pass (max-active 2000: two utterances per CU) and the heavy main pass; [layout - 1]
  struct Dec { vbh_decoder *dec = nullptr; vbh_decode_cfg cfg{}; int batch = 0, T = 0; } decs[2];
  void DropDecoders() { for (Dec &d : decs) { if (d.dec) vbh_decoder_destroy(d.dec); d = Dec(); } }
  void *d_ll = nullptr;            // the batch's scores [frames x pdfs]: gigabytes per job, kept (grow-only) like the search workspace
  size_t d_ll_bytes = 0;
  void *Scores(vbh_ctx *c, size_t bytes) {
    if (bytes > d_ll_bytes) {
      if (d_ll) vbh_dev_free(c, d_ll);
      d_ll = nullptr; d_ll_bytes = 0;
      Check(vbh_dev_alloc(c, bytes, &d_ll), "vbh_dev_alloc");
      d_ll_bytes = bytes;
    }
    return d_ll;
  }
  void Release() {
    if (d_ll && ctx) vbh_dev_free(ctx, d_ll);
    d_ll = nullptr; d_ll_bytes = 0;
    DropDecoders();
    gh.reset();
    if (gmm) vbh_gmm_destroy(gmm);
    gmm = nullptr; model_path.clear(); graph_path.clear(); ctx = nullptr;
  }
  ~LatgenCache() { /* the process is ending or the thread is: the runtime reclaims device memory */ }
};
thread_local LatgenCache g_latgen;
std::atomic<int> g_workspace_max_active{0};       // vb_set_workspace_max_active (process-wide: a step's jobs run on threads of their own): see GmmLatgenFaster

long long FileSig(const std::string &path) {
  struct stat st;
  if (stat(path.c_str(), &st) != 0) return -1;
  return (long long)st.st_size * 1000003LL + (long long)st.st_mtime * 7919LL + (long long)st.st_mtim.tv_nsec;
}

int WindowType(const std::string &s) {
  if (s == "hanning") return VBH_WIN_HANNING;
  if (s == "hamming") return VBH_WIN_HAMMING;
  if (s == "povey") return VBH_WIN_POVEY;
  if (s == "rectangular") return VBH_WIN_RECTANGULAR;
  if (s == "blackman") return VBH_WIN_BLACKMAN;
  throw ApiError("Invalid window type " + s);
}

}  // namespace

// =====================================================================================
vbh_ctx *vbapi::SharedCtx() { return Ctx(); }

// GetRawLattice (K/decoder/lattice-faster-decoder.cc:110-192) from a search result: one state per
// token (state 0 = the start token), arc weights (graph, acoustic - cost_offset), acoustic costs
// still scaled, final weights (final cost, 0)
void vbapi::RawLatticeFromResult(const vbh_lattice &L, int32_t graph_start, vblat::Lattice *lat) {
  const int n = L.num_toks;
  // tokens are sorted by (frame, HCLG state); the start token must become state 0
  std::vector<int32_t> remap(n);
  int start_idx = 0;
  for (int i = 0; i < n && L.tok_frame[i] == 0; i++) if (L.tok_state[i] == graph_start) start_idx = i;
  for (int i = 0; i < n; i++) remap[i] = i == start_idx ? 0 : (i < start_idx ? i + 1 : i);
  lat->st.assign(n, vblat::State());
  lat->start = n ? 0 : -1;
  for (int i = 0; i < n; i++) if (!std::isinf(L.tok_final[i])) lat->st[remap[i]].final = vblat::Weight{L.tok_final[i], 0.0f};
  for (int l = 0; l < L.num_links; l++)
    lat->st[remap[L.link_src[l]]].arcs.push_back(vblat::Arc{L.link_ilabel[l], L.link_olabel[l], vblat::Weight{L.link_graph[l], L.link_acoustic[l]},
                                                            remap[L.link_dst[l]]});
}

// compute-mfcc-feats   (VB/src/featbin/compute-mfcc-feats.cpp:21-198)
int ComputeMFCCFeats(int argc, char *argv[], std::ofstream &file_log) {
  return Guard("compute-mfcc-feats", &file_log, [&](Log &log) -> int {
    Options po;
    po.Know({"sample-frequency", "frame-length", "frame-shift", "preemphasis-coefficient", "remove-dc-offset", "dither",
             "window-type", "blackman-coeff", "round-to-power-of-two", "snip-edges", "allow-downsample", "num-mel-bins",
             "low-freq", "high-freq", "vtln-low", "vtln-high", "debug-mel", "num-ceps", "use-energy", "energy-floor",
             "raw-energy", "cepstral-lifter", "htk-compat", "output-format", "subtract-mean", "vtln-warp", "vtln-map",
             "utt2spk", "channel", "min-duration",
             // extensions of this build: what the MakeMfcc step otherwise pipes through copy-feats (VB/scr/steps/make_mfcc.cpp:215)
             "compress", "write-num-frames", "dither-seed", "segments", "min-segment-length", "max-overshoot"});
    po.Parse(argc, argv);
    if (po.args.size() != 2) throw ApiError("wrong arguments.");
    vbh_mfcc_opts o;
    vbh_mfcc_opts_default(&o);
    // --dither: the reference's default 1.0 (K/feat/feature-window.h:55), so that a stock mfcc.conf runs.  The noise has the
    // reference's distribution but comes from a counter-based generator, not from libc rand() (include/voicebridge_hip.h,
    // vbh_mfcc_set_dither_stream): features equal the reference's statistically, bit for bit only with --dither=0.
    // --dither-seed (extension) selects the stream; an utterance's noise depends on the seed and on its position in the table.
    o.dither = (float)po.Num("dither", 1.0);
    if (o.dither < 0.0f) throw ApiError("--dither must not be negative");
    const uint64_t dither_seed = (uint64_t)po.Num("dither-seed", 0.0);
    o.samp_freq = (float)po.Num("sample-frequency", o.samp_freq);
    o.frame_length_ms = (float)po.Num("frame-length", o.frame_length_ms);
    o.frame_shift_ms = (float)po.Num("frame-shift", o.frame_shift_ms);
    o.preemph_coeff = (float)po.Num("preemphasis-coefficient", o.preemph_coeff);
    o.remove_dc_offset = po.Bool("remove-dc-offset", true);
    o.window_type = WindowType(po.Str("window-type", "povey"));
    o.blackman_coeff = (float)po.Num("blackman-coeff", o.blackman_coeff);
    o.round_to_power_of_two = po.Bool("round-to-power-of-two", true);
    o.snip_edges = po.Bool("snip-edges", true);
    o.num_bins = (int)po.Num("num-mel-bins", 23);
    o.low_freq = (float)po.Num("low-freq", o.low_freq); o.high_freq = (float)po.Num("high-freq", o.high_freq);
    o.vtln_low = (float)po.Num("vtln-low", o.vtln_low); o.vtln_high = (float)po.Num("vtln-high", o.vtln_high);
    o.num_ceps = (int)po.Num("num-ceps", 13); o.use_energy = po.Bool("use-energy", true);
    o.energy_floor = (float)po.Num("energy-floor", 0.0); o.raw_energy = po.Bool("raw-energy", true);
    o.cepstral_lifter = (float)po.Num("cepstral-lifter", 22.0); o.htk_compat = po.Bool("htk-compat", false);
    if (po.Str("output-format", "kaldi") != "kaldi") throw ApiError("Invalid output_format string " + po.Str("output-format", ""));
    // --vtln-map=<rspecifier> [--utt2spk=<rspecifier>]: a warp factor per utterance, or per speaker (K/featbin/compute-mfcc-feats.cc:82-86,
    // 127-137).  A warp factor is a set of mel-bank tables (vbh_mfcc_create); consecutive utterances with the same factor -- a speaker's,
    // in a sorted table -- go to the device in one call.
    const bool use_vtln_map = po.Has("vtln-map");
    if (po.Has("utt2spk") && !use_vtln_map) throw ApiError("the utt2spk option is only needed if the vtln-map option is used.");
    std::map<std::string, float> vtln_map;
    std::map<std::string, std::string> vtln_utt2spk;
    if (use_vtln_map) {
      vtln_map = vbio::ReadFloatTable(po.Str("vtln-map", ""));
      if (po.Has("utt2spk")) for (auto &kv : vbio::ReadTokenTable(po.Str("utt2spk", ""))) if (!kv.second.empty()) vtln_utt2spk[kv.first] = kv.second[0];
    }
    if (po.Bool("allow-downsample", false)) throw ApiError("--allow-downsample is not supported");
    const bool subtract_mean = po.Bool("subtract-mean", false);
    const float vtln_warp = (float)po.Num("vtln-warp", 1.0);
    const int channel = (int)po.Num("channel", -1);
    const double min_duration = po.Num("min-duration", 0.0);

    const vbio::Rspec rs = vbio::ParseRspecifier(po.args[0]);
    if (!rs.is_scp) throw ApiError("wav rspecifier must be scp: (got " + po.args[0] + ")");
    std::vector<vbio::ScpEntry> scp = vbio::ReadScp(rs.path);
    // --segments=<file> (extension): what the reference's MakeMfcc pipes through extract-segments when the data directory has a
    // segments file (VB/scr/steps/make_mfcc.cpp:93-190; K/featbin/extract-segments.cc:60-195): the table holds recordings, an
    // utterance is the piece <segment-id> <recording-id> <start> <end> [channel] of one, with that program's checks and warnings
    struct Segment { std::string id; size_t rec = 0; double start = 0, end = 0; int channel = -1; };
    std::vector<Segment> segments;
    const bool use_segments = po.Has("segments");
    if (use_segments) {
      std::map<std::string, size_t> rec_index;
      for (size_t k = 0; k < scp.size(); k++) rec_index[scp[k].key] = k;
      std::ifstream sf(po.Str("segments", ""));
      if (!sf) throw ApiError("cannot open " + po.Str("segments", ""));
      std::string line;
      while (std::getline(sf, line)) {
        std::vector<std::string> f; { std::istringstream is(line); std::string w; while (is >> w) f.push_back(w); }
        if (f.empty()) continue;
        if (f.size() != 4 && f.size() != 5) { log.os() << "WARNING (extract-segments) Invalid line in segments file: " << line << "\n"; continue; }
        char *e1 = nullptr, *e2 = nullptr;
        Segment sg; sg.id = f[0];
        sg.start = strtod(f[2].c_str(), &e1); sg.end = strtod(f[3].c_str(), &e2);
        if (e1 == f[2].c_str() || *e1) { log.os() << "WARNING (extract-segments) Invalid line in segments file [bad start]: " << line << "\n"; continue; }
        if (e2 == f[3].c_str() || *e2) { log.os() << "WARNING (extract-segments) Invalid line in segments file [bad end]: " << line << "\n"; continue; }
        if (sg.start < 0 || (sg.end != -1.0 && sg.end <= 0) || (sg.start >= sg.end && sg.end > 0)) {
          log.os() << "WARNING (extract-segments) Invalid line in segments file [empty or invalid segment]: " << line << "\n"; continue;
        }
        if (f.size() == 5) {
          char *e3 = nullptr; const long c = strtol(f[4].c_str(), &e3, 10);
          if (e3 == f[4].c_str() || *e3 || c < 0) { log.os() << "WARNING (extract-segments) Invalid line in segments file [bad channel]: " << line << "\n"; continue; }
          sg.channel = (int)c;
        }
        auto it = rec_index.find(f[1]);
        if (it == rec_index.end()) { log.os() << "WARNING (extract-segments) Could not find recording " << f[1] << ", skipping segment " << sg.id << "\n"; continue; }
        sg.rec = it->second;
        segments.push_back(sg);
      }
    }
    const double min_segment_length = po.Num("min-segment-length", 0.1), max_overshoot = po.Num("max-overshoot", 0.5);
    vbio::MatrixWriter writer(po.args[1], po.Bool("compress", false));
    std::ofstream num_frames_out;
    if (po.Has("write-num-frames")) {
      const vbio::Wspec ws = vbio::ParseWspecifier(po.Str("write-num-frames", ""));
      if (!ws.has_ark || !ws.text) throw ApiError("--write-num-frames expects ark,t:<file>");
      num_frames_out.open(ws.ark_path);
      if (!num_frames_out) throw ApiError("cannot open " + ws.ark_path);
    }
    vbh_ctx *ctx = Ctx();
    // one front end per warp factor in use (the mel banks are host-built tables on the device); all destroyed at the end
    struct Banks {
      vbh_ctx *ctx; const vbh_mfcc_opts *o; std::map<uint32_t, vbh_mfcc *> by_warp;
      vbh_mfcc *Get(float warp) {
        uint32_t bits; memcpy(&bits, &warp, 4);
        auto it = by_warp.find(bits);
        if (it != by_warp.end()) return it->second;
        vbh_mfcc *m = nullptr;
        Check(vbh_mfcc_create(ctx, o, warp, &m), "vbh_mfcc_create");
        by_warp[bits] = m;
        return m;
      }
      ~Banks() { for (auto &kv : by_warp) vbh_mfcc_destroy(kv.second); }
    } banks{ctx, &o, {}};
    if (!use_vtln_map) banks.Get(vtln_warp);       // (option errors surface before any file is read)
    int num_utts = 0, num_success = 0;
    size_t num_success_batches_base = 0;      // utterances handed to the device so far: the dither stream's utterance index
    // utterances are packed into batches of ~64 M samples per device pass
    // the units of work: the table's entries, or the pieces of them a segments file names
    struct Unit { std::string key; size_t rec; int seg; };
    std::vector<Unit> units;
    if (use_segments) for (size_t k = 0; k < segments.size(); k++) units.push_back(Unit{segments[k].id, segments[k].rec, (int)k});
    else for (size_t k = 0; k < scp.size(); k++) units.push_back(Unit{scp[k].key, k, -1});
    size_t i = 0;
    while (i < units.size()) {
      std::vector<std::string> keys; std::vector<int64_t> woff(1, 0); std::vector<float> samples; std::vector<float> warps;
      // the files of a batch are read and parsed on several host threads (a 30 s utterance is 1 MB of PCM), then taken in table order;
      // a recording that several segments of the batch cut from is read once
      const size_t i0 = i;
      size_t i1 = i0, est = 0;
      std::vector<size_t> recs; std::map<size_t, size_t> rec_slot;
      while (i1 < units.size() && est < (size_t)(64 << 20)) {
        if (!rec_slot.count(units[i1].rec)) {
          struct stat st;
          est += stat(scp[units[i1].rec].path.c_str(), &st) == 0 ? (size_t)st.st_size / 2 : (size_t)1 << 16;
          rec_slot[units[i1].rec] = recs.size(); recs.push_back(units[i1].rec);
        }
        i1++;
      }
      std::vector<vbio::Wave> waves(recs.size());
      std::vector<std::string> werr(recs.size());
      {
        std::atomic<size_t> next{0};
        auto work = [&]() {
          for (size_t k = next++; k < recs.size(); k = next++) {
            const vbio::ScpEntry &e = scp[recs[k]];
            if (!e.path.empty() && e.path.back() == '|') continue;
            try { vbio::ReadWaveFile(e.path, e.offset, &waves[k]); } catch (const std::exception &ex) { werr[k] = ex.what(); if (werr[k].empty()) werr[k] = "read error"; }
          }
        };
        const unsigned nt = std::max(1u, std::min<unsigned>(8, std::min<unsigned>((unsigned)recs.size(), std::thread::hardware_concurrency())));
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < nt; t++) pool.emplace_back(work);
        work();
        for (std::thread &t : pool) t.join();
      }
      for (; i < i1; i++) {
        const Unit &un = units[i];
        const vbio::ScpEntry &e = scp[un.rec];
        num_utts++;
        if (!e.path.empty() && e.path.back() == '|') { log.os() << "WARNING (compute-mfcc-feats) pipe input not supported: " << e.key << "\n"; continue; }
        vbio::Wave &w = waves[rec_slot[un.rec]];
        if (!werr[rec_slot[un.rec]].empty()) { log.os() << "WARNING (compute-mfcc-feats) " << e.key << ": " << werr[rec_slot[un.rec]] << "\n"; continue; }
        // the samples of the unit: the whole recording, or the segment's piece of it (extract-segments.cc:126-190)
        size_t s0 = 0, s1 = w.data.empty() ? 0 : w.data[0].size();
        int ch = channel;
        if (un.seg >= 0) {
          const Segment &sg = segments[(size_t)un.seg];
          const int32_t num_samp = (int32_t)s1;
          const int32_t start_samp = (int32_t)(sg.start * w.samp_freq);
          int32_t end_samp = sg.end != -1.0 ? (int32_t)(sg.end * w.samp_freq) : num_samp;
          if (start_samp < 0 || start_samp >= num_samp) {
            log.os() << "WARNING (extract-segments) Start sample out of range " << start_samp << " [length:] " << num_samp << ", skipping segment " << sg.id << "\n"; continue;
          }
          if (end_samp > num_samp) {
            if (end_samp >= num_samp + (int32_t)(max_overshoot * w.samp_freq)) {
              log.os() << "WARNING (extract-segments) End sample too far out of range " << end_samp << " [length:] " << num_samp << ", skipping segment " << sg.id << "\n"; continue;
            }
            end_samp = num_samp;
          }
          if (end_samp <= start_samp + (int32_t)(min_segment_length * w.samp_freq)) { log.os() << "WARNING (extract-segments) Segment " << sg.id << " too short, skipping it.\n"; continue; }
          if (sg.channel == -1) {
            if (w.channels != 1) throw ApiError("If your data has multiple channels, you must specify the channel in the segments file.  Processing segment " + sg.id);
            ch = 0;
          } else if (sg.channel >= w.channels) { log.os() << "WARNING (extract-segments) Invalid channel " << sg.channel << " >= " << w.channels << ", processing segment " << sg.id << "\n"; continue; }
          else ch = sg.channel;
          s0 = (size_t)start_samp; s1 = (size_t)end_samp;
        }
        const double dur = (double)(s1 - s0) / w.samp_freq;
        if (dur < min_duration) { log.os() << "WARNING (compute-mfcc-feats) File: " << un.key << " is too short (" << dur << " sec): producing no output.\n"; continue; }
        if (un.seg < 0) {
          if (ch == -1) {
            ch = 0;
            if (w.channels != 1) log.os() << "WARNING (compute-mfcc-feats) Channel not specified but you have data with " << w.channels << " channels; defaulting to zero\n";
          } else if (ch >= w.channels) { log.os() << "WARNING (compute-mfcc-feats) File with id " << un.key << " has " << w.channels << " channels but you specified channel " << channel << ", producing no output.\n"; continue; }
        }
        if (w.samp_freq != o.samp_freq) { log.os() << "WARNING (compute-mfcc-feats) Failed to compute features for utterance " << un.key << ": sample frequency mismatch " << w.samp_freq << " vs " << o.samp_freq << "\n"; continue; }
        float warp_local = vtln_warp;
        if (use_vtln_map) {
          std::string mk = un.key;
          if (!vtln_utt2spk.empty() || po.Has("utt2spk")) { const auto iu = vtln_utt2spk.find(un.key); mk = iu == vtln_utt2spk.end() ? std::string() : iu->second; }
          const auto iw = mk.empty() ? vtln_map.end() : vtln_map.find(mk);
          if (iw == vtln_map.end()) { log.os() << "WARNING (compute-mfcc-feats) No vtln-map entry for utterance-id (or speaker-id) " << un.key << "\n"; continue; }
          warp_local = iw->second;
        }
        warps.push_back(warp_local);
        keys.push_back(un.key);
        samples.insert(samples.end(), w.data[ch].begin() + (std::ptrdiff_t)s0, w.data[ch].begin() + (std::ptrdiff_t)s1);
        woff.push_back((int64_t)samples.size());
      }
      std::vector<vbio::Wave>().swap(waves);
      if (keys.empty()) continue;
      int64_t total = 0;
      for (size_t u = 0; u < keys.size(); u++) total += vbh_mfcc_num_frames(&o, woff[u + 1] - woff[u]);
      DevMem d_wave(ctx, sizeof(float) * samples.size(), 3), d_feats(ctx, sizeof(float) * (size_t)total * o.num_ceps, 4);
      Check(vbh_copy_h2d(ctx, d_wave.p, samples.data(), sizeof(float) * samples.size()), "vbh_copy_h2d");
      std::vector<int64_t> foff(keys.size() + 1, 0);
      for (size_t g0 = 0; g0 < keys.size();) {        // runs of utterances with one warp factor
        size_t g1 = g0 + 1;
        while (g1 < keys.size() && warps[g1] == warps[g0]) g1++;
        vbh_mfcc *mf = banks.Get(warps[g0]);
        if (o.dither != 0.0f) {
          // the noise of an utterance is keyed by a hash of its NAME (FNV-1a), not by its position in the table: the same utterance gets the
          // same noise whatever job it is in, and no two jobs of MakeMfcc share a noise sequence
          Check(vbh_mfcc_set_dither_stream(mf, dither_seed ? dither_seed : 0x4F6CDD1D2545F491ull, 0), "vbh_mfcc_set_dither_stream");
          std::vector<int64_t> ids;
          for (size_t u = g0; u < g1; u++) {
            uint64_t h = 1469598103934665603ull;
            for (unsigned char c : keys[u]) { h ^= c; h *= 1099511628211ull; }
            ids.push_back((int64_t)(h & 0x7FFFFFFFFFFFFFFFull));
          }
          Check(vbh_mfcc_set_dither_ids(mf, ids.data(), (int32_t)ids.size()), "vbh_mfcc_set_dither_ids");
        }
        std::vector<int64_t> fo(g1 - g0 + 1);
        Check(vbh_mfcc_compute(mf, d_wave.p, 0, woff.data() + g0, (int32_t)(g1 - g0), (float *)d_feats.p + foff[g0] * o.num_ceps, fo.data()), "vbh_mfcc_compute");
        for (size_t u = g0; u < g1; u++) foff[u + 1] = foff[g0] + fo[u - g0 + 1];
        g0 = g1;
      }
      num_success_batches_base += keys.size();
      std::vector<float> feats((size_t)total * o.num_ceps);
      Check(vbh_copy_d2h(ctx, feats.data(), d_feats.p, sizeof(float) * feats.size()), "vbh_copy_d2h");
      // the utterances are encoded (compressed archives: a sort per column) on several threads, written in table order
      std::vector<std::string> objs(keys.size());
      {
        std::atomic<size_t> next{0};
        auto work = [&]() {
          for (size_t u = next++; u < keys.size(); u = next++) {
            vbio::Matrix m;
            m.rows = (int32_t)(foff[u + 1] - foff[u]); m.cols = o.num_ceps;
            m.data.assign(feats.begin() + foff[u] * o.num_ceps, feats.begin() + foff[u + 1] * o.num_ceps);
            if (m.rows == 0) m.cols = 0;   // too short for one frame: the reference emits a 0 x 0 matrix (feature-common-inl.h:81-84)
            if (subtract_mean && m.rows > 0) {   // compute-mfcc-feats.cpp:153-158
              for (int c = 0; c < m.cols; c++) {
                double s = 0.0;
                for (int r = 0; r < m.rows; r++) s += m.at(r, c);
                const float mean = (float)(s / m.rows);
                for (int r = 0; r < m.rows; r++) m.at(r, c) -= mean;
              }
            }
            objs[u] = vbio::SerializeMatrix(m, writer.text(), writer.compress());
          }
        };
        const unsigned nt = std::max(1u, std::min<unsigned>(8, std::min<unsigned>((unsigned)keys.size(), std::thread::hardware_concurrency())));
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < nt; t++) pool.emplace_back(work);
        work();
        for (std::thread &t : pool) t.join();
      }
      for (size_t u = 0; u < keys.size(); u++) {
        writer.WriteSerialized(keys[u], objs[u]);
        if (num_frames_out.is_open()) num_frames_out << keys[u] << ' ' << (foff[u + 1] - foff[u]) << '\n';
        num_success++;
      }
    }
    writer.Close();
    log.os() << "LOG (compute-mfcc-feats)  Done " << num_success << " out of " << num_utts << " utterances.\n";
    return num_success != 0 ? 0 : 1;
  });
}

// =====================================================================================
// compute-cmvn-stats --spk2utt   (VB/src/featbin/compute-cmvn-stats.cpp)
int ComputeCmvnStats(int argc, char *argv[], std::ofstream &file_log) {
  return Guard("compute-cmvn-stats", &file_log, [&](Log &log) -> int {
    Options po;
    po.Know({"spk2utt", "binary", "weights"});
    po.Parse(argc, argv);
    if (po.args.size() != 2) throw ApiError("wrong arguments.");
    if (po.Has("weights")) throw ApiError("--weights is not supported");
    // spk2utt: "spk utt1 utt2 ..." lines; without it every utterance is its own speaker
    std::map<std::string, std::string> utt2spk;
    std::vector<std::string> spk_order;
    if (po.Has("spk2utt")) {
      const vbio::Rspec rs = vbio::ParseRspecifier(po.Str("spk2utt", ""));
      std::ifstream f(rs.path);
      if (!f) throw ApiError("cannot open " + rs.path);
      std::string line;
      while (std::getline(f, line)) {
        std::istringstream ss(line);
        std::string spk, utt;
        if (!(ss >> spk)) continue;
        spk_order.push_back(spk);
        while (ss >> utt) utt2spk[utt] = spk;
      }
    }
    vbio::MatrixReader reader(po.args[0]);
    Batch b; std::string key; vbio::Matrix m;
    std::map<std::string, int> spk_index;
    std::vector<int32_t> u2s;
    while (reader.Next(&key, &m)) {
      std::string spk = key;
      if (!utt2spk.empty()) {
        auto it = utt2spk.find(key);
        if (it == utt2spk.end()) { log.os() << "WARNING (compute-cmvn-stats) no speaker for utterance " << key << "\n"; continue; }
        spk = it->second;
      } else spk_order.push_back(key);
      if (b.dim < 0) b.dim = m.cols;
      if (m.cols != b.dim || m.rows == 0) continue;
      if (!spk_index.count(spk)) { const int idx = (int)spk_index.size(); spk_index[spk] = idx; }
      u2s.push_back(spk_index[spk]);
      b.Add(key, m);
    }
    if (b.keys.empty()) { log.os() << "LOG (compute-cmvn-stats) no features\n"; return 1; }
    vbh_ctx *ctx = Ctx();
    const int n_spk = (int)spk_index.size(), D = b.dim;
    DevMem d_in(ctx, sizeof(float) * b.data.size(), 1);
    Check(vbh_copy_h2d(ctx, d_in.p, b.data.data(), sizeof(float) * b.data.size()), "vbh_copy_h2d");
    std::vector<double> stats((size_t)n_spk * 2 * (D + 1));
    Check(vbh_cmvn_stats(ctx, (const float *)d_in.p, D, b.offs.data(), (int32_t)b.keys.size(), u2s.data(), n_spk, stats.data()), "vbh_cmvn_stats");
    const vbio::Wspec ws = vbio::ParseWspecifier(po.args[1]);
    std::ofstream ark(ws.ark_path, std::ios::binary), scp;
    if (!ark) throw ApiError("cannot open " + ws.ark_path);
    if (ws.has_scp) scp.open(ws.scp_path);
    int done = 0;
    for (const std::string &spk : spk_order) {
      auto it = spk_index.find(spk);
      if (it == spk_index.end()) continue;
      vbio::MatrixD sm; sm.rows = 2; sm.cols = D + 1;
      sm.data.assign(stats.begin() + (size_t)it->second * 2 * (D + 1), stats.begin() + (size_t)(it->second + 1) * 2 * (D + 1));
      ark << spk << ' ';
      const int64_t off = (int64_t)ark.tellp();
      vbio::WriteMatrixD(ark, !ws.text, sm);
      if (ws.has_scp) scp << spk << ' ' << ws.ark_path << ':' << off << '\n';
      done++;
    }
    log.os() << "LOG (compute-cmvn-stats) Done accumulating CMVN stats for " << done << " speakers.\n";
    return done != 0 ? 0 : 1;
  });
}

// =====================================================================================
// apply-cmvn   (VB/src/featbin/apply-cmvn.cpp:26-187)
int ApplyCmvn(int argc, char *argv[], std::ofstream &file_log) {
  return Guard("apply-cmvn", &file_log, [&](Log &log) -> int {
    Options po;
    po.Know({"utt2spk", "norm-vars", "norm-means", "skip-dims", "reverse"});
    po.Parse(argc, argv);
    if (po.args.size() != 3) throw ApiError("wrong arguments.");
    if (!po.Bool("norm-means", true) || po.Has("skip-dims") || po.Bool("reverse", false))
      throw ApiError("--norm-means=false / --skip-dims / --reverse are not supported");
    const bool norm_vars = po.Bool("norm-vars", false);
    std::map<std::string, std::string> utt2spk;
    if (po.Has("utt2spk")) utt2spk = vbio::ReadStringMap(po.Str("utt2spk", ""));
    // per-speaker stats table ("ark:" / "scp:"); a plain rxfilename means global stats
    std::map<std::string, vbio::MatrixD> stats;
    vbio::MatrixD global;
    bool use_global = false;
    const std::string &st = po.args[0];
    if (st.compare(0, 3, "ark") == 0 || st.compare(0, 3, "scp") == 0) stats = vbio::ReadMatrixDTable(st);
    else {
      std::ifstream f(st, std::ios::binary);
      if (!f) throw ApiError("cannot open " + st);
      const bool binary = vbio::ReadBinaryHeader(f);
      vbio::ReadMatrixD(f, binary, &global);
      use_global = true;
    }
    return TableToTable(po.args[1], po.args[2], log, "apply-cmvn", 1, [&](const Batch &b, std::vector<float> *out, int *od) {
      std::map<std::string, int> idx; std::vector<int32_t> u2s; std::vector<double> packed;
      const int D = b.dim;
      for (const std::string &k : b.keys) {
        std::string spk = k;
        if (!utt2spk.empty()) { auto it = utt2spk.find(k); if (it != utt2spk.end()) spk = it->second; }
        if (use_global) spk = "";
        if (!idx.count(spk)) {
          const vbio::MatrixD *m = use_global ? &global : nullptr;
          if (!m) { auto it = stats.find(spk); if (it == stats.end()) throw ApiError("No normalization statistics available for key " + k); m = &it->second; }
          if (m->cols != D + 1 || m->rows < 1 || m->rows > 2) throw ApiError("Dim mismatch: cmvn " + std::to_string(m->rows) + "x" + std::to_string(m->cols));
          if (m->rows == 1 && norm_vars) throw ApiError("You requested variance normalization but no variance stats are supplied.");
          const int n = (int)idx.size(); idx[spk] = n;
          packed.insert(packed.end(), m->data.begin(), m->data.end());
          if (m->rows == 1) packed.insert(packed.end(), (size_t)(D + 1), 0.0);
        }
        u2s.push_back(idx[spk]);
      }
      vbh_featpost_cfg cfg{};
      cfg.in_dim = D; cfg.apply_cmvn = 1; cfg.norm_vars = norm_vars; cfg.mode = 0;
      RunFeatpost(cfg, b, u2s, (int)idx.size(), packed.data(), nullptr, nullptr, out, od);
    });
  });
}

// =====================================================================================
// add-deltas   (VB/src/featbin/add-deltas.cpp:19-72)
int AddDeltas(int argc, char *argv[]) {
  return Guard("add-deltas", nullptr, [&](Log &log) -> int {
    Options po;
    po.Know({"delta-order", "delta-window", "truncate"});
    po.Parse(argc, argv);
    if (po.args.size() != 2) throw ApiError("wrong arguments.");
    if (po.Num("truncate", 0) != 0) throw ApiError("--truncate is not supported");
    const int order = (int)po.Num("delta-order", 2), window = (int)po.Num("delta-window", 2);
    return TableToTable(po.args[0], po.args[1], log, "add-deltas", 1, [&](const Batch &b, std::vector<float> *out, int *od) {
      vbh_featpost_cfg cfg{};
      cfg.in_dim = b.dim; cfg.mode = 1; cfg.delta_order = order; cfg.delta_window = window;
      RunFeatpost(cfg, b, {}, 1, nullptr, nullptr, nullptr, out, od);
    });
  });
}

// =====================================================================================
// splice-feats   (VB/src/featbin/splice-feats.cpp:20-62)
int SpliceFeats(int argc, char *argv[]) {
  return Guard("splice-feats", nullptr, [&](Log &log) -> int {
    Options po;
    po.Know({"left-context", "right-context"});
    po.Parse(argc, argv);
    if (po.args.size() != 2) throw ApiError("wrong arguments.");
    const int left = (int)po.Num("left-context", 4), right = (int)po.Num("right-context", 4);
    return TableToTable(po.args[0], po.args[1], log, "splice-feats", 1, [&](const Batch &b, std::vector<float> *out, int *od) {
      vbh_featpost_cfg cfg{};
      cfg.in_dim = b.dim; cfg.mode = 2; cfg.splice_left = left; cfg.splice_right = right;
      RunFeatpost(cfg, b, {}, 1, nullptr, nullptr, nullptr, out, od);
    });
  });
}

// =====================================================================================
// transform-feats   (VB/src/featbin/transform-feats.cpp:19-176)
int TransformFeats(int argc, char *argv[], std::ofstream &file_log) {
  return Guard("transform-feats", &file_log, [&](Log &log) -> int {
    Options po;
    po.Know({"utt2spk"});
    po.Parse(argc, argv);
    if (po.args.size() != 3) throw ApiError("wrong arguments.");
    std::map<std::string, std::string> utt2spk;
    if (po.Has("utt2spk")) utt2spk = vbio::ReadStringMap(po.Str("utt2spk", ""));
    const std::string &ts = po.args[0];
    std::map<std::string, vbio::Matrix> table;
    vbio::Matrix global;
    bool use_global = false;
    if (ts.compare(0, 3, "ark") == 0 || ts.compare(0, 3, "scp") == 0) table = vbio::ReadMatrixTable(ts);
    else {
      std::ifstream f(ts, std::ios::binary);
      if (!f) throw ApiError("cannot open " + ts);
      const bool binary = vbio::ReadBinaryHeader(f);
      vbio::ReadMatrix(f, binary, &global);
      use_global = true;
    }
    return TableToTable(po.args[1], po.args[2], log, "transform-feats", 1, [&](const Batch &b, std::vector<float> *out, int *od) {
      vbh_featpost_cfg cfg{};
      cfg.in_dim = b.dim; cfg.mode = 0;
      if (use_global) {
        cfg.lda_rows = global.rows; cfg.lda_cols = global.cols;
        RunFeatpost(cfg, b, {}, 1, nullptr, global.data.data(), nullptr, out, od);
        return;
      }
      std::map<std::string, int> idx; std::vector<int32_t> u2s; std::vector<float> packed;
      int R = -1, C = -1;
      for (const std::string &k : b.keys) {
        std::string spk = k;
        if (!utt2spk.empty()) { auto it = utt2spk.find(k); if (it != utt2spk.end()) spk = it->second; }
        if (!idx.count(spk)) {
          auto it = table.find(spk);
          if (it == table.end()) throw ApiError("No fMLLR transform available for utterance " + k);
          if (R < 0) { R = it->second.rows; C = it->second.cols; }
          if (it->second.rows != R || it->second.cols != C) throw ApiError("transform matrices of different sizes");
          const int n = (int)idx.size(); idx[spk] = n;
          packed.insert(packed.end(), it->second.data.begin(), it->second.data.end());
        }
        u2s.push_back(idx[spk]);
      }
      cfg.spk_rows = R; cfg.spk_cols = C;
      RunFeatpost(cfg, b, u2s, (int)idx.size(), nullptr, nullptr, packed.data(), out, od);
    });
  });
}

// =====================================================================================
// gmm-latgen-faster   (VB/src/gmmbin/gmm-latgen-faster.cpp:27-198)
int GmmLatgenFaster(int argc, char *argv[], std::ofstream &file_log) {
  return Guard("gmm-latgen-faster", &file_log, [&](Log &log) -> int {
    Options po;
    po.Know({"beam", "max-active", "min-active", "lattice-beam", "prune-interval", "determinize-lattice", "beam-delta",
             "hash-ratio", "acoustic-scale", "word-symbol-table", "allow-partial", "max-mem", "phone-determinize",
             "word-determinize", "minimize", "decode-batch", "workspace-max-active", "delta",
             // extensions of this build (not in the reference's program): the feature chain of the Decode step fused in front of the
             // scoring, on the device, so that the steps need no temporary archives between apply-cmvn, add-deltas / splice-feats +
             // transform-feats, transform-feats(fMLLR) and this program (VB/scr/steps/decode_gmm.cpp:201-293).  With --fe-cmvn the
             // feature rspecifier names the RAW features (e.g. the 13-dim MFCCs of feats.scp).
             "fe-cmvn", "fe-utt2spk", "fe-norm-vars", "fe-type", "fe-delta-order", "fe-delta-window", "fe-left-context", "fe-right-context",
             "fe-lda", "fe-trans",
             // --job-of-utt=<file of "utterance job" lines>: the table holds the utterances of SEVERAL jobs of a step, searched as one batch
             // (one persistent grid over all of them); every "JOB" in the three wspecifiers is replaced by the utterance's job number, so
             // that the step still leaves lat.JOB / w.JOB per job, in table order (VB/scr/steps/decode_gmm.cpp:300-324 runs the jobs side by
             // side; on one GPU one grid of all their utterances fills the chip better than nj grids that queue behind each other)
             "job-of-utt",
             // --admission=ordered|exact (vbh_decode_cfg::admission; default ordered = the reference's rule in a canonical visiting order,
             // exact = the order-independent subset: 3 % fewer tokens, ~5 % faster, not the reference's amount of search)
             "admission"});
    po.Parse(argc, argv);
    if (po.args.size() < 4 || po.args.size() > 6) throw ApiError("wrong arguments.");
    vbh_decode_cfg cfg;
    vbh_decode_cfg_default(&cfg);
    {
      const std::string adm = po.Str("admission", "ordered");
      if (adm == "exact") cfg.admission = VBH_ADMISSION_EXACT;
      else if (adm == "ordered") cfg.admission = VBH_ADMISSION_ORDERED;
      else throw ApiError("--admission must be ordered or exact, not " + adm);
    }
    // LatticeFasterDecoderConfig defaults (K/decoder/lattice-faster-decoder.h:59-67), not Decode()'s
    cfg.beam = (float)po.Num("beam", 16.0);
    cfg.max_active = (int)po.Num("max-active", (double)std::numeric_limits<int32_t>::max());
    cfg.min_active = (int)po.Num("min-active", 200);
    cfg.lattice_beam = (float)po.Num("lattice-beam", 10.0);
    cfg.prune_interval = (int)po.Num("prune-interval", 25);
    cfg.beam_delta = (float)po.Num("beam-delta", 0.5);
    cfg.acoustic_scale = (float)po.Num("acoustic-scale", 0.1);
    const bool allow_partial = po.Bool("allow-partial", false);
    const bool determinize = po.Bool("determinize-lattice", true);
    const std::string &lat_wspec = po.args[3];
    const bool want_lat = !lat_wspec.empty() && lat_wspec != "ark:/dev/null";
    vblat::PhoneDetOptions det_opts;      // LatticeFasterDecoderConfig::det_opts (K/decoder/lattice-faster-decoder.h:69-92)
    det_opts.delta = (float)po.Num("delta", vblat::kDelta);
    det_opts.max_mem = (int)po.Num("max-mem", 50000000);
    det_opts.phone_determinize = po.Bool("phone-determinize", true);
    det_opts.word_determinize = po.Bool("word-determinize", true);
    det_opts.minimize = po.Bool("minimize", false);
    const bool laps = getenv("VBH_DEBUG_TIMES") != nullptr;      // diagnostic: where the host time of a job goes
    auto t_lap = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
      if (!laps) return;
      const auto now = std::chrono::steady_clock::now();
      fprintf(stderr, "gmm-latgen-faster lap %-28s %8.1f ms\n", what, 1e3 * std::chrono::duration<double>(now - t_lap).count());
      t_lap = now;
    };
    vbh_ctx *ctx = Ctx();
    LatgenCache &cache = g_latgen;
    std::future<std::shared_ptr<GraphHolder>> graph_loading;
    // (the helper must not outlive this call: an exception on the way out waits for it first)
    struct JoinOnExit { std::future<std::shared_ptr<GraphHolder>> &f; ~JoinOnExit() { if (f.valid()) f.wait(); } } join_graph{graph_loading};
    auto need_graph = [&]() {
      if (!graph_loading.valid()) return;
      cache.gh = graph_loading.get();
      cache.graph_start = cache.gh->start;
      cache.ctx = ctx; cache.graph_path = po.args[1]; cache.graph_sig = FileSig(po.args[1]);
      lap("graph (loaded meanwhile)");
    };
    {
      const long long ms = FileSig(po.args[0]), gs = FileSig(po.args[1]);
      if (cache.ctx != ctx) cache.Release();
      // the graph (and with it the search workspace) and the model are kept independently: DecodeFmllr's passes use two or
      // three models over one graph
      if (!cache.gh || cache.graph_path != po.args[1] || cache.graph_sig != gs || gs < 0) {
        cache.DropDecoders();
        cache.gh.reset();
        // The graph (file -> host -> device form: the longest piece of a cold start) is fetched by a helper thread while this one
        // reads the model and the features and scores the first batch; it is waited for where the search workspace is made.
        const int device = g_device;
        const std::string gpath = po.args[1];
        graph_loading = std::async(std::launch::async, [ctx, device, gpath, gs]() { return SharedGraph(ctx, device, gpath, gs, [](const char *) {}); });
      }
      if (cache.gmm == nullptr || cache.model_path != po.args[0] || cache.model_sig != ms || ms < 0) {
        if (cache.gmm) { vbh_gmm_destroy(cache.gmm); cache.gmm = nullptr; }
        vbio::ReadModel(po.args[0], &cache.model);
        lap("read model");
        Check(vbh_gmm_create(ctx, cache.model.dim, cache.model.num_pdfs, cache.model.pdf_offsets.data(), cache.model.weights.data(),
                             cache.model.means_invvars.data(), cache.model.inv_vars.data(), &cache.gmm), "vbh_gmm_create");
        cache.ctx = ctx; cache.model_path = po.args[0]; cache.model_sig = ms;
        lap("device model");
      }
    }
    const vbio::Model &model = cache.model;
    vbh_gmm *gmm = cache.gmm;
    // ---- fused front end (extension, device_feats.h)
    FrontEnd fe;
    fe.Parse(po);
    lap("front-end options and tables");
    // the writers of an utterance's job (one set, job 0, without --job-of-utt)
    struct JobOut { vbio::LatticeWriter lat; vbio::CompactLatticeWriter clat; vbio::IntVectorWriter words, ali; };
    std::map<std::string, int> job_of_utt;
    if (po.Has("job-of-utt")) {
      std::ifstream f(po.Str("job-of-utt", ""));
      if (!f) throw ApiError("cannot open " + po.Str("job-of-utt", ""));
      std::string k; int j;
      while (f >> k >> j) job_of_utt[k] = j;
    }
    std::map<int, std::unique_ptr<JobOut>> outs;
    auto for_job = [](std::string spec, int job) {
      if (job > 0) for (size_t q = spec.find("JOB"); q != std::string::npos; q = spec.find("JOB", q)) { const std::string n = std::to_string(job); spec.replace(q, 3, n); q += n.size(); }
      return spec;
    };
    auto out_of_job = [&](int job) -> JobOut & {
      std::unique_ptr<JobOut> &o = outs[job];
      if (!o) {
        o.reset(new JobOut());
        if (want_lat) { if (determinize) o->clat.Open(for_job(lat_wspec, job)); else o->lat.Open(for_job(lat_wspec, job)); }
        if (po.args.size() >= 5 && !po.args[4].empty()) o->words.Open(for_job(po.args[4], job));
        if (po.args.size() >= 6 && !po.args[5].empty()) o->ali.Open(for_job(po.args[5], job));
      }
      return *o;
    };
    auto out_of = [&](const std::string &utt) -> JobOut & {
      if (job_of_utt.empty()) return out_of_job(0);
      const auto it = job_of_utt.find(utt);
      if (it == job_of_utt.end()) throw ApiError("--job-of-utt names no job for utterance " + utt);
      return out_of_job(it->second);
    };
    // (the files exist even when no utterance of a job is decoded, as the reference's do)
    if (job_of_utt.empty()) (void)out_of_job(0);
    else for (const auto &kv : job_of_utt) (void)out_of_job(kv.second);

    lap("output tables opened");
    int batch_utts = (int)po.Num("decode-batch", 256);
    if (!po.Has("decode-batch")) if (const char *e = getenv("VB_DECODE_BATCH")) { const int x = atoi(e); if (x > 0) batch_utts = x; }      // (tuning: utterances per search grid)
    // the cached workspace serves this call when its capacities cover the search configuration (vbh_decoder_set_cfg: beams and
    // scales are free, max_active up to what the workspaces were made for); --workspace-max-active (extension) asks for workspaces
    // that also cover a later call with a larger max_active (DecodeFmllr: first pass 2000, main pass 7000)
    // A light configuration (layout 2) gets workspaces of its own, sized for itself; the workspace-wide max_active applies to the heavy layout
    const int layout = vbh_decode_layout_for(&cfg);
    LatgenCache::Dec &cd = cache.decs[layout == 2 ? 1 : 0];
    vbh_decode_cfg ws_cfg = cfg;
    if (layout != 2) {
      ws_cfg.max_active = std::max(cfg.max_active, (int)po.Num("workspace-max-active", 0.0));
      if (g_workspace_max_active.load() > ws_cfg.max_active) ws_cfg.max_active = g_workspace_max_active.load();
    }
    if (cd.dec && memcmp(&cd.cfg, &cfg, sizeof(cfg)) != 0 && vbh_decoder_set_cfg(cd.dec, &cfg) != 0) {
      vbh_decoder_destroy(cd.dec); cd = LatgenCache::Dec();
    }
    cd.cfg = cfg;
    vbh_decoder *&dec = cd.dec;
    int &dec_batch = cd.batch, &dec_T = cd.T;
    // the feature table on the device: read and uploaded once per job (the thread's cached copy serves the job's other programs)
    lap("search configuration");
    const RawTable &tab = CachedRawTable(po.args[2]);
    lap("features to the device");
    double tot_like = 0.0;
    int64_t frame_count = 0;
    int num_done = 0, num_err = 0;
    for (size_t u0 = 0; u0 < tab.keys.size();) {
      const size_t u1 = std::min(tab.keys.size(), u0 + (size_t)std::max(batch_utts, 1));
      Batch b;        // keys and row offsets of the batch (the rows themselves stay on the device)
      int max_T = 0;
      for (size_t u = u0; u < u1; u++) {
        b.keys.push_back(tab.keys[u]);
        b.offs.push_back(tab.offs[u + 1] - tab.offs[u0]);
        max_T = std::max(max_T, tab.rows(u));
      }
      u0 = u1;
      const int64_t T = b.frames();
      if (T == 0) { for (const auto &k : b.keys) { log.os() << "WARNING (gmm-latgen-faster) Zero-length utterance: " << k << "\n"; num_err++; } continue; }
      int fdim = 0;
      std::unique_ptr<DevBuffer> d_fe;
      const float *d_feats_p = DeviceFeatures(fe, tab, u1 - b.keys.size(), u1, &fdim, &d_fe);
      if (fdim != model.dim)
        throw ApiError(fe.fused ? "Dim mismatch: front end gives dim " + std::to_string(fdim) + " vs. model dim = " + std::to_string(model.dim)
                                : "Dim mismatch: data dim = " + std::to_string(fdim) + " vs. model dim = " + std::to_string(model.dim));
      const int32_t ld = vbh_gmm_padded_pdfs(gmm);      // score rows on 128-byte lines (vbh_gmm_loglikes_ld); the search takes ld as its row length
      struct { void *p; } d_ll{g_latgen.Scores(ctx, sizeof(float) * (size_t)T * ld)};
      Check(vbh_gmm_loglikes_ld(gmm, d_feats_p, T, (float *)d_ll.p, ld), "vbh_gmm_loglikes");
      // the decoder (tens of GB of search workspace at the reference's batch sizes) is kept across the batches of the table
      // and only rebuilt when a batch needs more utterances or longer ones than it was made for
      need_graph();
      vbh_graph *g = cache.gh->g;
      struct { int32_t start; } graph{cache.graph_start};
      if (dec == nullptr || (int)b.keys.size() > dec_batch || max_T > dec_T) {
        if (dec) vbh_decoder_destroy(dec);
        dec = nullptr;
        dec_batch = std::max((int)b.keys.size(), dec_batch); dec_T = std::max(max_T, dec_T);
        Check(vbh_decoder_create(ctx, g, &ws_cfg, dec_batch, dec_T, &dec), "vbh_decoder_create");
        if (ws_cfg.max_active != cfg.max_active) Check(vbh_decoder_set_cfg(dec, &cfg), "vbh_decoder_set_cfg");
        lap("front end + decoder create");
      }
      const int rc = vbh_decode_batch(dec, (const float *)d_ll.p, ld, b.offs.data(), (int32_t)b.keys.size(),
                                      model.tid2pdf.data(), (int32_t)model.tid2pdf.size() - 1);
      if (rc != 0) throw ApiError(std::string("vbh_decode_batch: ") + vbh_last_error());
      lap("scores + search");
      // GetRawLattice (K/decoder/lattice-faster-decoder.cc:110-192): one state per token (state 0 =
      // the start token), arc weights (graph, acoustic - cost_offset), acoustic costs still scaled
      const size_t nu = b.keys.size();
      std::vector<vbh_lattice> Ls(nu);
      for (size_t u = 0; u < nu; u++) Check(vbh_decoder_result(dec, (int32_t)u, &Ls[u]), "vbh_decoder_result");
      auto usable = [&](const vbh_lattice &L) { return L.ok && (L.reached_final || allow_partial); };
      auto raw_lattice = [&](const vbh_lattice &L, vblat::Lattice *lat) { RawLatticeFromResult(L, graph.start, lat); };
      // "We'll write the lattice without acoustic scaling" (decoder-wrappers.cc:268-284)
      const double inv = cfg.acoustic_scale != 0.0f ? 1.0 / (double)cfg.acoustic_scale : 1.0;
      // determinization (DeterminizeLatticePhonePrunedWrapper, decoder-wrappers.cc:262-275) is host
      // work per utterance: the utterances of the batch are spread over the host's cores
      // Both are done here, with the entry's bytes, so that the writing loop below only appends to the archives.
      std::vector<std::string> entry(want_lat ? nu : 0);
      std::vector<int> det_ok(nu, 1);
      std::vector<std::string> det_err(nu);
      if (want_lat) {
        std::atomic<size_t> next{0};
        auto work = [&]() {
          for (size_t u = next++; u < nu; u = next++) {
            if (!usable(Ls[u])) continue;
            try {
              vblat::Lattice lat;
              raw_lattice(Ls[u], &lat);
              if (lat.st.empty()) throw ApiError("Unexpected problem getting lattice");
              vblat::Connect(&lat);
              if (determinize) {
                vblat::CompactLattice clat;
                det_ok[u] = vblat::DeterminizeLatticePhonePrunedWrapper(model.trans, &lat, cfg.lattice_beam, &clat, det_opts) ? 1 : 0;
                vblat::ScaleLattice(1.0, 0.0, 0.0, inv, &clat);
                vbio::SerializeLattice(clat, &entry[u]);
              } else {
                vblat::ScaleLattice(1.0, 0.0, 0.0, inv, &lat);
                vbio::SerializeLattice(lat, &entry[u]);
              }
            } catch (const std::exception &e) { det_err[u] = e.what(); if (det_err[u].empty()) det_err[u] = "error"; }
          }
        };
        const unsigned nt = std::max(1u, std::min((unsigned)nu, std::min(16u, std::thread::hardware_concurrency())));
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < nt; t++) pool.emplace_back(work);
        work();
        for (std::thread &t : pool) t.join();
      }
      lap("determinization");
      for (size_t u = 0; u < nu; u++) {
        const vbh_lattice &L = Ls[u];
        const std::string &utt = b.keys[u];
        if (b.offs[u + 1] == b.offs[u]) { log.os() << "WARNING (gmm-latgen-faster) Zero-length utterance: " << utt << "\n"; num_err++; continue; }
        // DecodeUtteranceLatticeFaster (K/decoder/decoder-wrappers.cc:198-293)
        if (!L.ok) { log.os() << "WARNING (gmm-latgen-faster) Failed to decode file " << utt << "\n"; num_err++; continue; }
        if (!L.reached_final) {
          if (allow_partial) log.os() << "WARNING (gmm-latgen-faster) Outputting partial output for utterance " << utt << " since no final-state reached\n";
          else { log.os() << "WARNING (gmm-latgen-faster) Not producing output for utterance " << utt << " since no final-state reached and --allow-partial=false.\n"; num_err++; continue; }
        }
        std::vector<int32_t> words(L.words, L.words + L.num_words), ali(L.ali, L.ali + L.num_ali);
        JobOut &jo = out_of(utt);
        if (jo.words.IsOpen()) jo.words.Write(utt, words);
        if (jo.ali.IsOpen()) jo.ali.Write(utt, ali);
        const double like = -((double)L.best_graph + (double)L.best_acoustic);
        if (want_lat) {
          if (!det_err[u].empty()) throw ApiError("lattice of " + utt + ": " + det_err[u]);
          if (!det_ok[u]) log.os() << "WARNING (gmm-latgen-faster) Determinization finished earlier than the beam for utterance " << utt << "\n";
          if (determinize) jo.clat.WriteSerialized(utt, std::move(entry[u])); else jo.lat.WriteSerialized(utt, std::move(entry[u]));
          std::string().swap(entry[u]);
        }
        log.os() << "LOG (gmm-latgen-faster) Log-like per frame for utterance " << utt << " is " << (like / (L.num_ali ? L.num_ali : 1))
                 << " over " << L.num_ali << " frames.\n";
        tot_like += like;
        frame_count += L.num_frames;
        num_done++;
      }
      lap("write results");
    }
    lap("end of the table");
    log.os() << "LOG (gmm-latgen-faster) Done " << num_done << " utterances, failed for " << num_err << "\n";
    log.os() << "LOG (gmm-latgen-faster) Overall log-likelihood per frame is " << (frame_count ? tot_like / frame_count : 0.0)
             << " over " << frame_count << " frames.\n";
    return num_done != 0 ? 0 : 1;
  });
}

// copy-feats (VB/src/featbin/copy-feats.cpp; K/featbin/copy-feats.cc:28-170): table to table copy with
// optional kSpeechFeature compression and a per-utterance frame count table
int CopyFeats(int argc, char *argv[], std::ofstream &file_log) {
  return Guard("copy-feats", &file_log, [&](Log &log) -> int {
    Options po;
    po.Know({"compress", "write-num-frames", "binary", "htk-in", "sphinx-in", "compression-method"});
    po.Parse(argc, argv);
    if (po.args.size() != 2) throw ApiError("wrong arguments.");
    if (po.Bool("htk-in", false) || po.Bool("sphinx-in", false)) throw ApiError("--htk-in / --sphinx-in are not supported");
    const bool compress = po.Bool("compress", false);
    vbio::MatrixReader reader(po.args[0]);
    vbio::MatrixWriter writer(po.args[1], compress);
    vbio::IntVectorWriter dummy;
    std::ofstream nf;
    if (po.Has("write-num-frames")) {
      const vbio::Wspec ws = vbio::ParseWspecifier(po.Str("write-num-frames", ""));
      if (!ws.has_ark || !ws.text) throw ApiError("--write-num-frames expects ark,t:<file>");
      nf.open(ws.ark_path);
      if (!nf) throw ApiError("cannot open " + ws.ark_path);
    }
    std::string key; vbio::Matrix m;
    int n = 0;
    while (reader.Next(&key, &m)) {
      writer.Write(key, m);
      if (nf.is_open()) nf << key << " " << m.rows << "\n";
      n++;
    }
    writer.Close();
    log.os() << "LOG (copy-feats) Copied " << n << " feature matrices.\n";
    return n != 0 ? 0 : 1;
  });
}

// gmm-decode-faster (K/gmmbin/gmm-decode-faster.cc:67-211; the VoiceBridge DLL does not wrap this
// program, so the upstream argv grammar and return convention are kept as they are):
//   gmm-decode-faster [options] model-in fst-in features-rspecifier words-wspecifier [alignments-wspecifier [lattice-wspecifier]]
// The 6th argument: the best path as a CompactLattice, built as the program builds it (:174-181): GetBestPath's linear lattice after
// RemoveEpsLocal (K/decoder/faster-decoder.cc:113-146, K/fstext/remove-eps-local-inl.h), acoustic costs un-scaled, ConvertLattice.
namespace {
// RemoveEpsLocal on a linear chain: every interior state has one arc in and one out, so only the reference's "pattern 2" applies
// (remove-eps-local-inl.h:229-283) and, visiting the states in order, it merges an arc with the arcs behind it for as long as the
// labels do not clash (CanCombineArcs :60-68: not two non-zero ilabels, not two non-zero olabels; the weights multiply, left to
// right); a trailing arc without labels folds into the final weight (CanCombineFinal :70-76).
vblat::Lattice LinearBestPathLattice(const vbh_best_path &bp) {
  std::vector<vblat::Arc> groups;
  int i = 0;
  while (i < bp.num_arcs) {
    vblat::Arc cur{bp.arc_ilabel[i], bp.arc_olabel[i], vblat::Weight{bp.arc_graph[i], bp.arc_acoustic[i]}, 0};
    i++;
    while (i < bp.num_arcs && !(cur.ilabel != 0 && bp.arc_ilabel[i] != 0) && !(cur.olabel != 0 && bp.arc_olabel[i] != 0)) {
      cur.w = vblat::Times(cur.w, vblat::Weight{bp.arc_graph[i], bp.arc_acoustic[i]});
      if (cur.ilabel == 0) cur.ilabel = bp.arc_ilabel[i];
      if (cur.olabel == 0) cur.olabel = bp.arc_olabel[i];
      i++;
    }
    groups.push_back(cur);
  }
  vblat::Weight fin = bp.reached_final ? vblat::Weight{bp.final_graph, 0.0f} : vblat::One();
  if (!groups.empty() && groups.back().ilabel == 0 && groups.back().olabel == 0) {      // (a path without a single label)
    fin = vblat::Times(groups.back().w, fin);
    groups.pop_back();
  }
  vblat::Lattice lat;
  lat.start = 0;
  lat.st.resize(groups.size() + 1);
  for (size_t k = 0; k < groups.size(); k++) { groups[k].next = (int32_t)k + 1; lat.st[k].arcs.push_back(groups[k]); }
  lat.st.back().final = fin;
  return lat;
}
}  // namespace

int GmmDecodeFaster(int argc, char *argv[], std::ofstream &file_log) {
  return Guard("gmm-decode-faster", &file_log, [&](Log &log) -> int {
    Options po;
    po.Know({"beam", "max-active", "min-active", "beam-delta", "hash-ratio", "acoustic-scale", "word-symbol-table",
             "allow-partial", "decode-batch", "admission"});      // (--admission=ordered|exact: an extension, as in gmm-latgen-faster above)
    po.Parse(argc, argv);
    if (po.args.size() < 4 || po.args.size() > 6) throw ApiError("wrong arguments.");
    vbh_faster_cfg cfg;
    vbh_faster_cfg_default(&cfg);
    {
      const std::string adm = po.Str("admission", "ordered");
      if (adm == "exact") cfg.admission = VBH_ADMISSION_EXACT;
      else if (adm == "ordered") cfg.admission = VBH_ADMISSION_ORDERED;
      else throw ApiError("--admission must be ordered or exact, not " + adm);
    }
    // FasterDecoderOptions defaults (K/decoder/faster-decoder.h:38-43), program default acoustic scale 0.1 (:79)
    cfg.beam = (float)po.Num("beam", 16.0);
    cfg.max_active = (int)po.Num("max-active", (double)std::numeric_limits<int32_t>::max());
    cfg.min_active = (int)po.Num("min-active", 20);
    cfg.beam_delta = (float)po.Num("beam-delta", 0.5);
    cfg.acoustic_scale = (float)po.Num("acoustic-scale", 0.1);
    const bool allow_partial = po.Bool("allow-partial", true);
    const bool laps = getenv("VBH_DEBUG_TIMES") != nullptr;      // diagnostic: where the host time of a job goes
    auto t_lap = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
      if (!laps) return;
      const auto now = std::chrono::steady_clock::now();
      fprintf(stderr, "gmm-latgen-faster lap %-28s %8.1f ms\n", what, 1e3 * std::chrono::duration<double>(now - t_lap).count());
      t_lap = now;
    };
    vbio::Model model;
    vbio::ReadModel(po.args[0], &model);
    lap("read model");
    vbio::Graph graph;
    vbio::ReadFst(po.args[1], &graph);
    lap("read graph");
    // ---- fused front end (extension, see the option list)
    const bool fused = po.Has("fe-cmvn");
    std::map<std::string, vbio::MatrixD> fe_stats;
    std::map<std::string, std::string> fe_utt2spk;
    std::map<std::string, vbio::Matrix> fe_trans;
    vbio::Matrix fe_lda;
    vbh_featpost_cfg fe_cfg{};
    if (fused) {
      fe_stats = vbio::ReadMatrixDTable(po.Str("fe-cmvn", ""));
      if (po.Has("fe-utt2spk")) fe_utt2spk = vbio::ReadStringMap(po.Str("fe-utt2spk", ""));
      fe_cfg.apply_cmvn = 1;
      fe_cfg.norm_vars = po.Bool("fe-norm-vars", false);
      const std::string type = po.Str("fe-type", "delta");
      if (type == "delta") { fe_cfg.mode = 1; fe_cfg.delta_order = (int)po.Num("fe-delta-order", 2); fe_cfg.delta_window = (int)po.Num("fe-delta-window", 2); }
      else if (type == "lda") {
        fe_cfg.mode = 2; fe_cfg.splice_left = (int)po.Num("fe-left-context", 4); fe_cfg.splice_right = (int)po.Num("fe-right-context", 4);
        if (!po.Has("fe-lda")) throw ApiError("--fe-type=lda needs --fe-lda=<final.mat>");
        std::ifstream f(po.Str("fe-lda", ""), std::ios::binary);
        if (!f) throw ApiError("cannot open " + po.Str("fe-lda", ""));
        const bool binary = vbio::ReadBinaryHeader(f);
        vbio::ReadMatrix(f, binary, &fe_lda);
        fe_cfg.lda_rows = fe_lda.rows; fe_cfg.lda_cols = fe_lda.cols;
      } else throw ApiError("--fe-type must be delta or lda");
      if (po.Has("fe-trans")) fe_trans = vbio::ReadMatrixTable(po.Str("fe-trans", ""));
    }
    vbh_ctx *ctx = Ctx();
    vbh_gmm *gmm = nullptr;
    Check(vbh_gmm_create(ctx, model.dim, model.num_pdfs, model.pdf_offsets.data(), model.weights.data(),
                         model.means_invvars.data(), model.inv_vars.data(), &gmm), "vbh_gmm_create");
    vbh_graph *g = nullptr;
    Check(vbh_graph_create(ctx, (int32_t)graph.final_cost.size(), graph.start, graph.final_cost.data(), graph.arc_offset.data(),
                           reinterpret_cast<const vbh_arc *>(graph.arcs.data()), &g), "vbh_graph_create");
    vbio::IntVectorWriter words_writer(po.args[3]), ali_writer;
    if (po.args.size() >= 5 && !po.args[4].empty()) ali_writer.Open(po.args[4]);
    vbio::CompactLatticeWriter clat_writer;
    if (po.args.size() == 6 && !po.args[5].empty()) clat_writer.Open(po.args[5]);
    const int batch_utts = (int)po.Num("decode-batch", 256);
    vbio::MatrixReader reader(po.args[2]);
    double tot_like = 0.0;
    int64_t frame_count = 0;
    int num_success = 0, num_fail = 0;
    std::string key; vbio::Matrix m;
    bool more = true;
    while (more) {
      Batch b;
      int max_T = 0;
      while ((int)b.keys.size() < batch_utts && (more = reader.Next(&key, &m))) {
        if (m.rows == 0) { log.os() << "WARNING (gmm-decode-faster) Zero-length utterance: " << key << "\n"; num_fail++; continue; }
        if (m.cols != model.dim) throw ApiError("Dim mismatch: data dim = " + std::to_string(m.cols) + " vs. model dim = " + std::to_string(model.dim));
        if (b.dim < 0) b.dim = m.cols;
        b.Add(key, m);
        if (m.rows > max_T) max_T = m.rows;
      }
      if (b.keys.empty()) break;
      lap("read features of a batch");
      const int64_t T = b.frames();
      const int32_t ld = vbh_gmm_padded_pdfs(gmm);
      DevMem d_in(ctx, sizeof(float) * b.data.size(), 1), d_ll(ctx, sizeof(float) * (size_t)T * ld, 5);
      Check(vbh_copy_h2d(ctx, d_in.p, b.data.data(), sizeof(float) * b.data.size()), "vbh_copy_h2d");
      std::unique_ptr<DevMem> d_fe;
      const float *d_feats_p = (const float *)d_in.p;
      if (fused) {
        // CMVN statistics and fMLLR transforms of the batch's speakers, packed in the order the speakers appear
        vbh_featpost_cfg cfg = fe_cfg;
        cfg.in_dim = b.dim;
        std::map<std::string, int> idx; std::vector<int32_t> u2s; std::vector<double> stats; std::vector<float> xf;
        for (const std::string &k : b.keys) {
          std::string spk = k;
          if (!fe_utt2spk.empty()) { auto it = fe_utt2spk.find(k); if (it != fe_utt2spk.end()) spk = it->second; }
          if (!idx.count(spk)) {
            auto it = fe_stats.find(spk);
            if (it == fe_stats.end()) throw ApiError("No normalization statistics available for key " + k);
            const vbio::MatrixD &ms = it->second;
            if (ms.cols != b.dim + 1 || ms.rows < 1 || ms.rows > 2) throw ApiError("Dim mismatch: cmvn " + std::to_string(ms.rows) + "x" + std::to_string(ms.cols));
            if (ms.rows == 1 && cfg.norm_vars) throw ApiError("You requested variance normalization but no variance stats are supplied.");
            stats.insert(stats.end(), ms.data.begin(), ms.data.end());
            if (ms.rows == 1) stats.insert(stats.end(), (size_t)(b.dim + 1), 0.0);
            if (!fe_trans.empty()) {
              auto tt = fe_trans.find(spk);
              if (tt == fe_trans.end()) throw ApiError("No fMLLR transform for speaker " + spk);
              if (cfg.spk_rows == 0) { cfg.spk_rows = tt->second.rows; cfg.spk_cols = tt->second.cols; }
              if (tt->second.rows != cfg.spk_rows || tt->second.cols != cfg.spk_cols) throw ApiError("fMLLR transforms of different sizes");
              xf.insert(xf.end(), tt->second.data.begin(), tt->second.data.end());
            }
            const int n = (int)idx.size(); idx[spk] = n;
          }
          u2s.push_back(idx[spk]);
        }
        if (!fe_trans.empty() && !idx.empty()) { cfg.spk_rows = fe_trans.begin()->second.rows; cfg.spk_cols = fe_trans.begin()->second.cols; }
        const int od = vbh_featpost_out_dim(&cfg);
        if (od != model.dim) throw ApiError("Dim mismatch: front end gives dim " + std::to_string(od) + " vs. model dim = " + std::to_string(model.dim));
        d_fe.reset(new DevMem(ctx, sizeof(float) * (size_t)T * od, 6));
        Check(vbh_featpost(ctx, &cfg, (const float *)d_in.p, b.offs.data(), (int32_t)b.keys.size(), u2s.data(), (int)idx.size(), stats.data(),
                           fe_lda.data.empty() ? nullptr : fe_lda.data.data(), xf.empty() ? nullptr : xf.data(), (float *)d_fe->p), "vbh_featpost");
        d_feats_p = (const float *)d_fe->p;
      }
      Check(vbh_gmm_loglikes_ld(gmm, d_feats_p, T, (float *)d_ll.p, ld), "vbh_gmm_loglikes");
      vbh_faster_decoder *dec = nullptr;
      Check(vbh_faster_decoder_create(ctx, g, &cfg, (int32_t)b.keys.size(), max_T, &dec), "vbh_faster_decoder_create");
      const int rc = vbh_faster_decode_batch(dec, (const float *)d_ll.p, ld, b.offs.data(), (int32_t)b.keys.size(),
                                             model.tid2pdf.data(), (int32_t)model.tid2pdf.size() - 1);
      if (rc != 0) { const std::string e = vbh_last_error(); vbh_faster_decoder_destroy(dec); throw ApiError("vbh_faster_decode_batch: " + e); }
      for (size_t u = 0; u < b.keys.size(); u++) {
        vbh_best_path bp;
        Check(vbh_faster_decoder_result(dec, (int32_t)u, &bp), "vbh_faster_decoder_result");
        const std::string &utt = b.keys[u];
        const int nrows = (int)(b.offs[u + 1] - b.offs[u]);
        if ((allow_partial || bp.reached_final) && bp.ok) {      // (:154-155)
          if (!bp.reached_final)
            log.os() << "WARNING (gmm-decode-faster) Decoder did not reach end-state, outputting partial traceback since --allow-partial=true\n";
          num_success++;
          frame_count += nrows;
          words_writer.Write(utt, std::vector<int32_t>(bp.words, bp.words + bp.num_words));
          if (ali_writer.IsOpen()) ali_writer.Write(utt, std::vector<int32_t>(bp.ali, bp.ali + bp.num_ali));
          if (clat_writer.IsOpen()) {          // "We'll write the lattice without acoustic scaling." (:175-181)
            vblat::Lattice lat = LinearBestPathLattice(bp);
            if (cfg.acoustic_scale != 0.0f) vblat::ScaleLattice(1.0, 0.0, 0.0, 1.0 / (double)cfg.acoustic_scale, &lat);
            vblat::CompactLattice clat;
            vblat::ConvertLattice(lat, &clat);
            clat_writer.Write(utt, clat);
          }
          const double like = -(double)bp.best_graph - (double)bp.best_acoustic;
          tot_like += like;
          log.os() << "LOG (gmm-decode-faster) Log-like per frame for utterance " << utt << " is " << (like / nrows) << " over " << nrows << " frames.\n";
        } else {
          num_fail++;
          log.os() << "WARNING (gmm-decode-faster) Did not successfully decode utterance " << utt << ", len = " << nrows << "\n";
        }
      }
      vbh_faster_decoder_destroy(dec);
    }
    vbh_graph_destroy(g);
    vbh_gmm_destroy(gmm);
    log.os() << "LOG (gmm-decode-faster) Done " << num_success << " utterances, failed for " << num_fail << "\n";
    log.os() << "LOG (gmm-decode-faster) Overall log-likelihood per frame is " << (frame_count ? tot_like / frame_count : 0.0)
             << " over " << frame_count << " frames.\n";
    return num_success != 0 ? 0 : 1;
  });
}

// =====================================================================================
namespace {
}  // namespace

extern "C" {
// Measurement hook (bench.py): determinize the lattices of the decoder's last batch exactly as
// GmmLatgenFaster does per utterance (Connect, DeterminizeLatticePhonePrunedWrapper, un-scaling),
// on `threads` host threads.  out6 = {seconds, input states, input arcs, output states, output arcs,
// utterances whose determinization stopped before the beam}.
int vb_determinize_decoder_results(void *decoder, int n_utt, int graph_start, int num_tids, const int32_t *tid_phone,
                                   const int32_t *tid_hmm_state, const uint8_t *tid_self_loop, float lattice_beam, float acoustic_scale,
                                   int threads, double *out6) {
  try {
    vblat::TransInfo tm;
    tm.phone.assign(tid_phone, tid_phone + num_tids + 1);
    tm.hmm_state.assign(tid_hmm_state, tid_hmm_state + num_tids + 1);
    tm.self_loop.assign(tid_self_loop, tid_self_loop + num_tids + 1);
    std::vector<vbh_lattice> Ls(n_utt);
    for (int u = 0; u < n_utt; u++) Check(vbh_decoder_result((vbh_decoder *)decoder, u, &Ls[u]), "vbh_decoder_result");
    if (const char *dump = getenv("VBH_DUMP_LATS")) {       // diagnostic: the first raw lattices as an archive (for profiling the host stage elsewhere)
      vbio::LatticeWriter w;
      w.Open(std::string("ark:") + dump);
      for (int u = 0; u < std::min(n_utt, 8); u++) {
        if (!Ls[u].ok) continue;
        vblat::Lattice lat;
        RawLatticeFromResult(Ls[u], graph_start, &lat);
        w.Write("utt" + std::to_string(u), lat);
      }
    }
    std::atomic<int> next{0};
    std::atomic<long long> in_s{0}, in_a{0}, out_s{0}, out_a{0}, early{0};
    std::string err;
    std::mutex mu;
    const double inv = acoustic_scale != 0.0f ? 1.0 / (double)acoustic_scale : 1.0;
    const auto t0 = std::chrono::steady_clock::now();
    auto work = [&]() {
      for (int u = next++; u < n_utt; u = next++) {
        if (!Ls[u].ok) continue;
        try {
          vblat::Lattice lat; vblat::CompactLattice clat;
          RawLatticeFromResult(Ls[u], graph_start, &lat);
          vblat::Connect(&lat);
          in_s += (long long)lat.st.size(); in_a += (long long)lat.NumArcs();
          if (!vblat::DeterminizeLatticePhonePrunedWrapper(tm, &lat, lattice_beam, &clat, vblat::PhoneDetOptions())) early++;
          vblat::ScaleLattice(1.0, 0.0, 0.0, inv, &clat);
          out_s += (long long)clat.st.size(); out_a += (long long)clat.NumArcs();
        } catch (const std::exception &e) { std::lock_guard<std::mutex> g(mu); err = e.what(); }
      }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < std::max(threads, 1); t++) pool.emplace_back(work);
    work();
    for (std::thread &t : pool) t.join();
    if (!err.empty()) throw ApiError(err);
    out6[0] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    out6[1] = (double)in_s; out6[2] = (double)in_a; out6[3] = (double)out_s; out6[4] = (double)out_a; out6[5] = (double)early;
    return 0;
  } catch (const std::exception &e) { fprintf(stderr, "ERROR (vb_determinize_decoder_results) %s\n", e.what()); return -1; }
}
// ---- the same as a pipeline stage: a pool of host threads that determinizes the lattices of batch k while
// the GPU searches batch k+1.  submit() copies the decoder's results into raw lattices (they are only valid
// until the decoder's next call) and queues one job per utterance; wait() blocks until the queue is empty and
// returns the totals since the last wait {seconds of worker time, input states, input arcs, output states,
// output arcs, stopped before the beam}.
// what RawLatticeFromResult reads of a search result, copied out of the decoder's buffers (which its next batch reuses): flat
// arrays, one allocation per utterance -- the state-level lattice with its vector per state is built by the worker that
// determinizes it, off the caller's critical path
struct RawSnap {
  std::vector<int32_t> i32;     // tok_frame prefix (frame 0, then a sentinel) | tok_state prefix | link src, dst, ilabel, olabel
  std::vector<float> f32;       // tok_final | link graph, acoustic
  int32_t n = 0, nl = 0, k0 = 0;
  void Take(const vbh_lattice &L) {
    n = L.num_toks; nl = L.num_links; k0 = 0;
    while (k0 < n && L.tok_frame[k0] == 0) k0++;
    i32.resize((size_t)2 * (k0 + 1) + (size_t)4 * nl);
    f32.resize((size_t)n + (size_t)2 * nl);
    int32_t *pi = i32.data();
    memcpy(pi, L.tok_frame, sizeof(int32_t) * k0); pi[k0] = 1; pi += k0 + 1;
    memcpy(pi, L.tok_state, sizeof(int32_t) * k0); pi[k0] = -1; pi += k0 + 1;
    for (const int32_t *src : {L.link_src, L.link_dst, L.link_ilabel, L.link_olabel}) { memcpy(pi, src, sizeof(int32_t) * nl); pi += nl; }
    float *pf = f32.data();
    memcpy(pf, L.tok_final, sizeof(float) * n); pf += n;
    memcpy(pf, L.link_graph, sizeof(float) * nl); pf += nl;
    memcpy(pf, L.link_acoustic, sizeof(float) * nl);
  }
  vbh_lattice View() const {
    vbh_lattice L{};
    L.ok = 1; L.num_toks = n; L.num_links = nl;
    const int32_t *pi = i32.data();
    L.tok_frame = pi; pi += k0 + 1;
    L.tok_state = pi; pi += k0 + 1;
    L.link_src = pi; L.link_dst = pi + nl; L.link_ilabel = pi + 2 * (size_t)nl; L.link_olabel = pi + 3 * (size_t)nl;
    const float *pf = f32.data();
    L.tok_final = pf; L.link_graph = pf + n; L.link_acoustic = pf + n + nl;
    return L;
  }
};
struct VbDetPool {
  std::vector<std::thread> workers;
  std::mutex mu;
  std::condition_variable cv, cv_done;
  std::deque<std::pair<RawSnap, int>> q;       // (result, the graph's start state)
  size_t in_flight = 0;
  bool stop = false;
  vblat::TransInfo tm;
  float lattice_beam = 6.0f;
  double inv = 1.0;
  double busy = 0.0, in_s = 0, in_a = 0, out_s = 0, out_a = 0, early = 0;
  std::string err;
};
void *vb_det_pool_create(int threads, int num_tids, const int32_t *tid_phone, const int32_t *tid_hmm_state, const uint8_t *tid_self_loop,
                         float lattice_beam, float acoustic_scale) {
  VbDetPool *p = new VbDetPool();
  p->tm.phone.assign(tid_phone, tid_phone + num_tids + 1);
  p->tm.hmm_state.assign(tid_hmm_state, tid_hmm_state + num_tids + 1);
  p->tm.self_loop.assign(tid_self_loop, tid_self_loop + num_tids + 1);
  p->lattice_beam = lattice_beam;
  p->inv = acoustic_scale != 0.0f ? 1.0 / (double)acoustic_scale : 1.0;
  for (int t = 0; t < std::max(threads, 1); t++)
    p->workers.emplace_back([p]() {
      for (;;) {
        std::pair<RawSnap, int> job;
        {
          std::unique_lock<std::mutex> lk(p->mu);
          p->cv.wait(lk, [p] { return p->stop || !p->q.empty(); });
          if (p->q.empty()) return;
          job = std::move(p->q.front());
          p->q.pop_front();
        }
        const auto t0 = std::chrono::steady_clock::now();
        double is = 0, ia = 0, os = 0, oa = 0, early = 0;
        std::string err;
        try {
          vblat::Lattice lat;
          RawLatticeFromResult(job.first.View(), job.second, &lat);
          RawSnap().i32.swap(job.first.i32); RawSnap().f32.swap(job.first.f32);
          vblat::CompactLattice clat;
          vblat::Connect(&lat);
          is = (double)lat.st.size(); ia = (double)lat.NumArcs();
          if (!vblat::DeterminizeLatticePhonePrunedWrapper(p->tm, &lat, p->lattice_beam, &clat, vblat::PhoneDetOptions())) early = 1;
          vblat::ScaleLattice(1.0, 0.0, 0.0, p->inv, &clat);
          os = (double)clat.st.size(); oa = (double)clat.NumArcs();
        } catch (const std::exception &e) { err = e.what(); }
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::lock_guard<std::mutex> lk(p->mu);
        p->busy += dt; p->in_s += is; p->in_a += ia; p->out_s += os; p->out_a += oa; p->early += early;
        if (!err.empty()) p->err = err;
        if (--p->in_flight == 0) p->cv_done.notify_all();
      }
    });
  return p;
}
int vb_det_pool_submit(void *pool, void *decoder, int n_utt, int graph_start) {
  VbDetPool *p = (VbDetPool *)pool;
  try {
    // the copy-out is itself spread over a few threads: it sits on the caller's critical path (the decoder's
    // result buffers are reused by its next batch)
    std::vector<vbh_lattice> Ls(n_utt);
    for (int u = 0; u < n_utt; u++) Check(vbh_decoder_result((vbh_decoder *)decoder, u, &Ls[u]), "vbh_decoder_result");
    std::vector<RawSnap> lats(n_utt);
    std::atomic<int> next{0};
    auto work = [&]() { for (int u = next++; u < n_utt; u = next++) if (Ls[u].ok) lats[u].Take(Ls[u]); };
    std::vector<std::thread> helpers;
    const int nh = std::min<int>(8, std::max<int>(1, (int)p->workers.size()));
    for (int t = 1; t < nh; t++) helpers.emplace_back(work);
    work();
    for (std::thread &t : helpers) t.join();
    std::lock_guard<std::mutex> lk(p->mu);
    for (int u = 0; u < n_utt; u++) if (Ls[u].ok) { p->q.emplace_back(std::move(lats[u]), graph_start); p->in_flight++; }
    p->cv.notify_all();
    return 0;
  } catch (const std::exception &e) { fprintf(stderr, "ERROR (vb_det_pool_submit) %s\n", e.what()); return -1; }
}
int vb_det_pool_wait(void *pool, double *out6) {
  VbDetPool *p = (VbDetPool *)pool;
  std::unique_lock<std::mutex> lk(p->mu);
  p->cv_done.wait(lk, [p] { return p->in_flight == 0; });
  out6[0] = p->busy; out6[1] = p->in_s; out6[2] = p->in_a; out6[3] = p->out_s; out6[4] = p->out_a; out6[5] = p->early;
  p->busy = p->in_s = p->in_a = p->out_s = p->out_a = p->early = 0;
  if (!p->err.empty()) { fprintf(stderr, "ERROR (vb_det_pool) %s\n", p->err.c_str()); p->err.clear(); return -1; }
  return 0;
}
void vb_det_pool_destroy(void *pool) {
  VbDetPool *p = (VbDetPool *)pool;
  { std::lock_guard<std::mutex> lk(p->mu); p->stop = true; }
  p->cv.notify_all();
  for (std::thread &t : p->workers) t.join();
  delete p;
}
// the calling thread's following gmm-latgen-faster calls make their search workspaces large enough for this max_active as well
int vb_set_workspace_max_active(int max_active) { g_workspace_max_active.store(max_active > 0 ? max_active : 0); return 0; }
int vb_get_workspace_max_active(void) { return g_workspace_max_active.load(); }
int vb_set_device(int device_id) { if (device_id != g_device) g_latgen.Release(); g_device = device_id; return 0; }
int vb_get_device(void) { return g_device; }
// frees the model, graph and search workspace the calling thread's last gmm-latgen-faster call left on the device
// (and the context's scratch buffers -- a job's score matrix and feature output among them, gigabytes per job thread: ADVICE r4)
void vb_release_device_cache(void) { g_latgen.Release(); ReleaseRawTableCache(); ReleaseDeviceGmmCache(); if (g_ctx) (void)vbh_ctx_scratch_release(g_ctx, -1); }
// ... and the thread's GPU context itself (stream, staging arenas): for a thread that is about to end
void vb_release_thread_context(void) {
  vb_release_device_cache();
  if (g_ctx) { vbh_ctx_destroy(g_ctx); g_ctx = nullptr; g_ctx_device = -1; }
}
int vb_ComputeMFCCFeats(int argc, char *argv[], const char *lp) { return WithLog(lp, [&](std::ofstream &l) { return ComputeMFCCFeats(argc, argv, l); }); }
int vb_ComputeCmvnStats(int argc, char *argv[], const char *lp) { return WithLog(lp, [&](std::ofstream &l) { return ComputeCmvnStats(argc, argv, l); }); }
int vb_ApplyCmvn(int argc, char *argv[], const char *lp) { return WithLog(lp, [&](std::ofstream &l) { return ApplyCmvn(argc, argv, l); }); }
int vb_AddDeltas(int argc, char *argv[]) { return AddDeltas(argc, argv); }
int vb_SpliceFeats(int argc, char *argv[]) { return SpliceFeats(argc, argv); }
int vb_TransformFeats(int argc, char *argv[], const char *lp) { return WithLog(lp, [&](std::ofstream &l) { return TransformFeats(argc, argv, l); }); }
int vb_GmmLatgenFaster(int argc, char *argv[], const char *lp) { return WithLog(lp, [&](std::ofstream &l) { return GmmLatgenFaster(argc, argv, l); }); }
int vb_CopyFeats(int argc, char *argv[], const char *lp) { return WithLog(lp, [&](std::ofstream &l) { return CopyFeats(argc, argv, l); }); }
int vb_GmmDecodeFaster(int argc, char *argv[], const char *lp) { return WithLog(lp, [&](std::ofstream &l) { return GmmDecodeFaster(argc, argv, l); }); }
}
