// voicebridge_steps.cc -- see voicebridge_steps.h.  Directory conventions only: every number
// comes out of the bin functions (voicebridge_api.cc), i.e. out of the HIP library.
#include "voicebridge_steps.h"

#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <vector>

#include "voicebridge_api.h"

namespace {

thread_local std::string g_err;

struct StepError : public std::runtime_error {
  explicit StepError(const std::string &m) : std::runtime_error(m) {}
};

bool Exists(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0; }
bool IsDir(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode); }
void MakeDir(const std::string &p) {
  if (IsDir(p)) return;
  if (mkdir(p.c_str(), 0777) != 0 && !IsDir(p)) throw StepError("cannot create directory " + p);
}
void MakeDirs(const std::string &p) {
  for (size_t i = 1; i <= p.size(); i++)
    if (i == p.size() || p[i] == '/') MakeDir(p.substr(0, i));
}
std::string Join(const std::string &a, const std::string &b) { return a.empty() || a.back() == '/' ? a + b : a + "/" + b; }
std::string Stem(std::string p) {          // fs::path::stem() of a directory path
  while (p.size() > 1 && p.back() == '/') p.pop_back();
  const size_t s = p.find_last_of('/');
  std::string name = s == std::string::npos ? p : p.substr(s + 1);
  const size_t d = name.find_last_of('.');
  if (d != std::string::npos && d != 0) name = name.substr(0, d);
  return name;
}
std::string Parent(std::string p) {
  while (p.size() > 1 && p.back() == '/') p.pop_back();
  const size_t s = p.find_last_of('/');
  return s == std::string::npos ? "." : (s == 0 ? "/" : p.substr(0, s));
}
std::vector<std::string> ReadLines(const std::string &path, bool required = true) {
  std::ifstream f(path);
  if (!f) { if (required) throw StepError(path + " is required but can not be found."); return {}; }
  std::vector<std::string> out; std::string line;
  while (std::getline(f, line)) if (!line.empty()) out.push_back(line);
  return out;
}
void WriteLines(const std::string &path, const std::vector<std::string> &lines) {
  std::ofstream f(path, std::ios::binary);
  if (!f) throw StepError("cannot write " + path);
  for (const auto &l : lines) f << l << "\n";
}
std::string FirstField(const std::string &line) { return line.substr(0, line.find_first_of(" \t")); }
std::string SecondField(const std::string &line) {
  std::istringstream is(line); std::string a, b; is >> a >> b; return b;
}
std::string FirstLine(const std::string &path) { const auto l = ReadLines(path, false); return l.empty() ? "" : l[0]; }
std::vector<std::string> SplitWords(const std::string &s) {
  std::istringstream is(s); std::vector<std::string> w; std::string x;
  while (is >> x) w.push_back(x);
  return w;
}

// argv for a bin function
struct Argv {
  std::vector<std::string> s; std::vector<char *> p;
  explicit Argv(std::vector<std::string> a) : s(std::move(a)) { for (auto &x : s) p.push_back(&x[0]); }
  int argc() { return (int)p.size(); }
  char **argv() { return p.data(); }
};

// split_scp.pl: contiguous pieces, the first (n % k) one line longer (VB/scr/utils/split_scp.cpp)
std::vector<std::vector<std::string>> SplitLines(const std::vector<std::string> &lines, int k) {
  std::vector<std::vector<std::string>> out(k);
  const size_t n = lines.size(), base = n / k, extra = n % k;
  size_t pos = 0;
  for (int j = 0; j < k; j++) {
    const size_t len = base + ((size_t)j < extra ? 1 : 0);
    out[j].assign(lines.begin() + pos, lines.begin() + pos + len);
    pos += len;
  }
  return out;
}

template <class F>
int Run(F f) {
  try { f(); return 0; }
  catch (const std::exception &e) { g_err = e.what(); fprintf(stderr, "ERROR (voicebridge steps) %s\n", e.what()); return -1; }
}

}  // namespace

// ------------------------------------------------------------------------------------ MakeMfcc
int MakeMfcc(const std::string &datadir, const std::string &mfcc_config, int nj, bool compress, bool write_utt2num_frames) {
  return Run([&] {
    if (nj < 1) throw StepError("nj must be >= 1");
    const std::string logdir = Join(datadir, "log"), mfccdir = Join(datadir, "mfcc"), name = Stem(datadir);
    MakeDir(logdir); MakeDir(mfccdir);
    if (Exists(Join(datadir, "feats.scp"))) {          // make_mfcc.cpp:38-45
      MakeDir(Join(datadir, "backup"));
      WriteLines(Join(Join(datadir, "backup"), "feats.scp"), ReadLines(Join(datadir, "feats.scp")));
      unlink(Join(datadir, "feats.scp").c_str());
    }
    if (ReadLines(mfcc_config, false).empty() && !Exists(mfcc_config)) throw StepError(mfcc_config + " is required but can not be found.");
    const std::string scp = Join(datadir, "wav.scp");
    const std::vector<std::string> wav = ReadLines(scp);
    if (wav.empty()) throw StepError(scp + " is empty.");
    if (Exists(Join(datadir, "segments"))) throw StepError("a segments file (extract-segments) is not part of this build");
    if (Exists(Join(datadir, "spk2warp")) || Exists(Join(datadir, "utt2warp"))) throw StepError("VTLN warp maps are not part of this build");
    const auto pieces = SplitLines(wav, nj);
    for (int j = 1; j <= nj; j++) {
      const std::string J = std::to_string(j);
      const std::string piece = Join(logdir, "wav_" + name + "." + J + ".scp"), temp = Join(logdir, "wav_" + name + "." + J + ".mfctemp");
      WriteLines(piece, pieces[j - 1]);
      std::ofstream log(Join(logdir, "make_mfcc_" + name + "." + J + ".log"), std::ios::binary);
      if (pieces[j - 1].empty()) continue;
      Argv a1({"compute-mfcc-feats", "--config=" + mfcc_config, "scp,p:" + piece, "ark:" + temp});
      if (ComputeMFCCFeats(a1.argc(), a1.argv(), log) < 0) throw StepError("compute-mfcc-feats failed for job " + J + ", see " + logdir);
      std::vector<std::string> c = {"copy-feats"};
      if (write_utt2num_frames) c.push_back("--write-num-frames=ark,t:" + Join(logdir, "utt2num_frames." + J));
      c.push_back(std::string("--compress=") + (compress ? "true" : "false"));
      c.push_back("ark:" + temp);
      c.push_back("ark,scp:" + Join(mfccdir, "raw_mfcc_" + name + "." + J + ".ark") + "," + Join(mfccdir, "raw_mfcc_" + name + "." + J + ".scp"));
      Argv a2(c);
      if (CopyFeats(a2.argc(), a2.argv(), log) < 0) throw StepError("copy-feats failed for job " + J + ", see " + logdir);
    }
    std::vector<std::string> feats, u2n;
    for (int j = 1; j <= nj; j++) {
      const std::string J = std::to_string(j);
      const auto part = ReadLines(Join(mfccdir, "raw_mfcc_" + name + "." + J + ".scp"), false);
      feats.insert(feats.end(), part.begin(), part.end());
      if (write_utt2num_frames) { const auto p2 = ReadLines(Join(logdir, "utt2num_frames." + J), false); u2n.insert(u2n.end(), p2.begin(), p2.end()); }
      unlink(Join(logdir, "utt2num_frames." + J).c_str());
      unlink(Join(logdir, "wav_" + name + "." + J + ".mfctemp").c_str());
      unlink(Join(logdir, "wav_" + name + "." + J + ".scp").c_str());
    }
    WriteLines(Join(datadir, "feats.scp"), feats);
    if (write_utt2num_frames) WriteLines(Join(datadir, "utt2num_frames"), u2n);
    const size_t nf = feats.size(), nu = ReadLines(Join(datadir, "utt2spk")).size();
    if (nf != nu) throw StepError("not all of the feature files were successfully processed (" + std::to_string(nf) + " != " + std::to_string(nu) + ")");
  });
}

// ------------------------------------------------------------------------------------ ComputeCmvnStats
int ComputeCmvnStats(const std::string &datadir, bool fake, bool two_channel, const std::string &fake_dims) {
  return Run([&] {
    if (fake || two_channel || !fake_dims.empty()) throw StepError("fake / two-channel / fake-dims CMVN statistics are not part of this build");
    const std::string logdir = Join(datadir, "log"), cmvndir = Join(datadir, "data"), name = Stem(datadir);
    MakeDir(cmvndir); MakeDir(logdir);
    if (!Exists(Join(datadir, "feats.scp"))) throw StepError(Join(datadir, "feats.scp") + " is required but can not be found.");
    if (!Exists(Join(datadir, "spk2utt"))) throw StepError(Join(datadir, "spk2utt") + " is required but can not be found.");
    const std::string ark = Join(cmvndir, "cmvn_" + name + ".ark"), scp = Join(cmvndir, "cmvn_" + name + ".scp");
    std::ofstream log(Join(cmvndir, "cmvn_" + name + ".log"), std::ios::binary);
    Argv a({"compute-cmvn-stats", "--spk2utt=ark:" + Join(datadir, "spk2utt"), "scp:" + Join(datadir, "feats.scp"), "ark,scp:" + ark + "," + scp});
    // (the bin function of the same name: voicebridge_api.h)
    if (::ComputeCmvnStats(a.argc(), a.argv(), log) < 0) throw StepError("Error computing CMVN stats.");
    WriteLines(Join(datadir, "cmvn.scp"), ReadLines(scp));                  // compute_cmvn_stats.cpp:171-176
    const size_t nc = ReadLines(Join(datadir, "cmvn.scp")).size(), nu = ReadLines(Join(datadir, "spk2utt")).size();
    if (nc != nu) fprintf(stderr, "WARNING (voicebridge steps) it seems not all of the speakers got cmvn stats (%zu != %zu)\n", nc, nu);
  });
}

// ------------------------------------------------------------------------------------ SplitData
int SplitData(const std::string &data, int numsplit, bool per_utt) {
  return Run([&] {
    if (numsplit <= 0) throw StepError("Invalid num-split argument " + std::to_string(numsplit) + ".");
    const std::vector<std::string> utt2spk = ReadLines(Join(data, "utt2spk"));
    const std::string root = Join(data, "split" + std::to_string(numsplit) + (per_utt ? "utt" : ""));
    // groups that must stay together: speakers (consecutive utt2spk lines of one speaker) or single utterances
    std::vector<std::vector<std::string>> groups;
    std::string last;
    for (const auto &l : utt2spk) {
      const std::string spk = SecondField(l);
      if (per_utt || groups.empty() || spk != last) groups.emplace_back();
      groups.back().push_back(l);
      last = spk;
    }
    // contiguous pieces of groups with utterance counts as equal as the group sizes allow (split_scp.pl --utt2spk)
    std::vector<std::vector<std::string>> piece(numsplit);
    const double per = (double)utt2spk.size() / numsplit;
    size_t gi = 0, taken = 0;
    for (int j = 0; j < numsplit; j++) {
      const double target = per * (j + 1);
      const size_t groups_left_for_others = (size_t)(numsplit - j - 1);
      while (gi < groups.size() && (groups.size() - gi) > groups_left_for_others &&
             (piece[j].empty() || taken + groups[gi].size() / 2.0 <= target || j == numsplit - 1)) {
        piece[j].insert(piece[j].end(), groups[gi].begin(), groups[gi].end());
        taken += groups[gi].size();
        gi++;
      }
    }
    MakeDirs(root);
    for (int j = 1; j <= numsplit; j++) {
      const std::string dsn = Join(root, std::to_string(j));
      MakeDir(dsn);
      WriteLines(Join(dsn, "utt2spk"), piece[j - 1]);
      std::map<std::string, bool> utts, spks;
      std::vector<std::string> spk_order; std::map<std::string, std::string> spk2utt;
      for (const auto &l : piece[j - 1]) {
        const std::string u = FirstField(l), sp = SecondField(l);
        utts[u] = true;
        if (!spks.count(sp)) { spks[sp] = true; spk_order.push_back(sp); spk2utt[sp] = sp; }
        spk2utt[sp] += " " + u;
      }
      std::vector<std::string> s2u;
      for (const auto &sp : spk_order) s2u.push_back(spk2utt[sp]);
      WriteLines(Join(dsn, "spk2utt"), s2u);
      for (const char *f : {"feats.scp", "text", "vad.scp", "utt2lang", "utt2dur", "utt2num_frames", "wav.scp"}) {
        if (!Exists(Join(data, f))) continue;
        std::vector<std::string> out;
        for (const auto &l : ReadLines(Join(data, f))) if (utts.count(FirstField(l))) out.push_back(l);
        WriteLines(Join(dsn, f), out);
      }
      for (const char *f : {"spk2gender", "spk2warp", "cmvn.scp"}) {
        if (!Exists(Join(data, f))) continue;
        std::vector<std::string> out;
        for (const auto &l : ReadLines(Join(data, f))) if (spks.count(FirstField(l))) out.push_back(l);
        WriteLines(Join(dsn, f), out);
      }
    }
  });
}

// ------------------------------------------------------------------------------------ Decode
int Decode(const std::string &graph_dir, const std::string &data_dir, const std::string &decode_dir, const std::string &model_in,
           const std::string &trans_dir, UMAPSS &, UMAPSS &, const std::string &iter, int nj, float acwt, int stage, int max_active,
           double beam, double lattice_beam, bool skip_scoring) {
  return Run([&] {
    if (!skip_scoring) throw StepError("scoring is not part of this build: call Decode with skip_scoring = true");
    if (!trans_dir.empty()) throw StepError("supplying fMLLR transforms (trans_dir) goes through DecodeFmllr, which is not part of this build");
    if (nj < 1) throw StepError("nj must be >= 1");
    (void)stage;
    const std::string srcdir = Parent(decode_dir);                  // the model directory (decode_gmm.cpp:80)
    std::string model = model_in;
    if (model.empty()) model = Join(srcdir, (iter.empty() ? std::string("final") : iter) + ".mdl");
    for (const std::string &f : {Join(data_dir, "feats.scp"), Join(data_dir, "cmvn.scp"), model, Join(graph_dir, "HCLG.fst")})
      if (!Exists(f)) throw StepError(f + " is required but can not be found.");
    const bool lda = Exists(Join(srcdir, "final.mat"));
    const std::string splice_opts = FirstLine(Join(srcdir, "splice_opts")), cmvn_opts = FirstLine(Join(srcdir, "cmvn_opts")),
                      delta_opts = FirstLine(Join(srcdir, "delta_opts"));
    MakeDirs(decode_dir); MakeDir(Join(decode_dir, "log"));
    WriteLines(Join(decode_dir, "num_jobs"), {std::to_string(nj)});
    if (SplitData(data_dir, nj, false) < 0) throw StepError("SplitData failed: " + g_err);
    const std::string sdata = Join(data_dir, "split" + std::to_string(nj));
    for (int j = 1; j <= nj; j++) {
      const std::string J = std::to_string(j), dsn = Join(sdata, J);
      unlink(Join(decode_dir, "lat." + J).c_str());
      std::ofstream log(Join(Join(decode_dir, "log"), "decode." + J + ".log"), std::ios::binary);
      if (ReadLines(Join(dsn, "feats.scp"), false).empty()) continue;
      const std::string t_cmvn = Join(dsn, "apply_cmvn.temp"), t_feat = Join(dsn, "feats.temp");
      std::vector<std::string> c = {"apply-cmvn"};
      for (const auto &o : SplitWords(cmvn_opts)) c.push_back(o);
      c.push_back("--utt2spk=ark:" + Join(dsn, "utt2spk"));
      c.push_back("scp:" + Join(dsn, "cmvn.scp")); c.push_back("scp:" + Join(dsn, "feats.scp")); c.push_back("ark:" + t_cmvn);
      Argv a1(c);
      if (ApplyCmvn(a1.argc(), a1.argv(), log) < 0) throw StepError("apply-cmvn failed for job " + J);
      if (!lda) {
        std::vector<std::string> dl = {"add-deltas"};
        for (const auto &o : SplitWords(delta_opts)) dl.push_back(o);
        dl.push_back("ark:" + t_cmvn); dl.push_back("ark:" + t_feat);
        Argv a2(dl);
        if (AddDeltas(a2.argc(), a2.argv()) < 0) throw StepError("add-deltas failed for job " + J);
      } else {
        const std::string t_spl = Join(dsn, "splice.temp");
        std::vector<std::string> sp = {"splice-feats"};
        for (const auto &o : SplitWords(splice_opts)) sp.push_back(o);
        sp.push_back("ark:" + t_cmvn); sp.push_back("ark:" + t_spl);
        Argv a2(sp);
        if (SpliceFeats(a2.argc(), a2.argv()) < 0) throw StepError("splice-feats failed for job " + J);
        Argv a3({"transform-feats", Join(srcdir, "final.mat"), "ark:" + t_spl, "ark:" + t_feat});
        if (TransformFeats(a3.argc(), a3.argv(), log) < 0) throw StepError("transform-feats failed for job " + J);
        unlink(t_spl.c_str());
      }
      std::vector<std::string> g = {"gmm-latgen-faster", "--max-active=" + std::to_string(max_active), "--beam=" + std::to_string(beam),
                                    "--lattice-beam=" + std::to_string(lattice_beam), "--acoustic-scale=" + std::to_string(acwt),
                                    "--allow-partial=true", "--determinize-lattice=false"};
      if (Exists(Join(graph_dir, "words.txt"))) g.push_back("--word-symbol-table=" + Join(graph_dir, "words.txt"));
      g.push_back(model); g.push_back(Join(graph_dir, "HCLG.fst")); g.push_back("ark:" + t_feat); g.push_back("ark:" + Join(decode_dir, "lat." + J));
      Argv a4(g);
      const int rc = GmmLatgenFaster(a4.argc(), a4.argv(), log);
      unlink(t_cmvn.c_str()); unlink(t_feat.c_str());
      if (rc < 0) throw StepError("gmm-latgen-faster failed for job " + J + ", see " + Join(decode_dir, "log"));
    }
  });
}

extern "C" {
int vb_MakeMfcc(const char *datadir, const char *mfcc_config, int nj, int compress, int write_utt2num_frames) {
  return MakeMfcc(datadir, mfcc_config, nj, compress != 0, write_utt2num_frames != 0);
}
int vb_ComputeCmvnStatsDir(const char *datadir) { return ComputeCmvnStats(std::string(datadir)); }
int vb_SplitData(const char *datadir, int numsplit, int per_utt) { return SplitData(datadir, numsplit, per_utt != 0); }
int vb_Decode(const char *graph_dir, const char *data_dir, const char *decode_dir, const char *model, int nj, float acwt, int max_active,
              double beam, double lattice_beam, int skip_scoring) {
  UMAPSS a, b;
  return Decode(graph_dir, data_dir, decode_dir, model ? model : "", "", a, b, "", nj, acwt, 0, max_active, beam, lattice_beam, skip_scoring != 0);
}
const char *vb_steps_last_error(void) { return g_err.c_str(); }
}
