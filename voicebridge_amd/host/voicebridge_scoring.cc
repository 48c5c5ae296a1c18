// voicebridge_scoring.cc -- the per-utterance statistics ScoreKaldiWER leaves under scoring_kaldi/wer_details when `stats` is
// set (VB/scr/steps/scoring/score_kaldi_wer.cpp:165-254): the word alignment of every hypothesis with its transcript
// (align-text), per-utterance and per-speaker error counts, the table of word confusions, and the bootstrap confidence
// interval of the WER.  Host code, text in and text out; declarations and the reference files they mirror: voicebridge_api.h.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <iomanip>
#include <map>
#include <sstream>
#include <unordered_map>

#include "api_common.h"
#include "kaldi_io.h"
#include "voicebridge_api.h"

using namespace vbapi;

namespace {

typedef std::vector<std::string> Words;

// IsToken (K/util/text-utils.cc:105-117): non-empty, no ASCII control or space characters, not 0xFF
bool IsToken(const std::string &t) {
  if (t.empty()) return false;
  for (unsigned char c : t) if ((c < 33 || c == 127) || c == 255) return false;
  return true;
}

// LevenshteinAlignment (K/util/edit-distance-inl.h:130-195): the pairs (ref word or eps, hyp word or eps) of one cheapest
// alignment; on the way back a match / substitution is preferred to a deletion, a deletion to an insertion
void Align(const Words &a, const Words &b, const std::string &eps, std::vector<std::pair<std::string, std::string>> *out) {
  const size_t M = a.size(), N = b.size(), W = N + 1;
  std::vector<int32_t> e((M + 1) * W);
  for (size_t n = 0; n <= N; n++) e[n] = (int32_t)n;
  for (size_t m = 1; m <= M; m++) {
    e[m * W] = e[(m - 1) * W] + 1;
    for (size_t n = 1; n <= N; n++) {
      const int32_t diag = e[(m - 1) * W + n - 1] + (a[m - 1] == b[n - 1] ? 0 : 1), del = e[(m - 1) * W + n] + 1, ins = e[m * W + n - 1] + 1;
      e[m * W + n] = std::min(diag, std::min(del, ins));
    }
  }
  out->clear();
  size_t m = M, n = N;
  while (m != 0 || n != 0) {
    bool step_m, step_n;
    if (m == 0) { step_m = false; step_n = true; }
    else if (n == 0) { step_m = true; step_n = false; }
    else {
      const int32_t diag = e[(m - 1) * W + n - 1] + (a[m - 1] == b[n - 1] ? 0 : 1), del = e[(m - 1) * W + n] + 1, ins = e[m * W + n - 1] + 1;
      if (diag <= std::min(del, ins)) { step_m = step_n = true; }
      else if (del <= ins) { step_m = true; step_n = false; }
      else { step_m = false; step_n = true; }
    }
    out->emplace_back(step_m ? a[m - 1] : eps, step_n ? b[n - 1] : eps);
    m -= step_m; n -= step_n;
  }
  std::reverse(out->begin(), out->end());
}

// the edit distance alone (the bootstrap needs errors and words per utterance): two rows
int32_t Distance(const Words &a, const Words &b) {
  std::vector<int32_t> prev(b.size() + 1), cur(b.size() + 1);
  for (size_t n = 0; n <= b.size(); n++) prev[n] = (int32_t)n;
  for (size_t m = 1; m <= a.size(); m++) {
    cur[0] = prev[0] + 1;
    for (size_t n = 1; n <= b.size(); n++)
      cur[n] = std::min(prev[n - 1] + (a[m - 1] == b[n - 1] ? 0 : 1), std::min(prev[n], cur[n - 1]) + 1);
    std::swap(prev, cur);
  }
  return prev[b.size()];
}

// rand() of the C library the reference is built against here (glibc: the additive feedback generator r[i] = r[i-3] + r[i-31]
// behind random(), seeded with 1 as in a program that never calls srand), as a value: Kaldi's RandInt (K/base/kaldi-math.cc:
// min + Rand() % (max + 1 - min)) draws from the process-wide generator, which in a library would make the interval depend on
// whoever else called rand() before; with a generator of its own every call gives what the reference program prints.
class LibcRand {
 public:
  LibcRand() {
    int32_t r[344 + 34];
    r[0] = 1;
    for (int i = 1; i < 31; i++) { r[i] = (int32_t)((16807LL * r[i - 1]) % 2147483647LL); if (r[i] < 0) r[i] += 2147483647; }
    for (int i = 31; i < 34; i++) r[i] = r[i - 31];
    uint32_t u[344];
    for (int i = 0; i < 34; i++) u[i] = (uint32_t)r[i];
    for (int i = 34; i < 344; i++) u[i] = u[i - 31] + u[i - 3];
    for (int i = 0; i < 31; i++) ring_[i] = u[344 - 31 + i];
    pos_ = 0;
  }
  int32_t Next() {       // ring_[pos_] is r[i-31], ring_[(pos_+28) % 31] is r[i-3]
    const uint32_t v = ring_[pos_] + ring_[(pos_ + 28) % 31];
    ring_[pos_] = v;
    pos_ = (pos_ + 1) % 31;
    return (int32_t)(v >> 1);
  }
 private:
  uint32_t ring_[31];
  int pos_;
};

std::string Fixed2(double v) { std::ostringstream o; o << std::fixed << std::setprecision(2) << v; return o.str(); }
std::string PadLeft(const std::string &s, size_t w) { return s.size() >= w ? s : std::string(w - s.size(), ' ') + s; }
std::string PadRight(const std::string &s, size_t w) { return s.size() >= w ? s : s + std::string(w - s.size(), ' '); }
// centred, the odd space on the left (the layout the reference documents, wer_per_utt_details.cpp:30-44)
std::string Centre(const std::string &s, size_t w) {
  if (s.size() >= w) return s;
  const size_t right = (w - s.size()) / 2, left = w - s.size() - right;
  return std::string(left, ' ') + s + std::string(right, ' ');
}
std::vector<Words> ReadFields(const std::string &path, const char *delims) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw ApiError("cannot open " + path);
  std::vector<Words> rows;
  std::string line;
  while (std::getline(f, line)) {
    Words w;
    size_t i = 0;
    while (i < line.size()) {
      while (i < line.size() && (strchr(delims, line[i]) || line[i] == '\r')) i++;
      size_t j = i;
      while (j < line.size() && !strchr(delims, line[j]) && line[j] != '\r') j++;
      if (j > i) w.push_back(line.substr(i, j - i));
      i = j;
    }
    rows.push_back(std::move(w));
  }
  return rows;
}
std::string JoinWords(const Words &w, const char *sep) {
  std::string s;
  for (size_t i = 0; i < w.size(); i++) { if (i) s += sep; s += w[i]; }
  return s;
}

}  // namespace

// K/bin/align-text.cc:29-135; "ark:" text tables in, "ark,t:" out: "key ref hyp ; ref hyp ; ... \n"
int AlignText(int argc, char *argv[]) { std::ofstream none; return AlignTextLogged(argc, argv, none); }      // (messages to stderr)
int AlignTextLogged(int argc, char *argv[], std::ofstream &file_log) {
  return Guard("align-text", &file_log, [&](Log &log) -> int {
    Options po;
    po.Know({"special-symbol", "separator"});
    po.Parse(argc, argv);
    if (po.args.size() != 3) throw ApiError("Usage: align-text [options] <text1-rspecifier> <text2-rspecifier> <alignment-wspecifier>");
    const std::string eps = po.Str("special-symbol", "<eps>"), sep = po.Str("separator", ";");
    const auto text1 = vbio::ReadTokenTable(po.args[0]);
    std::map<std::string, Words> text2;
    for (auto &kv : vbio::ReadTokenTable(po.args[1])) text2[kv.first] = std::move(kv.second);
    std::string wpath = po.args[2];
    for (const char *pre : {"ark,t:", "ark:"}) if (wpath.compare(0, strlen(pre), pre) == 0) { wpath = wpath.substr(strlen(pre)); break; }
    std::ofstream out(wpath, std::ios::binary);
    if (!out) throw ApiError("cannot open " + wpath + " for writing");
    int n_done = 0, n_fail = 0;
    std::vector<std::pair<std::string, std::string>> aligned;
    for (const auto &kv : text1) {
      auto it = text2.find(kv.first);
      if (it == text2.end()) { log.os() << "WARNING (align-text) Key " << kv.first << " is in " << po.args[0] << ", but not in " << po.args[1] << "\n"; n_fail++; continue; }
      for (const Words *t : std::initializer_list<const Words *>{&kv.second, &it->second}) {
        if (std::find(t->begin(), t->end(), eps) != t->end()) throw ApiError("the utterance " + kv.first + " contains the special symbol " + eps);
        for (const std::string &w : *t) if (!IsToken(w)) throw ApiError("the utterance " + kv.first + " contains unprintable characters");
      }
      Align(kv.second, it->second, eps, &aligned);
      out << kv.first << ' ';
      for (size_t i = 0; i < aligned.size(); i++) {
        out << aligned[i].first << ' ' << aligned[i].second << ' ';
        if (i + 1 != aligned.size()) out << sep << ' ';
      }
      out << '\n';
      n_done++;
    }
    log.os() << "LOG (align-text) Done " << n_done << " sentences, failed for " << n_fail << "\n";
    return n_done != 0 ? 0 : 1;
  });
}

// K/bin/compute-wer-bootci.cc:29-254 with the reference wrapper's extra argument (VB/src/bin/compute-wer-bootci.cpp:209-212,
// 277-288): <ref-rspecifier> <hyp-rspecifier> <output file> [<hyp2-rspecifier>]
int ComputeWerBootci(int argc, char *argv[], std::ofstream &file_log) {
  return Guard("compute-wer-bootci", &file_log, [&](Log &log) -> int {
    Options po;
    po.Know({"mode", "replications"});
    po.Parse(argc, argv);
    if (po.args.size() < 3 || po.args.size() > 4) throw ApiError("Usage: compute-wer-bootci [options] <ref-rspecifier> <hyp-rspecifier> <output-file> [<hyp2-rspecifier>]");
    const std::string mode = po.Str("mode", "strict");
    if (mode != "strict" && mode != "present" && mode != "all") throw ApiError("--mode option invalid: expected \"present\"|\"all\"|\"strict\", got " + mode);
    const int replications = (int)po.Num("replications", 10000);
    if (replications < 1) throw ApiError("--replications must be positive");
    const bool dual = po.args.size() == 4;
    const auto ref = vbio::ReadTokenTable(po.args[0]);
    std::map<std::string, Words> hyp, hyp2;
    for (auto &kv : vbio::ReadTokenTable(po.args[1])) hyp[kv.first] = std::move(kv.second);
    if (dual) for (auto &kv : vbio::ReadTokenTable(po.args[3])) hyp2[kv.first] = std::move(kv.second);
    std::vector<std::pair<int32_t, int32_t>> e1, e2;       // (word errors, words) per utterance
    for (const auto &kv : ref) {
      const auto i1 = hyp.find(kv.first);
      const int32_t nw = (int32_t)kv.second.size();
      if (!dual) {
        if (i1 == hyp.end()) {
          if (mode == "strict") throw ApiError("No hypothesis for key " + kv.first + " and strict mode specifier.");
          if (mode == "present") continue;
        }
        e1.emplace_back(Distance(kv.second, i1 == hyp.end() ? Words() : i1->second), nw);
      } else {
        const auto i2 = hyp2.find(kv.first);
        if (i1 == hyp.end() || i2 == hyp2.end()) {
          if (mode == "strict") throw ApiError("No hypothesis for key " + kv.first + " in both transcripts comparison is not possible.");
          if (mode == "present") continue;
        }
        e1.emplace_back(i1 == hyp.end() ? nw : Distance(kv.second, i1->second), nw);
        e2.emplace_back(i2 == hyp2.end() ? nw : Distance(kv.second, i2->second), nw);
      }
    }
    if (e1.empty()) throw ApiError("no utterance to score");
    LibcRand rng;
    const int32_t n = (int32_t)e1.size();
    auto draw = [&]() { return n == 1 ? 0 : (int32_t)(rng.Next() % n); };      // RandInt(0, n - 1)
    auto interval = [&](const std::vector<std::pair<int32_t, int32_t>> &e, float *mean, float *half) {
      float acc = 0.0f, acc2 = 0.0f;
      for (int r = 0; r < replications; r++) {
        int32_t words = 0, errs = 0;
        for (int32_t j = 0; j < n; j++) { const int32_t p = draw(); errs += e[p].first; words += e[p].second; }
        const float wer = static_cast<float>(errs) / words;
        acc += wer; acc2 += wer * wer;
      }
      *mean = acc / replications;
      *half = 1.96 * sqrt(acc2 / replications - (*mean) * (*mean));
    };
    float mean1 = 0, half1 = 0, mean2 = 0, half2 = 0, p_improv = 0;
    interval(e1, &mean1, &half1);
    if (dual) {
      interval(e2, &mean2, &half2);
      int32_t better = 0;
      for (int r = 0; r < replications; r++) {
        int32_t d = 0;
        for (int32_t j = 0; j < n; j++) { const int32_t p = draw(); d += e1[p].first - e2[p].first; }
        better += d > 0;
      }
      p_improv = static_cast<float>(better) / replications;
    }
    std::ostringstream o;
    o << std::fixed << std::setprecision(2);
    o << "Set1: %WER " << 100 * mean1 << " 95% Conf Interval [ " << 100 * mean1 - 100 * half1 << ", " << 100 * mean1 + 100 * half1 << " ]\n";
    if (dual) {
      o << "Set2: %WER " << 100 * mean2 << " 95% Conf Interval [ " << 100 * mean2 - 100 * half2 << ", " << 100 * mean2 + 100 * half2 << " ]\n";
      o << "Probability of Set2 improving Set1: " << 100 * p_improv << "\n";
    }
    log.os() << o.str();
    if (!po.args[2].empty()) {
      std::ofstream f(po.args[2], std::ios::binary);
      if (!f) throw ApiError("Output file is not accessible " + po.args[2] + ".");
      f << std::fixed << std::setprecision(2);
      // (the reference writes the plus-minus sign as the single byte 0xB1 of its source file's code page)
      f << "Set1: WER " << 100 * mean1 << "% \xb1 " << 100 * half1 << "%\n";
      if (dual) {
        f << "Set2: WER " << 100 * mean2 << "% \xb1 " << 100 * half2 << "%\n";
        f << "Probability of Set2 improving Set1: " << 100 * p_improv << "\n";
      }
    }
    return 0;
  });
}

// VB/scr/utils/scoring/wer_per_utt_details.cpp:46-145 (after egs/wsj/s5/utils/scoring/wer_per_utt_details.pl): align-text output
// in, per utterance the lines "ref", "hyp", "op" (C / S / I / D under each pair, columns centred) and "#csid" counts out.
// The layout is the one the reference documents (:30-44); its own loop assigns hyp := ref (:93) and drops the padding its
// cjustify computes (:22-28), so every line it writes says "all correct" -- that is not reproduced.
int WerPerUttDetails(const std::string &in, const std::string &out, const std::string &special_symbol, const std::string &separator,
                     bool output_hyp, bool output_ref, bool output_ops, bool output_csid) {
  return Guard("wer_per_utt_details", nullptr, [&](Log &) -> int {
    const std::string delims = " \t" + separator;
    const std::vector<Words> rows = ReadFields(in, delims.c_str());
    std::ofstream f(out, std::ios::binary);
    if (!f) throw ApiError("Could not write to file " + out);
    for (const Words &r : rows) {
      if (r.empty()) continue;
      if (r.size() % 2 == 0) throw ApiError("Wrong line in input file (WerPerUttDetails): " + r[0]);
      int n_c = 0, n_s = 0, n_i = 0, n_d = 0;
      Words ref, hyp, op;
      for (size_t k = 1; k + 1 < r.size(); k += 2) {
        const std::string &rw = r[k], &hw = r[k + 1];
        const char *o;
        if (hw == special_symbol) { o = "D"; n_d++; }
        else if (rw == special_symbol) { o = "I"; n_i++; }
        else if (rw != hw) { o = "S"; n_s++; }
        else { o = "C"; n_c++; }
        const size_t w = std::max(rw.size(), hw.size());
        ref.push_back(Centre(rw, w)); hyp.push_back(Centre(hw, w)); op.push_back(Centre(o, w));
      }
      if (output_ref) f << r[0] << " ref  " << JoinWords(ref, "  ") << "\n";
      if (output_hyp) f << r[0] << " hyp  " << JoinWords(hyp, "  ") << "\n";
      if (output_ops) f << r[0] << " op   " << JoinWords(op, "  ") << "\n";
      if (output_csid) f << r[0] << " #csid " << n_c << " " << n_s << " " << n_i << " " << n_d << "\n";
    }
    return 0;
  });
}

// VB/scr/utils/scoring/wer_per_spk_details.cpp:85-232: the "#csid" lines summed per speaker (utt2spk), raw counts and
// percentages, a SUM block at the end; speakers in sorted order, the speaker column as wide as the longest name (>= SPK_WIDTH)
int WerPerSpkDetails(const std::string &in, const std::string &utt2spk, const std::string &out, int WIDTH, int SPK_WIDTH) {
  return Guard("wer_per_spk_details", nullptr, [&](Log &) -> int {
    struct Counts { long long c = 0, s = 0, i = 0, d = 0, sent = 0, serr = 0; };
    std::unordered_map<std::string, std::string> spk_of;
    std::map<std::string, Counts> per_spk;
    size_t spk_w = (size_t)std::max(SPK_WIDTH, 0);
    const size_t w = (size_t)std::max(WIDTH, 0);
    for (const Words &r : ReadFields(utt2spk, " \t")) {
      if (r.empty()) continue;
      if (r.size() != 2) throw ApiError("There should be 2 columns in the utt2spk file!");
      spk_of[r[0]] = r[1];
      per_spk[r[1]];
      spk_w = std::max(spk_w, r[1].size());
    }
    for (const Words &r : ReadFields(in, " \t")) {
      if (r.size() < 2 || r[1] != "#csid") continue;
      if (r.size() != 6) throw ApiError("Incompatible entry in " + in);
      long long v[4];
      for (int k = 0; k < 4; k++) {
        char *end = nullptr;
        v[k] = strtoll(r[2 + k].c_str(), &end, 10);
        if (end == r[2 + k].c_str() || *end || v[k] < 0) throw ApiError("Entry expected to be a number in " + in);
      }
      const auto it = spk_of.find(r[0]);
      if (it == spk_of.end()) throw ApiError("Utterance " + r[0] + " can not be found in utt2spk.");
      Counts &c = per_spk[it->second];
      c.c += v[0]; c.s += v[1]; c.i += v[2]; c.d += v[3]; c.sent++;
      if (v[1] + v[2] + v[3] != 0) c.serr++;
    }
    std::ofstream f(out, std::ios::binary);
    if (!f) throw ApiError("Could not write to file " + out);
    auto N = [&](long long x) { return PadRight(std::to_string(x), w); };
    auto P = [&](double x) { return PadRight(Fixed2(x), w); };
    f << PadLeft("SPEAKER", spk_w) << " id  " << PadRight("#SENT", w) << " " << PadRight("#WORD", w) << " " << PadRight("Corr", w) << " " << PadRight("Sub", w)
      << " " << PadRight("Ins", w) << " " << PadRight("Del", w) << " " << PadRight("Err", w) << " " << PadRight("S.Err", w) << "\n";
    auto block = [&](const std::string &name, const Counts &c) {
      const long long word = c.c + c.s + c.d, err = c.s + c.d + c.i;
      f << PadLeft(name, spk_w) << " raw " << N(c.sent) << " " << N(word) << " " << N(c.c) << " " << N(c.s) << " " << N(c.i) << " " << N(c.d) << " " << N(err) << " "
        << N(c.serr) << "\n";
      if (word != 0) {
        const double wd = (double)word;
        f << PadLeft(name, spk_w) << " sys " << N(c.sent) << " " << N(word) << " " << P(100.0 * c.c / wd) << " " << P(100.0 * c.s / wd) << " " << P(100.0 * c.i / wd) << " "
          << P(100.0 * c.d / wd) << " " << P(100.0 * err / wd) << " " << P(100.0 * c.serr / (double)c.sent) << "\n";
      }
    };
    Counts sum;
    for (const auto &kv : per_spk) {
      block(kv.first, kv.second);
      sum.c += kv.second.c; sum.s += kv.second.s; sum.i += kv.second.i; sum.d += kv.second.d; sum.sent += kv.second.sent; sum.serr += kv.second.serr;
    }
    block("SUM", sum);
    return 0;
  });
}

// VB/scr/utils/scoring/wer_ops_details.cpp:36-193 + the sort ScoreKaldiWER applies to its output (score_kaldi_wer.cpp:203-210):
// how often each reference word was recognised as each hypothesis word, "correct | substitution | insertion | deletion  ref  hyp
// count", one space between the fields, by kind of edit and then by falling count (equal counts: reference word, hypothesis word)
int WerOpsDetails(const std::string &in, const std::string &out, const std::string &special_symbol) {
  return Guard("wer_ops_details", nullptr, [&](Log &) -> int {
    std::map<std::string, std::pair<Words, Words>> utt;      // utterance -> (ref fields, hyp fields)
    std::map<std::string, int> seen;
    for (Words &r : ReadFields(in, " \t")) {
      if (r.size() < 2 || (r[1] != "hyp" && r[1] != "ref")) continue;
      const int bit = r[1] == "ref" ? 1 : 2;
      if (seen[r[0]] & bit) throw ApiError("The input stream contains duplicate entry " + r[0] + " " + r[1]);
      seen[r[0]] |= bit;
      Words fields(r.begin() + 2, r.end());
      (bit == 1 ? utt[r[0]].first : utt[r[0]].second) = std::move(fields);
    }
    std::map<std::pair<std::string, std::string>, long long> ops;
    for (const auto &kv : utt) {
      if (seen[kv.first] != 3) throw ApiError(std::string("The input stream does not contain entry '") + ((seen[kv.first] & 1) ? "hyp" : "ref") + "' for utterance " + kv.first);
      const Words &ref = kv.second.first, &hyp = kv.second.second;
      if (ref.size() != hyp.size()) throw ApiError("The 'ref' an 'hyp' entries do not have the same number of fields.");
      for (size_t i = 0; i < ref.size(); i++) ops[{ref[i], hyp[i]}]++;
    }
    struct Row { const char *kind; const std::string *ref, *hyp; long long n; };
    std::vector<Row> rows;
    for (const auto &kv : ops) {
      const std::string &r = kv.first.first, &h = kv.first.second;
      rows.push_back({r == h ? "correct" : r == special_symbol ? "insertion" : h == special_symbol ? "deletion" : "substitution", &r, &h, kv.second});
    }
    std::stable_sort(rows.begin(), rows.end(), [](const Row &a, const Row &b) { const int c = strcmp(a.kind, b.kind); return c != 0 ? c < 0 : a.n > b.n; });
    std::ofstream f(out, std::ios::binary);
    if (!f) throw ApiError("Could not write to file " + out);
    for (const Row &r : rows) f << r.kind << " " << *r.ref << " " << *r.hyp << " " << r.n << "\n";
    return 0;
  });
}

extern "C" {
int vb_AlignText(int argc, char *argv[]) { return AlignText(argc, argv); }
int vb_ComputeWerBootci(int argc, char *argv[], const char *lp) { return WithLog(lp, [&](std::ofstream &l) { return ComputeWerBootci(argc, argv, l); }); }
int vb_WerPerUttDetails(const char *in, const char *out, const char *special_symbol) { return WerPerUttDetails(in, out, special_symbol ? special_symbol : "<eps>"); }
int vb_WerPerSpkDetails(const char *in, const char *utt2spk, const char *out) { return WerPerSpkDetails(in, utt2spk, out); }
int vb_WerOpsDetails(const char *in, const char *out, const char *special_symbol) { return WerOpsDetails(in, out, special_symbol ? special_symbol : "<eps>"); }
}
