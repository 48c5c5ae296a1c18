// voicebridge_steps.cc -- see voicebridge_steps.h.  Directory conventions only: every number
// comes out of the bin functions (voicebridge_api.cc), i.e. out of the HIP library.
#include "voicebridge_steps.h"

#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <regex>
#include <sstream>
#include <stdexcept>
#include <thread>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <vector>
#include <cctype>
#include <cstring>
#include <memory>

#include "api_common.h"
#include "kaldi_io.h"
#include "lattice.h"
#include "voicebridge_api.h"

namespace {

thread_local std::string g_err;
thread_local int g_step_depth = 0;     // > 0 while a step runs inside another (DecodeFmllr -> Decode)
// Model, graph and search workspace stay on the device from one step to the next (an application calls Decode, then DecodeFmllr,
// then the next data set: re-creating a 30 GB workspace after it has been freed cost 2.3 s of DecodeFmllr's 7.9 s on 1024 utterances,
// r02).  vb_release_device_cache() frees them; vb_set_release_device_cache(1) / VB_RELEASE_DEVICE_CACHE=1: every outermost step does.
std::atomic<int> g_release_at_step_end{-1};
bool ReleaseAtStepEnd() {
  int v = g_release_at_step_end.load();
  if (v < 0) { const char *e = getenv("VB_RELEASE_DEVICE_CACHE"); v = (e && *e && *e != '0') ? 1 : 0; g_release_at_step_end.store(v); }
  return v != 0;
}
struct StepScope {
  StepScope() { g_step_depth++; }
  ~StepScope() { if (--g_step_depth == 0 && ReleaseAtStepEnd()) vb_release_device_cache(); }
};

struct StepError : public std::runtime_error {
  explicit StepError(const std::string &m) : std::runtime_error(m) {}
};

// Devices the jobs of a step are spread over (vb_set_step_devices / VB_STEP_DEVICES; default: the calling thread's device named
// four times).  The reference starts its nj jobs as nj threads (VB/scr/steps/decode_gmm.cpp:300-324); here a job is a batch for a
// GPU, so the jobs are dealt round-robin to the entries of the list, one host thread per entry, each running its jobs in order: a
// device named several times is shared by that many job threads (one job's host work overlaps the others' kernels).
std::vector<int> g_step_devices;
std::mutex g_step_devices_mu;
std::vector<int> StepDevices() {
  {
    std::lock_guard<std::mutex> lk(g_step_devices_mu);
    if (!g_step_devices.empty()) return g_step_devices;
  }
  std::vector<int> d;
  if (const char *e = getenv("VB_STEP_DEVICES")) {
    std::string w; std::istringstream is(e);
    while (std::getline(is, w, ',')) if (!w.empty()) d.push_back(atoi(w.c_str()));
  }
  if (d.empty()) {        // default: the calling thread's device, its jobs on up to four host threads (VB_STEP_JOB_THREADS; 1 = one after the other)
    int n = 4;
    if (const char *e = getenv("VB_STEP_JOB_THREADS")) n = std::max(1, std::min(64, atoi(e)));
    d.assign((size_t)n, vb_get_device());
  }
  return d;
}
// The job threads live as long as the device list does: a job thread's caches (model, graph, feature table, search workspace:
// thread-local in voicebridge_api.cc) then persist from step to step exactly as the calling thread's do.  Threads started per
// step re-created a 20 GB workspace per thread and step (and never freed the old one).  The object is never destroyed: at
// process end the idle threads simply go away with the process.
class JobWorkers {
 public:
  explicit JobWorkers(const std::vector<int> &devs) : devs_(devs), w_(devs.size()) {
    for (size_t t = 0; t < w_.size(); t++) w_[t].th = std::thread([this, t]() { Loop((int)t); });
  }
  ~JobWorkers() {
    for (Worker &w : w_) { { std::lock_guard<std::mutex> lk(w.mu); w.quit = true; } w.cv.notify_all(); }
    for (Worker &w : w_) if (w.th.joinable()) w.th.join();
  }
  const std::vector<int> &devices() const { return devs_; }
  // fn(t) on worker t for t < nt, all at once; returns when every one has finished
  void Run(int nt, const std::function<void(int)> &fn) {
    for (int t = 0; t < nt; t++) { { std::lock_guard<std::mutex> lk(w_[t].mu); w_[t].task = &fn; w_[t].done = false; } w_[t].cv.notify_all(); }
    for (int t = 0; t < nt; t++) { std::unique_lock<std::mutex> lk(w_[t].mu); w_[t].cv.wait(lk, [&] { return w_[t].done; }); }
  }
 private:
  struct Worker { std::thread th; std::mutex mu; std::condition_variable cv; const std::function<void(int)> *task = nullptr; bool done = true, quit = false; };
  void Loop(int t) {
    Worker &w = w_[t];
    vb_set_device(devs_[t]);
    for (;;) {
      const std::function<void(int)> *task;
      {
        std::unique_lock<std::mutex> lk(w.mu);
        w.cv.wait(lk, [&] { return w.quit || w.task; });
        if (!w.task) break;
        task = w.task;
      }
      (*task)(t);          // (catches its own exceptions)
      { std::lock_guard<std::mutex> lk(w.mu); w.task = nullptr; w.done = true; }
      w.cv.notify_all();
    }
    vb_release_thread_context();    // the device list changed: this thread's device memory and its GPU context go with it
  }
  std::vector<int> devs_;
  std::vector<Worker> w_;
};
std::mutex g_workers_mu;            // one step at a time uses the workers
JobWorkers *g_workers = nullptr;
// what the last ForEachJob did: the device of every job (in job order) and the worker thread that ran it (-1: the calling thread).
// The fan-out is ONE code path whatever the list holds -- worker t calls vb_set_device(devs[t]) and runs jobs t+1, t+1+nt, ... --
// so a list that names one GPU twice exercises everything a two-GPU list does except the device index (vb_last_step_jobs, tests)
std::mutex g_last_jobs_mu;
std::vector<int> g_last_job_device, g_last_job_worker;
thread_local bool g_on_job_thread = false;

// fn(j) for j = 1..nj; an exception of any job is rethrown (the first one) after all threads have finished
template <class F>
void ForEachJob(int nj, F fn) {
  const std::vector<int> devs = StepDevices();
  if (devs.size() <= 1 || nj <= 1 || g_on_job_thread) {
    if (devs.size() == 1 && !g_on_job_thread) vb_set_device(devs[0]);
    if (!g_on_job_thread) { std::lock_guard<std::mutex> lk(g_last_jobs_mu); g_last_job_device.assign((size_t)nj, vb_get_device()); g_last_job_worker.assign((size_t)nj, -1); }
    for (int j = 1; j <= nj; j++) fn(j);
    return;
  }
  std::lock_guard<std::mutex> lk(g_workers_mu);
  if (g_workers && g_workers->devices() != devs) { delete g_workers; g_workers = nullptr; }
  if (!g_workers) g_workers = new JobWorkers(devs);
  const int nt = (int)std::min<size_t>(devs.size(), (size_t)nj);
  std::vector<std::string> errs(nt);
  { std::lock_guard<std::mutex> lk2(g_last_jobs_mu); g_last_job_device.assign((size_t)nj, -1); g_last_job_worker.assign((size_t)nj, -1); }
  const std::function<void(int)> body = [&](int t) {
    g_on_job_thread = true;
    try {
      StepScope scope;                     // (VB_RELEASE_DEVICE_CACHE=1: the thread's device cache is released when its last job is done)
      for (int j = 1 + t; j <= nj; j += nt) {
        { std::lock_guard<std::mutex> lk2(g_last_jobs_mu); g_last_job_device[(size_t)j - 1] = vb_get_device(); g_last_job_worker[(size_t)j - 1] = t; }
        fn(j);
      }
    } catch (const std::exception &e) { errs[t] = e.what(); if (errs[t].empty()) errs[t] = "error"; }
    catch (...) { errs[t] = "error"; }
  };
  g_workers->Run(nt, body);
  for (const auto &e : errs) if (!e.empty()) throw StepError(e);
}

// Utterances per search grid of a job's gmm-latgen-faster: 1024 shared out among the job threads that search side by side on one device
// (four threads: 256 each, one utterance per CU and the other jobs' grids fill the drains; one thread: 1024 in one grid -- four grids of 256
// in a row were 4 x 230 ms against 780, round 5), never fewer than 256.  VB_DECODE_BATCH overrides.
std::string DecodeBatchOption(int nj) {
  if (const char *e = getenv("VB_DECODE_BATCH")) if (atoi(e) > 0) return std::string("--decode-batch=") + std::to_string(atoi(e));
  const std::vector<int> devs = StepDevices();
  std::map<int, int> per_dev;
  const int nt = (int)std::min<size_t>(std::max<size_t>(devs.size(), 1), (size_t)std::max(nj, 1));
  for (int t = 0; t < nt; t++) per_dev[devs.empty() ? 0 : devs[(size_t)t]]++;
  int most = 1;
  for (const auto &kv : per_dev) most = std::max(most, kv.second);
  return "--decode-batch=" + std::to_string(std::max(256, 1024 / most));
}

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

// words.txt: "symbol id" per line
// words.txt: "word id" lines.  A lexicon of a quarter of a million words takes 40 ms to read into a map, and a step reads it once per job and
// once per scoring run: the table is parsed once per (file, size, mtime) and kept for the process (ids are dense: a vector).
struct SymTab {
  std::vector<std::string> w; std::vector<char> has;
  const std::string *Find(long id) const { return id >= 0 && (size_t)id < w.size() && has[(size_t)id] ? &w[(size_t)id] : nullptr; }
};
std::shared_ptr<const SymTab> ReadSymbolTable(const std::string &path) {
  static std::mutex mu;
  static std::map<std::string, std::shared_ptr<const SymTab>> cache;
  struct stat st;
  if (stat(path.c_str(), &st) != 0) throw StepError("cannot open " + path);
  const std::string key = path + "|" + std::to_string((long long)st.st_size) + "|" + std::to_string((long long)st.st_mtim.tv_sec) + "." + std::to_string((long long)st.st_mtim.tv_nsec);
  std::lock_guard<std::mutex> lk(mu);       // (a second thread asking for the same table waits for the first one's parse)
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  std::string buf;
  {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw StepError("cannot open " + path);
    buf.resize((size_t)st.st_size);
    f.read(&buf[0], (std::streamsize)buf.size());
    buf.resize((size_t)f.gcount());
  }
  auto t = std::make_shared<SymTab>();
  const char *p = buf.data(), *end = p + buf.size();
  while (p < end) {
    const char *eol = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));
    if (!eol) eol = end;
    const char *a = p;
    while (a < eol && isspace((unsigned char)*a)) a++;
    const char *b = a;
    while (b < eol && !isspace((unsigned char)*b)) b++;
    const char *c = b;
    while (c < eol && isspace((unsigned char)*c)) c++;
    if (b > a && c < eol && (isdigit((unsigned char)*c) || *c == '-' || *c == '+')) {      // (`is >> w >> id`: lines without both fields are skipped)
      char *e = nullptr;
      const long id = strtol(c, &e, 10);
      if (e != c && id >= 0 && id < (1L << 28)) {
        if ((size_t)id >= t->w.size()) { t->w.resize((size_t)id + 1); t->has.resize((size_t)id + 1, 0); }
        t->w[(size_t)id].assign(a, b); t->has[(size_t)id] = 1;
      }
    }
    p = eol + 1;
  }
  for (auto i = cache.begin(); i != cache.end();) i = (i->first.compare(0, path.size() + 1, path + "|") == 0) ? cache.erase(i) : std::next(i);      // (an older version of the file)
  cache[key] = t;
  return t;
}
// int2sym.pl -f 2-: every field after the key through the table (VB/scr/utils/int2sym.cpp)
std::string Int2SymLine(const SymTab &sym, const std::string &line) {
  std::istringstream is(line);
  std::string key, tok, out;
  is >> key;
  out = key;
  while (is >> tok) {
    const std::string *w = sym.Find(atoi(tok.c_str()));
    if (!w) throw StepError("Word-id " + tok + " not in symbol table.");
    out += " " + *w;
  }
  return out;
}
// FilterFile (VB/scr/utils/...): every (regex -> replacement) of the map applied to the line
std::string FilterLine(std::string line, const UMAPSS &filter) {
  for (const auto &kv : filter) line = std::regex_replace(line, std::regex(kv.first), kv.second);
  return line;
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
    // a segments file: the table holds recordings and the utterances are pieces of them (make_mfcc.cpp:93-190 pipes through
    // extract-segments; here compute-mfcc-feats takes the job's share of the segments file itself, --segments)
    const bool use_segments = Exists(Join(datadir, "segments"));
    const std::vector<std::string> seg_lines = use_segments ? ReadLines(Join(datadir, "segments")) : std::vector<std::string>();
    // VTLN warp factors per speaker or per utterance (make_mfcc.cpp:63-76)
    std::vector<std::string> vtln_opts;
    if (Exists(Join(datadir, "spk2warp"))) vtln_opts = {"--vtln-map=ark:" + Join(datadir, "spk2warp"), "--utt2spk=ark:" + Join(datadir, "utt2spk")};
    else if (Exists(Join(datadir, "utt2warp"))) vtln_opts = {"--vtln-map=ark:" + Join(datadir, "utt2warp")};
    const auto pieces = SplitLines(use_segments ? seg_lines : wav, nj);
    ForEachJob(nj, [&](int j) {
      const std::string J = std::to_string(j);
      const std::string piece = Join(logdir, "wav_" + name + "." + J + ".scp"), temp = Join(logdir, "wav_" + name + "." + J + ".mfctemp");
      WriteLines(piece, pieces[j - 1]);
      std::ofstream log(Join(logdir, "make_mfcc_" + name + "." + J + ".log"), std::ios::binary);
      if (pieces[j - 1].empty()) return;
      // the reference pipes compute-mfcc-feats through copy-feats --compress (make_mfcc.cpp:215, a temporary archive in its
      // port); here the program writes the final archive itself (its --compress / --write-num-frames extensions)
      (void)temp;
      std::vector<std::string> c = {"compute-mfcc-feats", "--config=" + mfcc_config, std::string("--compress=") + (compress ? "true" : "false")};
      if (write_utt2num_frames) c.push_back("--write-num-frames=ark,t:" + Join(logdir, "utt2num_frames." + J));
      for (const auto &o : vtln_opts) c.push_back(o);
      if (use_segments) { c.push_back("--segments=" + piece); c.push_back("scp,p:" + scp); }
      else c.push_back("scp,p:" + piece);
      c.push_back("ark,scp:" + Join(mfccdir, "raw_mfcc_" + name + "." + J + ".ark") + "," + Join(mfccdir, "raw_mfcc_" + name + "." + J + ".scp"));
      Argv a1(c);
      if (ComputeMFCCFeats(a1.argc(), a1.argv(), log) < 0) throw StepError("compute-mfcc-feats failed for job " + J + ", see " + logdir);
    });
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
    // Nothing to do when the split exists and none of its files is older than its source (split_data.cpp:64-82: judged by job 1's
    // directory).  Besides the work, leaving the files alone keeps their modification times: a job's feats.scp names the feature table
    // its thread holds on the device, and a rewritten file is a new table to upload (13-80 ms per job at the head of every step, round 5).
    {
      const std::string s1 = Join(root, "1");
      bool need = !IsDir(s1);
      if (!need) {
        auto mtime_ns = [](const std::string &f) { struct stat st; if (stat(f.c_str(), &st) != 0) return -1LL; return (long long)st.st_mtim.tv_sec * 1000000000LL + st.st_mtim.tv_nsec; };
        for (const char *f : {"utt2spk", "spk2utt", "spk2warp", "feats.scp", "text", "wav.scp", "cmvn.scp", "spk2gender", "vad.scp", "segments", "reco2file_and_channel", "utt2lang"}) {
          const std::string src = Join(data, f), dst = Join(s1, f);
          if (Exists(src) && (!Exists(dst) || mtime_ns(dst) < mtime_ns(src))) need = true;
        }
      }
      if (!need) return;
    }
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

// The reference's feature pipeline of a decoding job -- apply-cmvn | add-deltas (or splice-feats | transform-feats final.mat)
// (VB/scr/steps/decode_gmm.cpp:201-293) -- as the --fe-* options of the program that consumes the features (device_feats.h):
// the job's raw feats.scp is the only feature table any program of the job reads.
std::vector<std::string> FrontEndOptions(const std::string &srcdir, const std::string &dsn, bool lda, const std::string &cmvn_opts,
                                         const std::string &delta_opts, const std::string &splice_opts) {
  std::vector<std::string> g = {"--fe-cmvn=scp:" + Join(dsn, "cmvn.scp"), "--fe-utt2spk=ark:" + Join(dsn, "utt2spk")};
  for (const auto &o : SplitWords(cmvn_opts)) {
    if (o == "--norm-vars=true" || o == "--norm-vars") g.push_back("--fe-norm-vars=true");
    else if (o == "--norm-vars=false" || o == "--norm-means=true") continue;
    else throw StepError("cmvn_opts option " + o + " is not supported");
  }
  if (!lda) {
    g.push_back("--fe-type=delta");
    for (const auto &o : SplitWords(delta_opts)) {
      if (o.compare(0, 14, "--delta-order=") == 0) g.push_back("--fe-delta-order=" + o.substr(14));
      else if (o.compare(0, 15, "--delta-window=") == 0) g.push_back("--fe-delta-window=" + o.substr(15));
      else throw StepError("delta_opts option " + o + " is not supported");
    }
  } else {
    g.push_back("--fe-type=lda"); g.push_back("--fe-lda=" + Join(srcdir, "final.mat"));
    for (const auto &o : SplitWords(splice_opts)) {
      if (o.compare(0, 15, "--left-context=") == 0) g.push_back("--fe-left-context=" + o.substr(15));
      else if (o.compare(0, 16, "--right-context=") == 0) g.push_back("--fe-right-context=" + o.substr(16));
      else throw StepError("splice_opts option " + o + " is not supported");
    }
  }
  return g;
}

// ------------------------------------------------------------------------------------ Decode
namespace {
// What the nj jobs of one Decode share.  Decode = PrepareDecode + DecodeJob for every job + FinishDecode (+ scoring); DecodeFmllr
// runs the jobs of its speaker-independent first pass inside its own per-job chains (no barrier between the passes: while one
// job estimates its transforms on the host another is still searching).
struct DecodeJobs {
  std::string graph_dir, decode_dir, data_dir, srcdir, sdata, model, trans_dir, trans_all, splice_opts, cmvn_opts, delta_opts;
  bool use_trans = false, lda = false;
  int nj = 1, max_active = 7000;
  double beam = 13.0, lattice_beam = 6.0;
  float acwt = 0.083333f;
};

DecodeJobs PrepareDecode(const std::string &graph_dir, const std::string &data_dir, const std::string &decode_dir, const std::string &model_in,
                         const std::string &trans_dir, const std::string &iter, int nj, float acwt, int max_active, double beam, double lattice_beam,
                         bool decode_mbr) {
  if (decode_mbr) throw StepError("MBR decoding (decode_mbr) is not part of this build");
  if (nj < 1) throw StepError("nj must be >= 1");
  DecodeJobs c;
  c.graph_dir = graph_dir; c.decode_dir = decode_dir; c.data_dir = data_dir; c.trans_dir = trans_dir; c.nj = nj; c.acwt = acwt; c.max_active = max_active; c.beam = beam;
  c.lattice_beam = lattice_beam;
  // fMLLR transforms of an earlier pass (decode_gmm.cpp:127-172): one archive per job when the job
  // counts agree, otherwise all of them joined (binary archives concatenate)
  c.use_trans = !trans_dir.empty() && Exists(trans_dir);
  if (c.use_trans) {
    if (!Exists(Join(trans_dir, "trans.1"))) throw StepError("Expected " + Join(trans_dir, "trans.1") + " to exist.");
    const int nj_orig = atoi(FirstLine(Join(trans_dir, "num_jobs")).c_str());
    if (nj_orig < 1) throw StepError("Could not read number of jobs from file " + Join(trans_dir, "num_jobs") + ".");
    if (nj_orig != nj) {
      MakeDirs(decode_dir);
      c.trans_all = Join(decode_dir, "trans.ark");
      std::ofstream out(c.trans_all, std::ios::binary);
      if (!out) throw StepError("cannot write " + c.trans_all);
      for (int n = 1; n <= nj_orig; n++) {
        std::ifstream in(Join(trans_dir, "trans." + std::to_string(n)), std::ios::binary);
        if (!in) throw StepError(Join(trans_dir, "trans." + std::to_string(n)) + " is required but can not be found.");
        out << in.rdbuf();
      }
    }
  }
  c.srcdir = Parent(decode_dir);                  // the model directory (decode_gmm.cpp:80)
  c.model = model_in;
  if (c.model.empty()) c.model = Join(c.srcdir, (iter.empty() ? std::string("final") : iter) + ".mdl");
  for (const std::string &f : {Join(data_dir, "feats.scp"), Join(data_dir, "cmvn.scp"), c.model, Join(graph_dir, "HCLG.fst")})
    if (!Exists(f)) throw StepError(f + " is required but can not be found.");
  c.lda = Exists(Join(c.srcdir, "final.mat"));
  c.splice_opts = FirstLine(Join(c.srcdir, "splice_opts")); c.cmvn_opts = FirstLine(Join(c.srcdir, "cmvn_opts")); c.delta_opts = FirstLine(Join(c.srcdir, "delta_opts"));
  MakeDirs(decode_dir); MakeDir(Join(decode_dir, "log"));
  WriteLines(Join(decode_dir, "num_jobs"), {std::to_string(nj)});
  if (SplitData(data_dir, nj, false) < 0) throw StepError("SplitData failed: " + g_err);
  c.sdata = Join(data_dir, "split" + std::to_string(nj));
  return c;
}

// ---- the searches of ALL jobs of a step as one batch (opt-in: VB_STEP_COMBINED_SEARCH=1).  The reference runs its nj jobs side by side
// on nj cores (VB/scr/steps/decode_gmm.cpp:300-324); a job here is a batch for a GPU, and nj grids of one GPU queue behind each other
// and hold up the small kernels of the other jobs' host stages (fMLLR statistics, rescoring: 145 -> 400-540 ms and 176 -> 400-575 ms per
// call in DecodeFmllr, r03).  With the option the search stage of a step whose jobs share one device is ONE call over the concatenated
// feature tables (gmm-latgen-faster --job-of-utt: one persistent grid over all utterances) that still leaves lat.JOB / w.JOB per job,
// and the host stages between the searches stay per job on the job threads.  Measured (r03, 1024 utterances, nj = 4, one MI355X): the
// statistics calls do fall to 115-275 ms, but the step as a whole gets SLOWER -- Decode 1.42 -> 1.6-3.5 s, DecodeFmllr 3.6 -> 4.2 s: one
// thread now uploads all features (224 ms instead of 4 x 54 ms side by side), allocates one 30 GB score matrix (hipMalloc of that size takes
// 1-2 s) and determinizes / writes after the search instead of under the other jobs' searches.  Per-job searches on job threads stay the
// default; the files are the same either way (tests/test_gpu_steps.py::test_job_threads_leave_the_same_files).
bool CombinedSearch(int nj) {
  if (nj <= 1 || !getenv("VB_STEP_COMBINED_SEARCH")) return false;
  const std::vector<int> devs = StepDevices();
  for (int d : devs) if (d != devs[0]) return false;
  return true;
}
// work_dir/all_feats.scp (the jobs' feats.scp in job order) and work_dir/job_of_utt; returns the number of utterances
size_t WriteCombinedTables(const std::string &sdata, int nj, const std::string &work_dir) {
  std::vector<std::string> scp, jobs;
  for (int j = 1; j <= nj; j++)
    for (const auto &l : ReadLines(Join(Join(sdata, std::to_string(j)), "feats.scp"), false)) { scp.push_back(l); jobs.push_back(FirstField(l) + " " + std::to_string(j)); }
  WriteLines(Join(work_dir, "all_feats.scp"), scp);
  WriteLines(Join(work_dir, "job_of_utt"), jobs);
  return scp.size();
}
// decode.JOB.log of a step whose searches ran as one batch
void CombinedLogStubs(const std::string &log_dir, const std::string &stem, int nj, const std::string &all_log) {
  for (int j = 1; j <= nj; j++) {
    std::ofstream f(Join(log_dir, stem + "." + std::to_string(j) + ".log"), std::ios::binary | std::ios::app);
    f << "LOG (" << stem << ") the utterances of the " << nj << " jobs of this step were searched as one batch on one GPU: see " << all_log << "\n";
  }
}

void DecodeAllJobs(const DecodeJobs &c) {
  const std::string log_dir = Join(c.decode_dir, "log"), all_log = Join(log_dir, "decode.all.log");
  for (int j = 1; j <= c.nj; j++) unlink(Join(c.decode_dir, "lat." + std::to_string(j)).c_str());
  if (WriteCombinedTables(c.sdata, c.nj, c.decode_dir) == 0) return;
  std::ofstream log(all_log, std::ios::binary);
  std::vector<std::string> g = {"gmm-latgen-faster", "--max-active=" + std::to_string(c.max_active), "--beam=" + std::to_string(c.beam),
                                "--lattice-beam=" + std::to_string(c.lattice_beam), "--acoustic-scale=" + std::to_string(c.acwt),
                                "--allow-partial=true", "--decode-batch=1024", "--job-of-utt=" + Join(c.decode_dir, "job_of_utt")};
  for (const auto &o : FrontEndOptions(c.srcdir, c.data_dir, c.lda, c.cmvn_opts, c.delta_opts, c.splice_opts)) g.push_back(o);
  std::string trans_all = c.trans_all;
  if (c.use_trans) {
    if (trans_all.empty()) {          // one archive per job: joined (binary archives concatenate)
      trans_all = Join(c.decode_dir, "trans.all.ark");
      std::ofstream out(trans_all, std::ios::binary);
      for (int n = 1; n <= c.nj; n++) { std::ifstream in(Join(c.trans_dir, "trans." + std::to_string(n)), std::ios::binary); if (in) out << in.rdbuf(); }
    }
    g.push_back("--fe-trans=ark:" + trans_all);
  }
  const std::string symtab = Join(c.graph_dir, "words.txt");
  if (Exists(symtab)) g.push_back("--word-symbol-table=" + symtab);
  g.push_back(c.model); g.push_back(Join(c.graph_dir, "HCLG.fst")); g.push_back("scp:" + Join(c.decode_dir, "all_feats.scp"));
  g.push_back("ark:" + Join(c.decode_dir, "lat.JOB")); g.push_back("ark,t:" + Join(c.decode_dir, "w.JOB"));
  Argv a4(g);
  const int rc = GmmLatgenFaster(a4.argc(), a4.argv(), log);
  if (c.use_trans && c.trans_all.empty()) unlink(trans_all.c_str());
  unlink(Join(c.decode_dir, "all_feats.scp").c_str()); unlink(Join(c.decode_dir, "job_of_utt").c_str());
  if (rc < 0) throw StepError("gmm-latgen-faster failed, see " + all_log);
  CombinedLogStubs(log_dir, "decode", c.nj, all_log);
  if (Exists(symtab)) {
    const std::shared_ptr<const SymTab> symp = ReadSymbolTable(symtab); const SymTab &sym = *symp;
    for (int j = 1; j <= c.nj; j++) {
      const std::string J = std::to_string(j);
      if (!Exists(Join(c.decode_dir, "w." + J))) continue;
      std::vector<std::string> out;
      for (const auto &l : ReadLines(Join(c.decode_dir, "w." + J), false)) out.push_back(Int2SymLine(sym, l));
      WriteLines(Join(c.decode_dir, "transcription" + J + ".trn"), out);
    }
  }
}

void DecodeJob(const DecodeJobs &c, int j) {
  const std::string J = std::to_string(j), dsn = Join(c.sdata, J);
  unlink(Join(c.decode_dir, "lat." + J).c_str());
  std::ofstream log(Join(Join(c.decode_dir, "log"), "decode." + J + ".log"), std::ios::binary);
  if (ReadLines(Join(dsn, "feats.scp"), false).empty()) return;
  // The reference chains apply-cmvn | add-deltas (or splice-feats | transform-feats final.mat) [| transform-feats trans.JOB] |
  // gmm-latgen-faster through temporary archives (decode_gmm.cpp:201-293).  Here the chain runs on the device in front of the
  // scoring (GmmLatgenFaster's --fe-* options): the raw features of feats.scp are the only thing read.
  std::vector<std::string> g = {"gmm-latgen-faster", "--max-active=" + std::to_string(c.max_active), "--beam=" + std::to_string(c.beam),
                                "--lattice-beam=" + std::to_string(c.lattice_beam), "--acoustic-scale=" + std::to_string(c.acwt),
                                "--allow-partial=true", DecodeBatchOption(c.nj)};
  for (const auto &o : FrontEndOptions(c.srcdir, dsn, c.lda, c.cmvn_opts, c.delta_opts, c.splice_opts)) g.push_back(o);
  if (c.use_trans) g.push_back("--fe-trans=ark:" + (c.trans_all.empty() ? Join(c.trans_dir, "trans." + J) : c.trans_all));
  const std::string symtab = Join(c.graph_dir, "words.txt");
  if (Exists(symtab)) g.push_back("--word-symbol-table=" + symtab);
  g.push_back(c.model); g.push_back(Join(c.graph_dir, "HCLG.fst")); g.push_back("scp:" + Join(dsn, "feats.scp")); g.push_back("ark:" + Join(c.decode_dir, "lat." + J));
  g.push_back("ark,t:" + Join(c.decode_dir, "w." + J));
  Argv a4(g);
  const int rc = GmmLatgenFaster(a4.argc(), a4.argv(), log);
  if (rc < 0) throw StepError("gmm-latgen-faster failed for job " + J + ", see " + Join(c.decode_dir, "log"));
  // transcriptionJOB.trn: the words of w.JOB through the symbol table (decode_gmm.cpp:356-376)
  if (Exists(symtab)) {
    const std::shared_ptr<const SymTab> symp = ReadSymbolTable(symtab); const SymTab &sym = *symp;
    std::vector<std::string> out;
    for (const auto &l : ReadLines(Join(c.decode_dir, "w." + J), false)) out.push_back(Int2SymLine(sym, l));
    WriteLines(Join(c.decode_dir, "transcription" + J + ".trn"), out);
  }
}

void FinishDecode(const DecodeJobs &c) { if (!c.trans_all.empty()) unlink(c.trans_all.c_str()); }
}  // namespace

int Decode(const std::string &graph_dir, const std::string &data_dir, const std::string &decode_dir, const std::string &model_in,
           const std::string &trans_dir, UMAPSS &wer_ref_filter, UMAPSS &wer_hyp_filter, const std::string &iter, int nj, float acwt,
           int stage, int max_active, double beam, double lattice_beam, bool skip_scoring, bool decode_mbr, bool stats,
           const std::string &word_ins_penalty, int min_lmwt, int max_lmwt) {
  StepScope scope;          // (VB_RELEASE_DEVICE_CACHE=1: model, graph and search workspace go when the outermost step returns)
  (void)stage;
  const int rc_decode = Run([&] {
    const DecodeJobs c = PrepareDecode(graph_dir, data_dir, decode_dir, model_in, trans_dir, iter, nj, acwt, max_active, beam, lattice_beam, decode_mbr);
    if (CombinedSearch(nj)) { const std::vector<int> devs = StepDevices(); vb_set_device(devs[0]); DecodeAllJobs(c); }
    else ForEachJob(nj, [&](int j) { DecodeJob(c, j); });
    FinishDecode(c);
  });
  if (rc_decode != 0 || skip_scoring) return rc_decode;
  if (!Exists(Join(data_dir, "text"))) { g_err = "Not scoring because " + Join(data_dir, "text") + " is missing"; return -1; }
  return ScoreKaldiWER(data_dir, graph_dir, decode_dir, wer_ref_filter, wer_hyp_filter, nj, 0, decode_mbr, stats, lattice_beam, word_ins_penalty,
                       min_lmwt, max_lmwt, iter.empty() ? "final" : iter);
}

// ------------------------------------------------------------------------------------ DecodeFmllr
// VB/scr/steps/decode_fmllr.cpp:91-760.  Every stage is the reference's chain of bin functions over
// temporary archives in the job's split directory; the jobs run one after the other (they are
// batches for one GPU here, see voicebridge_steps.h).
int DecodeFmllr(const std::string &graphdir, const std::string &data, const std::string &dir, const std::string &alignment_model_in,
                const std::string &adapt_model_in, const std::string &final_model_in, int nj, float acwt, int stage, int max_active, double beam,
                double lattice_beam, double silence_weight, const std::string &si_dir_in, const std::string &fmllr_update_type, bool skip_scoring,
                bool decode_mbr, bool stats, const std::string &word_ins_penalty, int min_lmwt, int max_lmwt) {
  StepScope scope;
  std::string si_dir = si_dir_in;
  // the workspace size both passes share is a process-wide setting (the jobs run on threads of their own): put back on the way out
  struct WorkspaceGuard { int old = vb_get_workspace_max_active(); ~WorkspaceGuard() { vb_set_workspace_max_active(old); } } workspace_guard;
  const int rc = Run([&] {
    if (decode_mbr) throw StepError("MBR decoding (decode_mbr) is not part of this build");
    if (nj < 1) throw StepError("nj must be >= 1");
    const double first_beam = 10.0; const int first_max_active = 2000;      // decode_fmllr.cpp:116-117
    DecodeJobs si_jobs;
    bool run_si = false;
    const std::string srcdir = Parent(dir), sdata = Join(data, "split" + std::to_string(nj));
    MakeDirs(Join(dir, "log"));
    if (SplitData(data, nj, false) < 0) throw StepError("SplitData failed: " + g_err);
    WriteLines(Join(dir, "num_jobs"), {std::to_string(nj)});
    const std::string splice_opts = FirstLine(Join(srcdir, "splice_opts")), cmvn_opts = FirstLine(Join(srcdir, "cmvn_opts")),
                      delta_opts = FirstLine(Join(srcdir, "delta_opts"));
    const std::string silfile = Join(Join(graphdir, "phones"), "silence.csl");
    const std::string silphonelist = FirstLine(silfile);
    if (silphonelist.empty()) throw StepError(silfile + " is required but can not be found (or is empty).");
    for (const std::string &f : {Join(graphdir, "HCLG.fst"), Join(data, "feats.scp")}) if (!Exists(f)) throw StepError("Failed to find " + f);
    std::string alignment_model = alignment_model_in;
    if (alignment_model.empty() || !Exists(alignment_model)) {
      if (Exists(Join(srcdir, "final.alimdl"))) alignment_model = Join(srcdir, "final.alimdl");
      else if (Exists(Join(srcdir, "final.mdl"))) alignment_model = Join(srcdir, "final.mdl");
      else throw StepError("Failed to find alignment model.");
    }
    if (si_dir.empty() || !Exists(si_dir)) {
      si_dir = dir;
      while (si_dir.size() > 1 && si_dir.back() == '/') si_dir.pop_back();
      si_dir += ".si";
      if (stage <= 0) {
        // the speaker-independent first pass (decode_fmllr.cpp:215-259 calls Decode): its jobs run at the head of the per-job chains
        // below.  It searches with max-active 2000, the main pass with max_active: one set of search workspaces for both
        vb_set_workspace_max_active(std::max(first_max_active, max_active));
        if (!skip_scoring && !Exists(Join(data, "text")))
          throw StepError("First pass speaker-independent decoding failed: Not scoring because " + Join(data, "text") + " is missing");
        si_jobs = PrepareDecode(graphdir, data, si_dir, alignment_model, "", "", nj, acwt, first_max_active, first_beam, 6.0, decode_mbr);
        run_si = true;
      }
    }
    if (atoi(FirstLine(Join(si_dir, "num_jobs")).c_str()) != nj) throw StepError("Mismatch in number of jobs with si-dir.");
    const std::string adapt_model = (adapt_model_in.empty() || !Exists(adapt_model_in)) ? Join(srcdir, "final.mdl") : adapt_model_in;
    const std::string final_model = (final_model_in.empty() || !Exists(final_model_in)) ? Join(srcdir, "final.mdl") : final_model_in;
    for (const std::string &f : {Join(si_dir, "lat.1"), adapt_model, final_model}) if (!Exists(f) && !(run_si && f == Join(si_dir, "lat.1"))) throw StepError("Failed to find " + f);
    const bool lda = Exists(Join(srcdir, "final.mat"));
    const std::string acw = "--acoustic-scale=" + std::to_string(acwt), symtab = Join(graphdir, "words.txt");
    const bool laps = getenv("VBH_DEBUG_TIMES") != nullptr;
    const auto t_proc = std::chrono::steady_clock::now();      // (diagnostic timeline, VBH_DEBUG_TIMES=1: start / end of every program relative to the step's start)
    auto timeline = [&](const char *what, std::chrono::steady_clock::time_point t0) {
      if (!laps) return;
      const auto t1 = std::chrono::steady_clock::now();
      fprintf(stderr, "DecodeFmllr %-28s %8.1f ms   [%8.1f .. %8.1f] thread %zx\n", what, 1e3 * std::chrono::duration<double>(t1 - t0).count(),
              1e3 * std::chrono::duration<double>(t0 - t_proc).count(), 1e3 * std::chrono::duration<double>(t1 - t_proc).count(),
              std::hash<std::thread::id>()(std::this_thread::get_id()) & 0xFFFF);
    };
    auto run = [&](const char *what, int (*fn)(int, char **, std::ofstream &), std::vector<std::string> a, std::ofstream &log, bool none_ok = false) {
      Argv av(std::move(a));
      const auto t0 = std::chrono::steady_clock::now();
      const int r = fn(av.argc(), av.argv(), log);
      timeline(what, t0);
      if (r < 0 || (r != 0 && !none_ok)) throw StepError(std::string(what) + " failed (return code " + std::to_string(r) + "), see " + Join(dir, "log"));
    };
    // One job's part of the recipe in three pieces: `search` = the two gmm-latgen-faster passes, `before` = what lies between them
    // (first fMLLR estimate), `after` = everything behind the main search.  Jobs on distinct GPUs run the pieces back to back per job,
    // side by side (no barrier between the passes: while one job estimates its transforms on the host another is still searching); jobs
    // that share ONE GPU run the two searches as one batch over all jobs' utterances each (CombinedSearch) and the host pieces side
    // by side on the job threads -- no search grid of one job holds up the statistics / rescoring kernels of another.
    // The two big temporary tables of a job -- the main pass's raw lattices (lat.tmp.JOB, ~100 MB per 256 utterances, written once and read
    // by the determinization for the second estimate and by the rescoring) and the rescored lattices -- are handed from program to program
    // in memory (vbio "mem:" tables) when all of them run in this call, and dropped however the call ends.  A run that resumes behind the main
    // search (stage > 2) reads lat.tmp.JOB as files, which only a run with VB_STEP_FILE_TEMPS=1 (files always, as the reference leaves them)
    // has written: without them the resume fails with a message that says so (below).
    const bool mem_tmp = stage <= 2 && !getenv("VB_STEP_FILE_TEMPS");
    const std::string tmp_prefix = mem_tmp ? "mem:" : "", small_tmp = getenv("VB_STEP_FILE_TEMPS") ? "" : "mem:";
    auto drop_tmp = [&](const std::string &path) { if (vbio::IsMemPath(path)) vbio::MemUnlink(path); else unlink(path.c_str()); };
    // (whatever way this call ends, no in-memory table of it stays behind: an exception between a table's writer and its last reader would
    // otherwise leave a job's hundred megabytes in the registry)
    struct MemTempGuard {
      std::vector<std::string> names;
      ~MemTempGuard() { for (const std::string &n : names) vbio::MemUnlink(n); }
    } mem_guard;
    for (int j = 1; j <= nj; j++) {
      const std::string J = std::to_string(j);
      if (mem_tmp) mem_guard.names.push_back("mem:" + Join(dir, "lat.tmp." + J));
      if (!getenv("VB_STEP_FILE_TEMPS")) for (const char *t : {"grl.temp", "post.temp", "wsp.temp", "ldp.temp"}) mem_guard.names.push_back("mem:" + Join(Join(sdata, J), t));
    }
    const bool combined = CombinedSearch(nj);
    auto job_files = [&](int j, std::string *J, std::string *dsn) { *J = std::to_string(j); *dsn = Join(sdata, *J); return !ReadLines(Join(*dsn, "feats.scp"), false).empty(); };
    auto fe_of = [&](const std::string &dsn) { return FrontEndOptions(srcdir, dsn, lda, cmvn_opts, delta_opts, splice_opts); };
    auto with_fe = [&](const std::vector<std::string> &fe, std::vector<std::string> a, const std::string &trans_file) {
      std::vector<std::string> r{a[0]};
      for (const auto &o : fe) r.push_back(o);
      if (!trans_file.empty()) r.push_back("--fe-trans=ark:" + trans_file);
      r.insert(r.end(), a.begin() + 1, a.end());
      return r;
    };
    auto open_log = [&](const std::string &J, std::ofstream *log) { log->open(Join(Join(dir, "log"), "fmllr." + J + ".log"), std::ios::binary | std::ios::app); };
    std::vector<std::string> main_opts = {"gmm-latgen-faster", "--max-active=" + std::to_string(max_active), "--beam=" + std::to_string(beam),
                                          "--lattice-beam=" + std::to_string(lattice_beam), acw, "--determinize-lattice=false", "--allow-partial=true", DecodeBatchOption(nj)};
    if (Exists(symtab)) main_opts.push_back("--word-symbol-table=" + symtab);
    // ---- the features of every program below: the job's raw feats.scp behind the fused front end (the reference writes
    //      sifeats / transform_pass1feats / final feature archives: decode_fmllr.cpp:275-311, 390-402, 491-500, 568-580); the raw
    //      rows are uploaded once per job (the thread's table cache) and the transforms applied on the device
    auto before = [&](int j) {      // first-pass fMLLR from the SI lattices (:314-388)
      std::string J, dsn;
      if (!job_files(j, &J, &dsn) || stage > 1) return;
      std::ofstream log; open_log(J, &log);
      const std::string raw = "scp:" + Join(dsn, "feats.scp"), pre_trans = Join(dir, "pre_trans." + J), spk2utt = "--spk2utt=ark:" + Join(dsn, "spk2utt");
      // (the small temporaries between the programs -- posteriors, determinized lattices -- are in-memory tables too; the posteriors as
      // objects, kaldi_io.h FlatPostTable; VB_STEP_FILE_TEMPS=1: files, as the reference leaves them)
      const std::string t_post = small_tmp + Join(dsn, "post.temp"), t_wsp = small_tmp + Join(dsn, "wsp.temp");
      run("lattice-to-post", LatticeToPost, {"lattice-to-post", acw, "ark:" + Join(si_dir, "lat." + J), "ark:" + t_post}, log);
      run("weight-silence-post", WeightSilencePost, {"weight-silence-post", std::to_string(silence_weight), silphonelist, alignment_model, "ark:" + t_post, "ark:" + t_wsp}, log);
      // the reference pipes gmm-post-to-gpost (alignment model) into gmm-est-fmllr-gpost (adapt model); here the Gaussian
      // posteriors never leave the GPU: gmm-est-fmllr takes them from --ali-model and the statistics from the adapt model
      run("gmm-est-fmllr", GmmEstFmllr, with_fe(fe_of(dsn), {"gmm-est-fmllr", "--fmllr-update-type=" + fmllr_update_type, spk2utt, "--ali-model=" + alignment_model, adapt_model,
                                                           raw, "ark:" + t_wsp, "ark:" + pre_trans}, ""), log);
      drop_tmp(t_post); drop_tmp(t_wsp);
    };
    auto main_search = [&](int j) {      // main search on the first-pass features (:390-489), one job
      std::string J, dsn;
      if (!job_files(j, &J, &dsn) || stage > 2) return;
      std::ofstream log; open_log(J, &log);
      std::vector<std::string> g = main_opts;
      g.insert(g.end(), {adapt_model, Join(graphdir, "HCLG.fst"), "scp:" + Join(dsn, "feats.scp"), "ark:" + tmp_prefix + Join(dir, "lat.tmp." + J)});
      run("gmm-latgen-faster", GmmLatgenFaster, with_fe(fe_of(dsn), g, Join(dir, "pre_trans." + J)), log);
    };
    auto main_search_all = [&]() {       // ... all jobs as one batch
      if (stage > 2 || WriteCombinedTables(sdata, nj, dir) == 0) return;
      const std::string all_log = Join(Join(dir, "log"), "fmllr.all.log"), trans_all = Join(dir, "pre_trans.all");
      {
        std::ofstream out(trans_all, std::ios::binary);
        for (int n = 1; n <= nj; n++) { std::ifstream in(Join(dir, "pre_trans." + std::to_string(n)), std::ios::binary); if (in) out << in.rdbuf(); }
      }
      std::ofstream log(all_log, std::ios::binary | std::ios::app);
      std::vector<std::string> g = main_opts;
      g.insert(g.end(), {"--decode-batch=1024", "--job-of-utt=" + Join(dir, "job_of_utt"), adapt_model, Join(graphdir, "HCLG.fst"), "scp:" + Join(dir, "all_feats.scp"),
                         "ark:" + tmp_prefix + Join(dir, "lat.tmp.JOB")});
      run("gmm-latgen-faster", GmmLatgenFaster, with_fe(fe_of(data), g, trans_all), log);
      unlink(trans_all.c_str()); unlink(Join(dir, "all_feats.scp").c_str()); unlink(Join(dir, "job_of_utt").c_str());
      CombinedLogStubs(Join(dir, "log"), "fmllr", nj, all_log);
    };
    auto after = [&](int j) {
      std::string J, dsn;
      if (!job_files(j, &J, &dsn)) return;
      std::ofstream log; open_log(J, &log);
      const std::vector<std::string> fe = fe_of(dsn);
      const std::string raw = "scp:" + Join(dsn, "feats.scp"), pre_trans = Join(dir, "pre_trans." + J), trans = Join(dir, "trans." + J), lat_tmp = tmp_prefix + Join(dir, "lat.tmp." + J);
      const std::string spk2utt = "--spk2utt=ark:" + Join(dsn, "spk2utt");
      // A run that resumes behind the main search reads lat.tmp.JOB from disk -- which only a run with file temporaries left there (the
      // default keeps them in process memory and drops them however the call ends): say so instead of failing inside a program
      if (stage > 2 && stage <= 4 && !Exists(lat_tmp))
        throw StepError(lat_tmp + " is required to resume DecodeFmllr at stage " + std::to_string(stage) + " but can not be found: a run keeps the main pass's raw "
                        "lattices in memory unless VB_STEP_FILE_TEMPS=1 is set (then they are the reference's files and survive a failed run); "
                        "resume at stage <= 2, or run the earlier stages again with VB_STEP_FILE_TEMPS=1");
      if (stage <= 3) {      // second fMLLR estimate, composed with the first (:491-566)
        const std::string t_ldp = small_tmp + Join(dsn, "ldp.temp"), t_post = small_tmp + Join(dsn, "post.temp"), t_wsp = small_tmp + Join(dsn, "wsp.temp"), trans_tmp = Join(dir, "trans_tmp." + J);
        run("lattice-determinize-pruned", LatticeDeterminizePruned, {"lattice-determinize-pruned", acw, "--beam=" + std::to_string(4.0), "ark:" + lat_tmp, "ark:" + t_ldp}, log);
        run("lattice-to-post", LatticeToPost, {"lattice-to-post", acw, "ark:" + t_ldp, "ark:" + t_post}, log);
        run("weight-silence-post", WeightSilencePost, {"weight-silence-post", std::to_string(silence_weight), silphonelist, adapt_model, "ark:" + t_post, "ark:" + t_wsp}, log);
        run("gmm-est-fmllr", GmmEstFmllr, with_fe(fe, {"gmm-est-fmllr", "--fmllr-update-type=" + fmllr_update_type, spk2utt, adapt_model, raw, "ark:" + t_wsp, "ark:" + trans_tmp}, pre_trans), log);
        run("compose-transforms", ComposeTransforms, {"compose-transforms", "--b-is-affine=true", "ark:" + trans_tmp, "ark:" + pre_trans, "ark:" + trans}, log);
        drop_tmp(t_ldp); drop_tmp(t_post); drop_tmp(t_wsp);
      }
      if (stage <= 4) {      // rescoring with the final model on the final features, determinization (:568-640)
        const std::string t_grl = (getenv("VB_STEP_FILE_TEMPS") ? "" : "mem:") + Join(dsn, "grl.temp");
        run("gmm-rescore-lattice", GmmRescoreLattice, with_fe(fe, {"gmm-rescore-lattice", final_model, "ark:" + lat_tmp, raw, "ark:" + t_grl}, trans), log);
        run("lattice-determinize-pruned", LatticeDeterminizePruned, {"lattice-determinize-pruned", acw, "--beam=" + std::to_string(lattice_beam), "ark:" + t_grl, "ark:" + Join(dir, "lat." + J)}, log);
        drop_tmp(t_grl); drop_tmp(lat_tmp);
      }
    };
    for (int j = 1; j <= nj; j++) unlink(Join(Join(dir, "log"), "fmllr." + std::to_string(j) + ".log").c_str());
    auto si_job = [&](int j) {
      const auto t0 = std::chrono::steady_clock::now();
      struct Tl { decltype(timeline) &f; std::chrono::steady_clock::time_point t0; ~Tl() { f("first pass (Decode job)", t0); } } tl{timeline, t0};
      try { DecodeJob(si_jobs, j); }
      catch (const std::exception &e) { throw StepError(std::string("First pass speaker-independent decoding failed: ") + e.what()); }
    };
    // The first pass is scored as Decode scores it when it runs as a step of its own (decode_fmllr.cpp:215-259) -- 33 LM-weight x penalty
    // combinations over its lattices, a quarter of a second of host threads that nothing later depends on.  It starts when the last job's
    // first-pass lattices are written and runs beside the main searches (the host waits for the GPU then), not behind the whole recipe
    // (round 5; VB_STEP_SI_SCORING_LAST=1: as before).
    std::atomic<int> si_done{0};
    std::thread si_scorer;
    std::string si_score_err;
    struct SiJoin { std::thread &t; ~SiJoin() { if (t.joinable()) t.join(); } } si_join{si_scorer};
    const bool si_score_early = run_si && !skip_scoring && !getenv("VB_STEP_SI_SCORING_LAST");
    auto score_si = [&]() {
      UMAPSS f1, f2;
      if (ScoreKaldiWER(data, graphdir, si_dir, f1, f2, nj, 0, decode_mbr, stats, 6.0, word_ins_penalty, min_lmwt, max_lmwt, "final") < 0)
        si_score_err = "First pass speaker-independent decoding failed: " + g_err;
    };
    auto si_job_done = [&]() {
      if (++si_done == nj && si_score_early) {
        static std::mutex mu; std::lock_guard<std::mutex> lk(mu);
        if (!si_scorer.joinable()) si_scorer = std::thread(score_si);
      }
    };
    if (combined) {
      vb_set_device(StepDevices()[0]);
      if (run_si) {
        try { DecodeAllJobs(si_jobs); }
        catch (const std::exception &e) { throw StepError(std::string("First pass speaker-independent decoding failed: ") + e.what()); }
        for (int j = 0; j < nj; j++) si_job_done();
      }
      ForEachJob(nj, before);
      main_search_all();
      ForEachJob(nj, after);
    } else {
      ForEachJob(nj, [&](int j) {
        if (run_si) { si_job(j); si_job_done(); }
        before(j);
        main_search(j);
        after(j);
      });
    }
    if (run_si) {
      FinishDecode(si_jobs);
      if (si_scorer.joinable()) si_scorer.join();
      if (!si_score_err.empty()) throw StepError(si_score_err);
      if (!skip_scoring && !si_score_early) score_si();
      if (!si_score_err.empty()) throw StepError(si_score_err);
    }
  });
  if (rc != 0) return rc;
  if (!skip_scoring) {
    UMAPSS f1, f2;
    if (ScoreKaldiWER(data, graphdir, dir, f1, f2, nj, stage, decode_mbr, stats, beam, word_ins_penalty, min_lmwt, max_lmwt, "") < 0) return -1;
  }
  return Run([&] {       // words and transcripts of the final lattices (:642-745)
    const std::string symtab = Join(graphdir, "words.txt");
    ForEachJob(nj, [&](int j) {
      const std::string J = std::to_string(j);
      if (!Exists(Join(dir, "lat." + J))) return;
      std::ofstream log(Join(Join(dir, "log"), "decode." + J + ".log"), std::ios::binary);
      Argv a({"lattice-best-path", "ark:" + Join(dir, "lat." + J), "ark,t:" + Join(dir, "w." + J)});
      if (LatticeBestPath(a.argc(), a.argv(), log) < 0) throw StepError("lattice-best-path failed for job " + J);
      if (Exists(symtab)) {
        const std::shared_ptr<const SymTab> symp = ReadSymbolTable(symtab); const SymTab &sym = *symp;
        std::vector<std::string> out;
        for (const auto &l : ReadLines(Join(dir, "w." + J), false)) out.push_back(Int2SymLine(sym, l));
        WriteLines(Join(dir, "transcription" + J + ".trn"), out);
      }
      unlink(Join(dir, "trans_tmp." + J).c_str()); unlink(Join(dir, "pre_trans." + J).c_str());
    });
  });
}

// ------------------------------------------------------------------------------------ ScoreKaldiWER
// VB/scr/steps/scoring/score_kaldi_wer.cpp:36-254 (+ :256-420 the per-LMWT job, VB/scr/utils/best_wer.cpp).
// The lattices are read once; every (word insertion penalty, LM weight) pair scales, penalises and
// takes the best path in memory (lattice-scale --inv-acoustic-scale=LMWT | lattice-add-penalty |
// lattice-best-path), on as many host threads as there are pairs (at most nj x 4).
int ScoreKaldiWER(const std::string &data, const std::string &lang_or_graph, const std::string &dir, UMAPSS &wer_ref_filter,
                  UMAPSS &wer_hyp_filter, int nj, int stage, bool decode_mbr, bool stats, double beam, const std::string &word_ins_penalty,
                  int min_lmwt, int max_lmwt, const std::string &iter) {
  return Run([&] {
    (void)beam; (void)iter;
    if (decode_mbr) throw StepError("MBR decoding (decode_mbr) is not part of this build");
    const std::string symtab = Join(lang_or_graph, "words.txt");
    for (const std::string &f : {symtab, Join(dir, "lat.1"), Join(data, "text")})
      if (!Exists(f)) throw StepError(f + " is required but can not be found.");
    if (min_lmwt > max_lmwt) throw StepError("min_lmwt > max_lmwt");
    const std::string sk = Join(dir, "scoring_kaldi");
    MakeDirs(sk); MakeDir(Join(sk, "log"));
    std::ofstream slog(Join(Join(sk, "log"), "score.log"), std::ios::binary);
    // reference transcripts through the caller's filter
    std::vector<std::string> ref_lines;
    for (const auto &l : ReadLines(Join(data, "text"))) ref_lines.push_back(FilterLine(l, wer_ref_filter));
    WriteLines(Join(sk, "test_filt.txt"), ref_lines);
    std::vector<std::string> wips;
    { std::string w; std::istringstream is(word_ins_penalty); while (std::getline(is, w, ',')) { w.erase(0, w.find_first_not_of(" \t")); w.erase(w.find_last_not_of(" \t") + 1); if (!w.empty()) wips.push_back(w); } }
    if (wips.empty()) throw StepError("no word insertion penalty given");
    vbapi::Laps lap("ScoreKaldiWER");
    if (stage <= 0) {
      // all lat.* of the directory, in job order
      const int njobs = std::max(1, atoi(FirstLine(Join(dir, "num_jobs")).c_str()));
      std::vector<std::pair<std::string, vblat::CompactLattice>> lats;
      for (int j = 1; j <= njobs; j++) {
        const std::string f = Join(dir, "lat." + std::to_string(j));
        if (!Exists(f)) continue;
        vbio::LatticeReader reader("ark:" + f);
        std::string key; vblat::CompactLattice clat;
        while (reader.NextCompact(&key, &clat)) lats.emplace_back(key, std::move(clat));
      }
      if (lats.empty()) throw StepError("Could not find lat.* in " + dir);
      lap("lattices read");
      const std::shared_ptr<const SymTab> symp = ReadSymbolTable(symtab); const SymTab &sym = *symp;
      lap("symbol table read");
      // 33 combinations x a thousand lattices: the work items are (combination, block of 128 utterances), so that sixteen threads stay busy
      // to the end (whole combinations as items: 33 on 16 threads = three rounds, the last one of a single item); the thread that finishes
      // a combination's last block writes its log and hypotheses and runs compute-wer
      struct Job {
        std::string wip; int lmwt = 0; std::string err; std::mutex mu;
        std::vector<std::string> hyp, logl; std::vector<char> has; int blocks_left = 0;
      };
      const size_t NL = lats.size(), BLK = 128, nblk = (NL + BLK - 1) / BLK;
      std::vector<std::unique_ptr<Job>> jobs;
      for (const auto &w : wips) {
        MakeDirs(Join(Join(sk, "penalty_" + w), "log"));
        for (int l = min_lmwt; l <= max_lmwt; l++) {
          jobs.emplace_back(new Job());
          Job &jb = *jobs.back();
          jb.wip = w; jb.lmwt = l; jb.hyp.resize(NL); jb.logl.resize(NL); jb.has.assign(NL, 0); jb.blocks_left = (int)nblk;
        }
      }
      std::atomic<size_t> next{0};
      auto work = [&]() {
        char num[96];
        for (size_t i = next++; i < jobs.size() * nblk; i = next++) {
          Job &jb = *jobs[i / nblk];
          const size_t u0 = (i % nblk) * BLK, u1 = std::min(NL, u0 + BLK);
          try {
            const float inv_acoustic_scale = (float)jb.lmwt, acoustic_scale = (float)(1.0 / inv_acoustic_scale);     // lattice-scale.cc:70-80
            const float pen = (float)atof(jb.wip.c_str());
            for (size_t u = u0; u < u1; u++) {
              const auto &kv = lats[u];
              vblat::CompactLattice c = kv.second;
              vblat::ScaleLattice(1.0, 0.0, 0.0, (double)acoustic_scale, &c);
              vblat::AddWordInsPen(pen, &c);
              vblat::CompactLattice best;
              vblat::CompactLatticeShortestPath(c, &best);
              if (best.start < 0) { jb.logl[u] = "WARNING (lattice-best-path) Best-path failed for key " + kv.first + "\n"; continue; }
              std::vector<int32_t> words; vblat::Weight w;
              if (!vblat::GetLinearSymbolSequence(best, nullptr, &words, &w)) throw StepError("best path of " + kv.first + " is not linear");
              { std::ostringstream o; o << "LOG (lattice-best-path) For utterance " << kv.first << ", best cost " << w.v1 << " + " << w.v2 << " = " << (w.v1 + w.v2) << "\n"; jb.logl[u] = o.str(); }
              std::string line = kv.first;
              for (int32_t x : words) { snprintf(num, sizeof(num), " %d", x); line += num; }
              jb.hyp[u] = FilterLine(Int2SymLine(sym, line), wer_hyp_filter);
              jb.has[u] = 1;
            }
          } catch (const std::exception &e) { std::lock_guard<std::mutex> lk(jb.mu); jb.err = e.what(); }
          bool last;
          { std::lock_guard<std::mutex> lk(jb.mu); last = --jb.blocks_left == 0; }
          if (!last || !jb.err.empty()) continue;
          try {
            std::ofstream jlog(Join(Join(Join(sk, "penalty_" + jb.wip), "log"), "LMWT." + std::to_string(jb.lmwt) + ".log"), std::ios::binary);
            std::vector<std::string> hyp;
            for (size_t u = 0; u < NL; u++) { jlog << jb.logl[u]; if (jb.has[u]) hyp.push_back(std::move(jb.hyp[u])); }
            const std::string hyp_file = Join(Join(sk, "penalty_" + jb.wip), std::to_string(jb.lmwt) + ".txt");
            WriteLines(hyp_file, hyp);
            Argv a({"compute-wer", "--mode=present", "ark:" + Join(sk, "test_filt.txt"), "ark,p:" + hyp_file,
                    Join(dir, "wer_" + std::to_string(jb.lmwt) + "_" + jb.wip)});
            if (ComputeWer(a.argc(), a.argv(), jlog) < 0) throw StepError("compute-wer failed, see " + Join(Join(sk, "penalty_" + jb.wip), "log"));
            std::vector<std::string>().swap(jb.hyp); std::vector<std::string>().swap(jb.logl);
          } catch (const std::exception &e) { std::lock_guard<std::mutex> lk(jb.mu); jb.err = e.what(); }
        }
      };
      const unsigned nt = std::max(1u, std::min((unsigned)(jobs.size() * nblk), std::min((unsigned)std::max(nj, 1) * 4u, std::max(1u, std::thread::hardware_concurrency()))));
      std::vector<std::thread> pool;
      for (unsigned t = 1; t < nt; t++) pool.emplace_back(work);
      work();
      for (std::thread &t : pool) t.join();
      for (const auto &jb : jobs) if (!jb->err.empty()) throw StepError("scoring LMWT " + std::to_string(jb->lmwt) + " penalty " + jb->wip + ": " + jb->err);
      lap("best paths + compute-wer of every combination");
    }
    if (stage <= 1) {
      // BestWer (VB/scr/utils/best_wer.cpp:25-71): smallest %SER, then smallest %WER, in the order wip-major / LMWT-minor
      double best_ser = 0, best_wer = 0; std::string best_wip; int best_lmwt = -1; std::string best_line;
      for (const auto &w : wips)
        for (int l = min_lmwt; l <= max_lmwt; l++) {
          const std::string f = Join(dir, "wer_" + std::to_string(l) + "_" + w);
          const std::vector<std::string> fields = SplitWords(FirstLine(f));
          if (fields.size() < 9) throw StepError("Could not read WER results from " + f);
          const double pwer = atof(fields[0].c_str()), pser = atof(fields[6].c_str());
          if (best_lmwt < 0 || pser < best_ser || (pser == best_ser && pwer < best_wer)) { best_ser = pser; best_wer = pwer; best_wip = w; best_lmwt = l; best_line = FirstLine(f); }
        }
      if (best_lmwt < 0) throw StepError("Could not determine best WER.");
      slog << "Found the best WER/SER combination: WER=" << best_wer << "%, SER=" << best_ser << "% (LMWT " << best_lmwt << ", word insertion penalty " << best_wip << ")\n";
      WriteLines(Join(sk, "best_wer"), {best_line + " " + Join(dir, "wer_" + std::to_string(best_lmwt) + "_" + best_wip)});
      if (stats) {
        MakeDir(Join(sk, "wer_details"));
        WriteLines(Join(Join(sk, "wer_details"), "lmwt"), {std::to_string(best_lmwt)});
        WriteLines(Join(Join(sk, "wer_details"), "wip"), {best_wip});
        // the hypotheses of the best combination aligned with the transcripts, then counted per utterance, per speaker and per
        // word pair, and the bootstrap interval of the WER (score_kaldi_wer.cpp:180-236; the special symbol is passed with its
        // quotes there, no shell strips them, so the files carry '***')
        const std::string details = Join(sk, "wer_details"), pen = Join(sk, "penalty_" + best_wip), at = Join(pen, std::to_string(best_lmwt) + ".at");
        const std::string eps = "'***'";
        // (the bootstrap interval needs the hypotheses only: it runs beside the alignment and the three tables made from it)
        std::string boot_err;
        std::thread boot([&]() {
          try {
            std::ofstream blog(Join(Join(sk, "log"), "wer_bootci.log"), std::ios::binary);
            Argv a({"compute-wer-bootci", "--mode=present", "ark:" + Join(sk, "test_filt.txt"), "ark:" + Join(pen, std::to_string(best_lmwt) + ".txt"), Join(details, "wer_bootci")});
            if (ComputeWerBootci(a.argc(), a.argv(), blog) < 0) boot_err = "compute-wer-bootci failed, see " + Join(Join(sk, "log"), "wer_bootci.log");
          } catch (const std::exception &e) { boot_err = e.what(); if (boot_err.empty()) boot_err = "compute-wer-bootci failed"; }
        });
        struct JoinBoot { std::thread &t; ~JoinBoot() { if (t.joinable()) t.join(); } } join_boot{boot};
        {
          Argv a({"align-text", "--special-symbol=" + eps, "ark:" + Join(sk, "test_filt.txt"), "ark:" + Join(pen, std::to_string(best_lmwt) + ".txt"), "ark,t:" + at});
          std::ofstream alog(Join(Join(sk, "log"), "align_text.log"), std::ios::binary);
          if (AlignTextLogged(a.argc(), a.argv(), alog) < 0) throw StepError("align-text failed, see " + Join(Join(sk, "log"), "align_text.log"));
        }
        if (WerPerUttDetails(at, Join(details, "per_utt"), eps) < 0) throw StepError("wer_per_utt_details failed");
        if (WerPerSpkDetails(Join(details, "per_utt"), Join(data, "utt2spk"), Join(details, "per_spk")) < 0) throw StepError("wer_per_spk_details failed");
        if (WerOpsDetails(Join(details, "per_utt"), Join(details, "ops"), eps) < 0) throw StepError("wer_ops_details failed");
        boot.join();
        if (!boot_err.empty()) throw StepError(boot_err);
        unlink(at.c_str());
      }
      lap("best combination + details");
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
int vb_DecodeEx(const char *graph_dir, const char *data_dir, const char *decode_dir, const char *model, const char *trans_dir, int nj,
                float acwt, int max_active, double beam, double lattice_beam, int skip_scoring, const char *word_ins_penalty,
                int min_lmwt, int max_lmwt) {
  UMAPSS a, b;
  return Decode(graph_dir, data_dir, decode_dir, model ? model : "", trans_dir ? trans_dir : "", a, b, "", nj, acwt, 0, max_active, beam, lattice_beam,
                skip_scoring != 0, false, true, word_ins_penalty ? word_ins_penalty : "0.0,0.5,1.0", min_lmwt, max_lmwt);
}
int vb_DecodeFmllr(const char *graphdir, const char *data, const char *dir, int nj, float acwt, int max_active, double beam, double lattice_beam,
                   double silence_weight, const char *si_dir, int skip_scoring, const char *word_ins_penalty, int min_lmwt, int max_lmwt) {
  return DecodeFmllr(graphdir, data, dir, "", "", "", nj, acwt, 0, max_active, beam, lattice_beam, silence_weight, si_dir ? si_dir : "", "full",
                     skip_scoring != 0, false, true, word_ins_penalty ? word_ins_penalty : "0.0,0.5,1.0", min_lmwt, max_lmwt);
}
int vb_ScoreKaldiWER(const char *data, const char *lang_or_graph, const char *dir, int nj, const char *word_ins_penalty, int min_lmwt,
                     int max_lmwt) {
  UMAPSS a, b;
  return ScoreKaldiWER(data, lang_or_graph, dir, a, b, nj, 0, false, true, 6.0, word_ins_penalty ? word_ins_penalty : "0.0,0.5,1.0", min_lmwt, max_lmwt);
}
int vb_set_release_device_cache(int on) { g_release_at_step_end.store(on ? 1 : 0); return 0; }
const char *vb_steps_last_error(void) { return g_err.c_str(); }
// jobs of the last step's last fan-out: out_device[j], out_worker[j] for j < min(n, cap); returns n
int vb_last_step_jobs(int *out_device, int *out_worker, int cap) {
  std::lock_guard<std::mutex> lk(g_last_jobs_mu);
  const int n = (int)g_last_job_device.size();
  for (int j = 0; j < n && j < cap; j++) { out_device[j] = g_last_job_device[(size_t)j]; out_worker[j] = g_last_job_worker[(size_t)j]; }
  return n;
}
int vb_set_step_devices(const int *devices, int n) {
  std::lock_guard<std::mutex> lk(g_step_devices_mu);
  g_step_devices.clear();
  for (int i = 0; i < n; i++) { if (devices[i] < 0) return -1; g_step_devices.push_back(devices[i]); }
  return 0;
}
}
