// api_common.h -- what every argv-style bin function of this build shares: the Kaldi command-line
// grammar, the log sink and the "never throw across the boundary" guard (return -1 instead).
#ifndef VOICEBRIDGE_API_COMMON_H_
#define VOICEBRIDGE_API_COMMON_H_

#include <chrono>
#include <sys/resource.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <initializer_list>
#include <iostream>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace vbapi {

struct ApiError : public std::runtime_error {
  explicit ApiError(const std::string &m) : std::runtime_error(m) {}
};

struct Log {
  std::ofstream *f;
  explicit Log(std::ofstream *file) : f(file) {}
  std::ostream &os() { return (f && f->is_open()) ? static_cast<std::ostream &>(*f) : std::cerr; }
};

// ---- Kaldi command-line grammar (K/util/parse-options.cc:341-352,477-508): --name=value,
// --flag (boolean true), --config=file with one option per line, "--" ends the options
// diagnostic (VBH_DEBUG_TIMES=1): where the host time of a program goes, lap by lap on stderr
struct Laps {
  const char *prog; bool on; std::chrono::steady_clock::time_point t; struct rusage r;
  explicit Laps(const char *p) : prog(p), on(getenv("VBH_DEBUG_TIMES") != nullptr), t(std::chrono::steady_clock::now()) { if (on) getrusage(RUSAGE_SELF, &r); }
  void operator()(const char *what) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    struct rusage q; getrusage(RUSAGE_SELF, &q);      // (of the whole process: every thread that ran during the lap)
    auto ms = [](const timeval &a, const timeval &b) { return (a.tv_sec - b.tv_sec) * 1e3 + (a.tv_usec - b.tv_usec) / 1e3; };
    fprintf(stderr, "%s lap %-34s %8.1f ms   (process: user %.0f ms, system %.0f ms, page faults %ld, context switches %ld + %ld)\n", prog, what,
            1e3 * std::chrono::duration<double>(now - t).count(), ms(q.ru_utime, r.ru_utime), ms(q.ru_stime, r.ru_stime), q.ru_minflt - r.ru_minflt,
            q.ru_nvcsw - r.ru_nvcsw, q.ru_nivcsw - r.ru_nivcsw);
    t = now; r = q;
  }
};

struct Options {
  std::map<std::string, std::string> kv;
  std::vector<std::string> args;
  std::map<std::string, bool> known;
  void Know(std::initializer_list<const char *> names) { for (const char *n : names) known[n] = true; }
  static std::string Norm(std::string s) { for (char &c : s) if (c == '_') c = '-'; return s; }
  void Set(const std::string &opt) {
    const size_t eq = opt.find('=');
    const std::string name = Norm(opt.substr(2, eq == std::string::npos ? std::string::npos : eq - 2));
    const std::string val = eq == std::string::npos ? "true" : opt.substr(eq + 1);
    if (name == "config") { ReadConfig(val); return; }
    if (name == "verbose" || name == "print-args" || name == "help") return;
    if (!known.count(name)) throw ApiError("invalid option --" + name);
    kv[name] = val;
  }
  void ReadConfig(const std::string &path) {
    std::ifstream f(path);
    if (!f) throw ApiError("cannot open config file " + path);
    std::string line;
    while (std::getline(f, line)) {
      const size_t h = line.find('#');
      if (h != std::string::npos) line = line.substr(0, h);
      const size_t a = line.find_first_not_of(" \t\r");
      if (a == std::string::npos) continue;
      const size_t b = line.find_last_not_of(" \t\r");
      line = line.substr(a, b - a + 1);
      if (line.compare(0, 2, "--") != 0) throw ApiError("bad line in config file " + path + ": " + line);
      Set(line);
    }
  }
  void Parse(int argc, char *argv[]) {
    bool opts_done = false;
    for (int i = 1; i < argc; i++) {
      const std::string a = argv[i];
      if (!opts_done && a == "--") { opts_done = true; continue; }
      if (!opts_done && a.compare(0, 2, "--") == 0) Set(a);
      else args.push_back(a);
    }
  }
  bool Has(const char *n) const { return kv.count(n) != 0; }
  std::string Str(const char *n, const std::string &d) const { auto i = kv.find(n); return i == kv.end() ? d : i->second; }
  double Num(const char *n, double d) const { auto i = kv.find(n); return i == kv.end() ? d : atof(i->second.c_str()); }
  bool Bool(const char *n, bool d) const {
    auto i = kv.find(n);
    if (i == kv.end()) return d;
    const std::string &v = i->second;
    if (v == "true" || v == "t" || v == "1" || v == "") return true;
    if (v == "false" || v == "f" || v == "0") return false;
    throw ApiError(std::string("bad boolean for --") + n + ": " + v);
  }
};

template <class F>
int Guard(const char *prog, std::ofstream *file_log, F body) {
  Log log(file_log);
  try {
    return body(log);
  } catch (const std::exception &e) {
    log.os() << "ERROR (" << prog << ") " << e.what() << "\n";
    return -1;
  } catch (...) {
    log.os() << "ERROR (" << prog << ") unknown exception\n";
    return -1;
  }
}

template <class F>
int WithLog(const char *log_path, F f) {
  std::ofstream log;
  if (log_path && *log_path) log.open(log_path, std::ios::app);
  return f(log);
}

}  // namespace vbapi

// the raw lattice of a search result (defined in voicebridge_api.cc)
struct vbh_lattice;
namespace vblat { struct Lattice; }
struct vbh_ctx;
namespace vbapi { vbh_ctx *SharedCtx(); }      // the calling thread's GPU context (vb_set_device), created on first use
namespace vbapi { void RawLatticeFromResult(const vbh_lattice &L, int32_t graph_start, vblat::Lattice *lat); }
#endif
