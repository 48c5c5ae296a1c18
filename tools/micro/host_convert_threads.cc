#include "lattice.h"
#include "kaldi_io.h"
#include <chrono>
#include <cstdio>
#include <random>
#include <thread>
#include <atomic>
#include <hip/hip_runtime_api.h>
#include <sys/resource.h>
#include <malloc.h>
using namespace vblat;
static Lattice make(int T, unsigned seed) {
  std::mt19937 rng(seed);
  Lattice lat; lat.start = 0;
  std::vector<std::vector<int>> layer(T + 1);
  lat.st.emplace_back(); layer[0].push_back(0);
  for (int t = 1; t <= T; t++) { int k = 2 + rng() % 3; for (int i = 0; i < k; i++) { layer[t].push_back((int)lat.st.size()); lat.st.emplace_back(); } }
  for (int t = 0; t < T; t++) for (int s : layer[t]) { int na = 1 + rng() % 2; for (int a = 0; a < na; a++) { Arc arc; arc.ilabel = 1 + rng() % 7000; arc.olabel = (rng() % 50 == 0) ? 1 + rng() % 1000 : 0; arc.w.v1 = (rng() % 100) * 0.01f; arc.w.v2 = (rng() % 1000) * 0.01f; arc.next = layer[t + 1][rng() % layer[t + 1].size()]; lat.st[s].arcs.push_back(arc); } }
  for (int s : layer[T]) lat.st[s].final = One();
  Connect(&lat);
  return lat;
}
int main(int argc, char **argv) {
  const int NT = argc > 1 ? atoi(argv[1]) : 1, N = 64;
  std::vector<Lattice> lats;
  if (argc > 2) {      // real raw lattices: files holding one serialised lattice object each (no key)
    for (int i = 0; i < N; i++) {
      FILE *f = fopen(argv[2 + i % (argc - 2)], "rb"); if (!f) { perror(argv[2]); return 1; }
      std::string b; char buf[65536]; size_t k; while ((k = fread(buf, 1, sizeof buf, f)) > 0) b.append(buf, k); fclose(f);
      Lattice l; vbio::ParseLatticeBytes(b.data(), b.size(), &l); lats.push_back(std::move(l));
    }
  } else for (int i = 0; i < N; i++) lats.push_back(make(3000, i));
  size_t arcs = 0; for (auto &s : lats[0].st) arcs += s.arcs.size();
  printf("states %zu arcs %zu\n", lats[0].st.size(), arcs);
  std::atomic<long long> ns_conv{0}, ns_ser{0}, ns_free{0}, ns_sort{0}, ns_back{0};
  if (getenv("HIPINIT")) { void *d = nullptr, *h = nullptr; hipSetDevice(0); hipMalloc(&d, 64 << 20); hipHostMalloc(&h, 64 << 20, 0); printf("hip initialised %p %p\n", d, h); }
  const bool keep = getenv("KEEP") != nullptr; std::vector<CompactLattice> kept(N * 8);
  std::atomic<int> next{0}; std::atomic<int> lim{N * 4};
  auto work = [&]() {
    for (int i = next++; i < lim.load(); i = next++) {
      auto t0 = std::chrono::steady_clock::now();
      auto tick = [&](std::atomic<long long> &a) { auto n = std::chrono::steady_clock::now(); a += std::chrono::duration_cast<std::chrono::nanoseconds>(n - t0).count(); t0 = n; };
      CompactLattice cl; CompactLattice &c = keep ? kept[i] : cl; ConvertLattice(lats[i % N], &c); tick(ns_conv);
      TopSort(&c); tick(ns_sort);
      std::string e; vbio::SerializeLattice(c, &e); tick(ns_ser);
      Lattice back; ConvertLattice(c, &back, false); tick(ns_back);
      if (!keep) CompactLattice().st.swap(c.st); tick(ns_free);
    }
  };
  if (getenv("MOPT")) { mallopt(M_MMAP_THRESHOLD, 1<<30); mallopt(M_TRIM_THRESHOLD, 1<<30); mallopt(M_TOP_PAD, 64<<20); }
  struct rusage r0; getrusage(RUSAGE_SELF, &r0);
  auto T0 = std::chrono::steady_clock::now();
  std::atomic<int> arrived{0};
  auto work2 = [&]() { work(); arrived++; while (arrived.load() < NT) {} 
     if (false) {} };
  std::vector<std::thread> pool; for (int t = 1; t < NT; t++) pool.emplace_back([&]() { work(); arrived++; while (lim.load() == N * 4) std::this_thread::yield(); work(); });
  work(); arrived++; while (arrived.load() < NT) std::this_thread::yield();
  double wall1 = std::chrono::duration<double>(std::chrono::steady_clock::now() - T0).count();
  printf("pass 1 wall %.1f ms convert thread-ms/lattice %.3f\n", wall1 * 1e3, ns_conv / 1e6 / (N * 4)); ns_conv = 0; ns_sort = 0; ns_ser = 0; ns_back = 0; ns_free = 0;
  getrusage(RUSAGE_SELF, &r0); T0 = std::chrono::steady_clock::now();
  lim = N * 8; work(); for (auto &t : pool) t.join();
  double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - T0).count();
  struct rusage r1; getrusage(RUSAGE_SELF, &r1); printf("minflt %ld nvcsw %ld nivcsw %ld utime %.0f ms stime %.0f ms\n", r1.ru_minflt-r0.ru_minflt, r1.ru_nvcsw-r0.ru_nvcsw, r1.ru_nivcsw-r0.ru_nivcsw, (r1.ru_utime.tv_sec-r0.ru_utime.tv_sec)*1e3+(r1.ru_utime.tv_usec-r0.ru_utime.tv_usec)/1e3, (r1.ru_stime.tv_sec-r0.ru_stime.tv_sec)*1e3+(r1.ru_stime.tv_usec-r0.ru_stime.tv_usec)/1e3);
  printf("threads %d wall %.1f ms; per lattice (thread-ms): convert %.3f topsort %.3f serialize %.3f expand %.3f free %.3f\n", NT, wall * 1e3, ns_conv / 1e6 / (N * 4), ns_sort / 1e6 / (N * 4), ns_ser / 1e6 / (N * 4), ns_back / 1e6 / (N * 4), ns_free / 1e6 / (N * 4));
}
