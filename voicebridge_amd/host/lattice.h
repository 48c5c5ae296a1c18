// lattice.h -- lattices after the search: connect / sort / determinize / scale / best path.
//
// Host C++ written for this repository; nothing of Kaldi/OpenFst is linked or copied.  What each
// piece follows (paths relative to /root/reference):
//   LatticeWeight (graph, acoustic), Compare/Plus/Times/Divide/ApproxEqual/ConvertToCost
//                                             kaldi-master/src/fstext/lattice-weight.h:225-370,786-805
//   CompactLatticeWeight (weight + transition-id string)   lattice-weight.h:400-660
//   Connect, TopSort (depth-first finishing order), Invert, ArcSort(ilabel)
//                                             openfst-win-1.6/src/include/fst/{connect,dfs-visit,topsort,invert,arcsort}.h
//   DeterminizeLatticePruned, the phone-level first pass, DeterminizeLatticePhonePrunedWrapper
//                                             kaldi-master/src/lat/determinize-lattice-pruned.cc
//   LatticeStringRepository                    kaldi-master/src/fstext/determinize-lattice-inl.h:37-260
//   PruneLattice                               kaldi-master/src/lat/lattice-functions.cc (PruneLattice)
//   ScaleLattice, AddWordInsPenToCompactLattice, CompactLatticeShortestPath, ConvertLattice
//                                             kaldi-master/src/lat/lattice-functions.cc, fstext/lattice-utils-inl.h
// This is the step right after the search in the reference's Decode:
// gmm-latgen-faster determinizes every utterance's raw lattice before writing it
// (kaldi-master/src/decoder/decoder-wrappers.cc:262-284), and scoring rescales it, adds the word
// insertion penalty and takes the best path (VB/scr/steps/scoring/score_kaldi_wer.cpp:36-254).
#ifndef VOICEBRIDGE_LATTICE_H_
#define VOICEBRIDGE_LATTICE_H_

#include <cstdint>
#include <functional>
#include <limits>
#include <string>
#include <utility>
#include <cstring>
#include <utility>
#include <vector>

namespace vblat {

constexpr float kInf = std::numeric_limits<float>::infinity();
constexpr float kDelta = 1.0f / 1024.0f;     // fst::kDelta

struct Weight {            // LatticeWeight: value1 = graph cost, value2 = acoustic cost
  float v1, v2;
};
inline Weight Zero() { return Weight{kInf, kInf}; }
inline Weight One() { return Weight{0.0f, 0.0f}; }
inline bool IsZero(const Weight &w) { return w.v1 == kInf && w.v2 == kInf; }
inline bool Same(const Weight &a, const Weight &b) { return a.v1 == b.v1 && a.v2 == b.v2; }
int Compare(const Weight &a, const Weight &b);         // +1: a is better (smaller cost)
inline Weight Plus(const Weight &a, const Weight &b) { return Compare(a, b) >= 0 ? a : b; }
inline Weight Times(const Weight &a, const Weight &b) { return Weight{a.v1 + b.v1, a.v2 + b.v2}; }
Weight Divide(const Weight &a, const Weight &b);
bool ApproxEqual(const Weight &a, const Weight &b, float delta = kDelta);
inline double Cost(const Weight &w) { return (double)w.v1 + (double)w.v2; }

struct Arc { int32_t ilabel, olabel; Weight w; int32_t next; };
// A state's arcs: the first N in the state itself, more on the heap.  A raw lattice of the search has 1.6 arcs per state (9 k states per
// 30 s utterance), and the lattice stage rebuilds lattices of that size half a dozen times per utterance (Connect, phone insertion,
// TopSort, compact -> lattice, ...): with a std::vector per state most of that stage's time went to the allocator (round 5: 5.6 -> 3.1 ms
// per utterance on the build container's cores with N = 2).  The subset of std::vector's interface the host code uses; trivially
// copyable elements only.
template <class T, int N>
class SmallVec {
 public:
  SmallVec() {}
  SmallVec(const SmallVec &o) { assign(o.begin(), o.end()); }
  SmallVec(SmallVec &&o) noexcept { steal(o); }
  SmallVec &operator=(const SmallVec &o) { if (this != &o) assign(o.begin(), o.end()); return *this; }
  SmallVec &operator=(SmallVec &&o) noexcept { if (this != &o) { release(); steal(o); } return *this; }
  ~SmallVec() { release(); }
  size_t size() const { return n_; }
  bool empty() const { return n_ == 0; }
  T *data() { return heap_ ? heap_ : inl_; }
  const T *data() const { return heap_ ? heap_ : inl_; }
  T *begin() { return data(); }
  T *end() { return data() + n_; }
  const T *begin() const { return data(); }
  const T *end() const { return data() + n_; }
  T &operator[](size_t i) { return data()[i]; }
  const T &operator[](size_t i) const { return data()[i]; }
  T &back() { return data()[n_ - 1]; }
  const T &back() const { return data()[n_ - 1]; }
  void clear() { n_ = 0; }
  void reserve(size_t c) { if (c > cap_) grow(c); }
  void push_back(const T &v) { if (n_ == cap_) grow(cap_ * 2); data()[n_++] = v; }
  template <class... A> void emplace_back(A &&...a) { push_back(T{std::forward<A>(a)...}); }
  void resize(size_t n) { if (n > cap_) grow(n); for (size_t i = n_; i < n; i++) data()[i] = T(); n_ = (uint32_t)n; }
  template <class It> void assign(It b, It e) { n_ = 0; const size_t n = (size_t)(e - b); if (n > cap_) grow(n); T *d = data(); for (size_t i = 0; i < n; i++) d[i] = b[i]; n_ = (uint32_t)n; }
  void swap(SmallVec &o) { SmallVec t(std::move(o)); o = std::move(*this); *this = std::move(t); }
  void erase(T *first, T *last) { T *e = end(); n_ -= (uint32_t)(last - first); while (last != e) *first++ = *last++; }
  bool operator==(const SmallVec &o) const { if (n_ != o.n_) return false; const T *a = data(), *b = o.data(); for (size_t i = 0; i < n_; i++) if (!(a[i] == b[i])) return false; return true; }
 private:
  void grow(size_t c) {
    T *h = static_cast<T *>(::operator new(sizeof(T) * c));
    memcpy(static_cast<void *>(h), static_cast<const void *>(data()), sizeof(T) * n_);
    if (heap_) ::operator delete(heap_);
    heap_ = h; cap_ = (uint32_t)c;
  }
  void release() { if (heap_) ::operator delete(heap_); heap_ = nullptr; cap_ = N; n_ = 0; }
  void steal(SmallVec &o) {
    n_ = o.n_; cap_ = o.cap_; heap_ = o.heap_;
    if (!heap_) memcpy(static_cast<void *>(inl_), static_cast<const void *>(o.inl_), sizeof(T) * n_);
    o.heap_ = nullptr; o.cap_ = N; o.n_ = 0;
  }
  T *heap_ = nullptr;
  uint32_t n_ = 0, cap_ = N;
  T inl_[N];
};
typedef SmallVec<Arc, 2> ArcVec;
struct State { Weight final = Zero(); ArcVec arcs; };
struct Lattice {           // VectorFst<LatticeArc>
  int32_t start = -1;
  std::vector<State> st;
  size_t NumArcs() const { size_t n = 0; for (const State &s : st) n += s.arcs.size(); return n; }
};

// A compact lattice weight's string of transition-ids: up to two in the weight itself.  The compact form of a raw lattice (what the
// rescoring of DecodeFmllr works on: 15 k arcs per 30 s utterance) has one transition-id on nearly every arc -- a std::vector per arc was
// 15 k allocations to build and as many to free, per utterance, on sixteen threads at once; determinized lattices (an arc = a word's
// alignment) keep theirs on the heap as before.
typedef SmallVec<int32_t, 2> LabelStr;
struct CWeight { Weight w = Zero(); LabelStr str; };   // CompactLatticeWeight
struct CArc { int32_t label; CWeight w; int32_t next; };          // acceptor: ilabel == olabel == word
struct CState { CWeight final; std::vector<CArc> arcs; };
struct CompactLattice {    // VectorFst<CompactLatticeArc>
  int32_t start = -1;
  std::vector<CState> st;
  size_t NumArcs() const { size_t n = 0; for (const CState &s : st) n += s.arcs.size(); return n; }
};

// ---- graph operations (state numbering and arc order as OpenFst leaves them)
void Connect(Lattice *lat);
void Connect(CompactLattice *clat);
bool IsTopSorted(const Lattice &lat);             // every arc goes to a higher-numbered state
bool TopSort(Lattice *lat);                        // false (and unchanged) when cyclic
bool TopSort(CompactLattice *clat);
void Invert(Lattice *lat);
void ArcSortILabel(Lattice *lat);

// ---- determinization
struct DetOptions {        // DeterminizeLatticePrunedOptions
  float delta = kDelta;
  int max_mem = -1, max_loop = -1, max_states = -1, max_arcs = -1;
  float retry_cutoff = 0.5f;
};
struct PhoneDetOptions {   // DeterminizeLatticePhonePrunedOptions (gmm-latgen-faster registers max_mem 50000000)
  float delta = kDelta;
  int max_mem = 50000000;
  bool phone_determinize = true, word_determinize = true, minimize = false;
};
// what the phone-level pass needs of the TransitionModel, indexed by transition-id (index 0 unused)
struct TransInfo {
  std::vector<int32_t> phone;        // TransitionIdToPhone
  std::vector<int32_t> hmm_state;    // TransitionIdToHmmState
  std::vector<uint8_t> self_loop;    // IsSelfLoop
};
// input: words on the INPUT side (i.e. after Invert), topologically sorted
bool DeterminizeLatticePruned(const Lattice &ifst, double beam, CompactLattice *ofst, DetOptions opts = DetOptions());
bool DeterminizeLatticePruned(const Lattice &ifst, double beam, Lattice *ofst, DetOptions opts = DetOptions());
// raw lattice as the search leaves it (transition-ids on the input side); *ifst is consumed
bool DeterminizeLatticePhonePrunedWrapper(const TransInfo &tm, Lattice *ifst, double beam, CompactLattice *ofst,
                                          PhoneDetOptions opts = PhoneDetOptions());
bool PruneLattice(float beam, Lattice *lat);
// --minimize: transition-id strings and weights pushed towards the start, then equivalent states merged
// (kaldi-master/src/lat/push-lattice.cc, minimize-lattice.cc); false when the lattice cannot be sorted topologically
bool PushCompactLatticeStrings(CompactLattice *clat);
bool PushCompactLatticeWeights(CompactLattice *clat);
bool MinimizeCompactLattice(CompactLattice *clat, float delta = kDelta);

// ---- what scoring uses
void ScaleLattice(double s00, double s01, double s10, double s11, Lattice *lat);
void ScaleLattice(double s00, double s01, double s10, double s11, CompactLattice *clat);
void AddWordInsPen(float word_ins_penalty, CompactLattice *clat);
void ConvertLattice(const CompactLattice &clat, Lattice *lat, bool invert = true);   // :102-160
// state-level -> compact: chains of states are folded into one arc (fstext/factor-inl.h:64-140), then TopSort
void ConvertLattice(const Lattice &lat, CompactLattice *clat);                        // lattice-utils-inl.h:27-100 (invert = true)

// ---- times, posteriors, rescoring (what fMLLR estimation in DecodeFmllr reads)
typedef std::vector<std::vector<std::pair<int32_t, float>>> Posterior;   // per frame: (transition-id, posterior)
int32_t LatticeStateTimes(const Lattice &lat, std::vector<int32_t> *times);                  // lattice-functions.cc:78-110
int32_t CompactLatticeStateTimes(const CompactLattice &clat, std::vector<int32_t> *times);   // :112-150
// forward-backward over a top-sorted lattice (lattice-functions.cc:314-395); returns the total log-prob
double LatticeForwardBackward(const Lattice &lat, Posterior *post, double *acoustic_like_sum);
// adds -loglike(t, tid) to the acoustic cost of every transition-id of the lattice
// (RescoreCompactLattice, lattice-functions.cc:1214-1345); false on lattice/feature mismatch
bool RescoreCompactLattice(const std::function<float(int32_t, int32_t)> &loglike, int32_t num_frames, CompactLattice *clat);
// best path of a compact lattice as a linear compact lattice; empty when there is none
void CompactLatticeShortestPath(const CompactLattice &clat, CompactLattice *shortest);
// words, alignment and total weight of a linear compact lattice; false if it is not linear
bool GetLinearSymbolSequence(const CompactLattice &linear, std::vector<int32_t> *ali, std::vector<int32_t> *words, Weight *tot);

}  // namespace vblat
#endif
