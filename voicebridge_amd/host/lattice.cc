// lattice.cc -- see lattice.h for what each function follows in the reference.
#include "lattice.h"

#include <algorithm>
#include <cassert>
#include <cmath>
#include <functional>
#include <queue>
#include <stdexcept>
#include <unordered_map>
#include <utility>

namespace vblat {

// ------------------------------------------------------------------ LatticeWeight
// lattice-weight.h:295-310: smaller total cost is "greater" in the semiring; ties on value1
int Compare(const Weight &a, const Weight &b) {
  const float f1 = a.v1 + a.v2, f2 = b.v1 + b.v2;
  if (f1 < f2) return 1;
  if (f1 > f2) return -1;
  if (a.v1 < b.v1) return 1;
  if (a.v1 > b.v1) return -1;
  return 0;
}
// lattice-weight.h:343-359
Weight Divide(const Weight &a, const Weight &b) {
  const float x = a.v1 - b.v1, y = a.v2 - b.v2;
  if (x != x || y != y || x == -kInf || y == -kInf) return Zero();
  if (x == kInf || y == kInf) return Zero();
  return Weight{x, y};
}
// lattice-weight.h:362-367
bool ApproxEqual(const Weight &a, const Weight &b, float delta) {
  if (a.v1 == b.v1 && a.v2 == b.v2) return true;
  return std::fabs((a.v1 + a.v2) - (b.v1 + b.v2)) <= delta;
}
// CompactLatticeWeight comparison (lattice-weight.h:564-592): weights first, then the shorter
// string is "greater", then lexicographic order with smaller labels "less"
static int CompareStr(const std::vector<int32_t> &a, const std::vector<int32_t> &b) {
  const int la = (int)a.size(), lb = (int)b.size();
  if (la > lb) return -1;
  if (la < lb) return 1;
  for (int i = 0; i < la; i++) {
    if (a[i] < b[i]) return -1;
    if (a[i] > b[i]) return 1;
  }
  return 0;
}

// ------------------------------------------------------------------ depth-first order
namespace {

// Depth-first visit in OpenFst's order (dfs-visit.h:93-200): trees rooted at the start state,
// then at every still unvisited state in increasing order; arcs in stored order.  Records the
// finishing order and whether a back arc (cycle) was met (the visit stops there, as
// TopOrderVisitor::BackArc makes it).
template <class Fst>
bool DfsFinishOrder(const Fst &fst, std::vector<int32_t> *finish) {
  finish->clear();
  const int32_t n = (int32_t)fst.st.size();
  if (fst.start < 0) return true;
  std::vector<uint8_t> color(n, 0);
  std::vector<std::pair<int32_t, size_t>> stack;
  for (int32_t root = fst.start; root < n;) {
    color[root] = 1;
    stack.emplace_back(root, 0);
    while (!stack.empty()) {
      const int32_t s = stack.back().first;
      const size_t ai = stack.back().second;
      if (ai >= fst.st[s].arcs.size()) {
        color[s] = 2;
        stack.pop_back();
        finish->push_back(s);
        if (!stack.empty()) stack.back().second++;
        continue;
      }
      const int32_t nx = fst.st[s].arcs[ai].next;
      if (color[nx] == 0) {
        color[nx] = 1;
        stack.emplace_back(nx, 0);
      } else if (color[nx] == 1) {
        return false;   // back arc: cyclic
      } else {
        stack.back().second++;
      }
    }
    for (root = (root == fst.start) ? 0 : root + 1; root < n && color[root] != 0; ++root) {}
  }
  return true;
}

template <class Fst>
bool TopSortImpl(Fst *fst) {
  std::vector<int32_t> finish;
  if (!DfsFinishOrder(*fst, &finish)) return false;
  const int32_t n = (int32_t)fst->st.size();
  if (fst->start < 0) return true;
  std::vector<int32_t> order(n, -1);
  for (size_t i = 0; i < finish.size(); i++) order[finish[finish.size() - 1 - i]] = (int32_t)i;
  // StateSort (statesort.h): state s moves to order[s]; arcs keep their order
  decltype(fst->st) ns(n);
  for (int32_t s = 0; s < n; s++) {
    auto &dst = ns[order[s]];
    dst = std::move(fst->st[s]);
    for (auto &a : dst.arcs) a.next = order[a.next];
  }
  fst->st.swap(ns);
  fst->start = order[fst->start];
  return true;
}

template <class Fst, class IsFinal>
void ConnectImpl(Fst *fst, IsFinal is_final) {
  const int32_t n = (int32_t)fst->st.size();
  if (n == 0) return;
  if (fst->start < 0) { fst->st.clear(); return; }
  // accessible: reachable from the start state
  std::vector<uint8_t> access(n, 0), coaccess(n, 0);
  std::vector<int32_t> stack(1, fst->start);
  access[fst->start] = 1;
  while (!stack.empty()) {
    const int32_t s = stack.back(); stack.pop_back();
    for (const auto &a : fst->st[s].arcs) if (!access[a.next]) { access[a.next] = 1; stack.push_back(a.next); }
  }
  // coaccessible: some final state is reachable (reverse adjacency)
  std::vector<int32_t> rhead(n + 1, 0), rlist;
  for (int32_t s = 0; s < n; s++) for (const auto &a : fst->st[s].arcs) rhead[a.next + 1]++;
  for (int32_t s = 0; s < n; s++) rhead[s + 1] += rhead[s];
  rlist.resize(rhead[n]);
  { std::vector<int32_t> fill(rhead.begin(), rhead.end() - 1);
    for (int32_t s = 0; s < n; s++) for (const auto &a : fst->st[s].arcs) rlist[fill[a.next]++] = s; }
  for (int32_t s = 0; s < n; s++) if (is_final(fst->st[s])) { coaccess[s] = 1; stack.push_back(s); }
  while (!stack.empty()) {
    const int32_t s = stack.back(); stack.pop_back();
    for (int32_t i = rhead[s]; i < rhead[s + 1]; i++) if (!coaccess[rlist[i]]) { coaccess[rlist[i]] = 1; stack.push_back(rlist[i]); }
  }
  // DeleteStates (vector-fst.h): survivors keep their relative order, arcs into deleted states go
  std::vector<int32_t> newid(n, -1);
  int32_t m = 0;
  for (int32_t s = 0; s < n; s++) if (access[s] && coaccess[s]) newid[s] = m++;
  if (m == n) return;
  if (newid[fst->start] < 0) { fst->st.clear(); fst->start = -1; return; }
  decltype(fst->st) ns(m);
  for (int32_t s = 0; s < n; s++) {
    if (newid[s] < 0) continue;
    auto &dst = ns[newid[s]];
    dst.final = std::move(fst->st[s].final);
    for (auto &a : fst->st[s].arcs) if (newid[a.next] >= 0) { dst.arcs.push_back(std::move(a)); dst.arcs.back().next = newid[a.next]; }
  }
  fst->st.swap(ns);
  fst->start = newid[fst->start];
}

}  // namespace

void Connect(Lattice *lat) { ConnectImpl(lat, [](const State &s) { return !IsZero(s.final); }); }
void Connect(CompactLattice *clat) { ConnectImpl(clat, [](const CState &s) { return !IsZero(s.final.w); }); }
bool TopSort(Lattice *lat) { return TopSortImpl(lat); }
bool TopSort(CompactLattice *clat) { return TopSortImpl(clat); }
bool IsTopSorted(const Lattice &lat) {
  for (size_t s = 0; s < lat.st.size(); s++) for (const Arc &a : lat.st[s].arcs) if (a.next <= (int32_t)s) return false;
  return true;
}
static bool IsTopSorted(const CompactLattice &lat) {
  for (size_t s = 0; s < lat.st.size(); s++) for (const CArc &a : lat.st[s].arcs) if (a.next <= (int32_t)s) return false;
  return true;
}
void Invert(Lattice *lat) { for (State &s : lat->st) for (Arc &a : s.arcs) std::swap(a.ilabel, a.olabel); }
void ArcSortILabel(Lattice *lat) {
  // std::sort with ILabelCompare (arcsort.h:48,170-176): the same comparison sequence, hence the
  // same order among equal labels, as the reference built with this libstdc++
  for (State &s : lat->st) std::sort(s.arcs.begin(), s.arcs.end(), [](const Arc &a, const Arc &b) { return a.ilabel < b.ilabel; });
}

// ------------------------------------------------------------------ string repository
namespace {

// Strings of labels stored as a trie: id 0 is the empty string, an entry is (parent, last label).
// Entries are never freed, but the determinizer's memory check counts the entries the reference's
// repository would hold (determinize-lattice-inl.h:186-207: Rebuild() keeps the strings still in
// use and their prefixes, everything else is deleted and re-created on demand); `live` entries
// are exactly those, so the check trips at the same point as in the reference.
class StringRepo {
 public:
  StringRepo() { n_.push_back(Node{-1, 0, 0, 0, 0, 0, 0}); }
  int Empty() const { return 0; }
  size_t NumEntries() const { return live_count_; }
  int Successor(int a, int32_t l) {
    // most nodes of these tries have one child (alignments are chains): the first child of a node is kept beside it and
    // found without hashing; only further children go through the open-addressing table over (parent, label)
    const int fc = n_[a].child;
    if (fc == 0) {
      const int id = NewNode(a, l);
      n_[a].child = id;
      return id;
    }
    if (n_[fc].label == l) {
      if (n_[fc].gen != cur_gen_) { n_[fc].gen = cur_gen_; live_count_++; }
      return fc;
    }
    if ((n_hashed_ + 1) * 2 > table_.size()) Grow();
    const size_t mask = table_.size() - 1;
    const uint64_t key = ((uint64_t)(uint32_t)a << 32) | (uint32_t)l;
    size_t h = Hash(key) & mask;
    for (;;) {
      const Slot &sl = table_[h];
      if (sl.id < 0) break;
      if (sl.key == key) {
        if (n_[sl.id].gen != cur_gen_) { n_[sl.id].gen = cur_gen_; live_count_++; }
        return sl.id;
      }
      h = (h + 1) & mask;
    }
    const int id = NewNode(a, l);
    table_[h] = Slot{key, id};
    n_hashed_++;
    return id;
  }
  int NewNode(int a, int32_t l) {
    const int id = (int)n_.size();
    const int ja = n_[a].jump;
    n_.push_back(Node{a, l, n_[a].size + 1, cur_gen_, a == 0 ? l : n_[a].first,
                      n_[a].size - n_[ja].size == n_[ja].size - n_[n_[ja].jump].size ? n_[ja].jump : a, 0});
    live_count_++;
    return id;
  }
  void Rebuild(const std::vector<int> &to_keep) {
    cur_gen_++;
    live_count_ = 0;
    for (int e : to_keep)
      for (; e != 0 && n_[e].gen != cur_gen_; e = n_[e].parent) { n_[e].gen = cur_gen_; live_count_++; }
  }
  int Size(int a) const { return n_[a].size; }
  // the prefix of length n of string a: an ancestor in the trie (alive whenever a is: no lookup, no entry created --
  // what the reference obtains by re-converting the prefix vector, determinize-lattice-inl.h:161-166)
  int Prefix(int a, size_t n) const {          // level ancestor in O(log) steps through the jump pointers
    while ((size_t)n_[a].size > n) a = (size_t)n_[n_[a].jump].size >= n ? n_[a].jump : n_[a].parent;
    return a;
  }
  // longest common prefix of two strings = their lowest common ancestor in the trie
  int CommonPrefix(int a, int b) const {
    if (n_[a].size > n_[b].size) a = Prefix(a, n_[b].size); else if (n_[b].size > n_[a].size) b = Prefix(b, n_[a].size);
    while (a != b) {
      if (n_[a].jump != n_[b].jump) { a = n_[a].jump; b = n_[b].jump; } else { a = n_[a].parent; b = n_[b].parent; }
    }
    return a;
  }
  void ToVector(int a, std::vector<int32_t> *out) const {
    out->resize(n_[a].size);
    for (int i = n_[a].size - 1; i >= 0; i--) { (*out)[i] = n_[a].label; a = n_[a].parent; }
  }
  int FromVector(const std::vector<int32_t> &v) { int e = 0; for (int32_t l : v) e = Successor(e, l); return e; }
  int Concatenate(int a, int b) {
    if (a == 0) return b;
    if (b == 0) return a;
    std::vector<int32_t> v; ToVector(b, &v);
    for (int32_t l : v) a = Successor(a, l);
    return a;
  }
  // shrink *b to the longest common prefix of a and *b (determinize-lattice-inl.h:96-115)
  void ReduceToCommonPrefix(int a, std::vector<int32_t> *b) const {
    if (b->empty()) return;
    // strings that differ in their first label have no common prefix: no need to walk a's whole length to learn it
    if (n_[a].size == 0 || n_[a].first != (*b)[0]) { b->clear(); return; }
    size_t a_size = n_[a].size, b_size = b->size();
    while (a_size > b_size) { a = n_[a].parent; a_size--; }
    if (b_size > a_size) b_size = a_size;
    while (a_size != 0) {
      if (n_[a].label != (*b)[a_size - 1]) b_size = a_size - 1;
      a = n_[a].parent; a_size--;
    }
    if (b_size != b->size()) b->resize(b_size);
  }
  int RemovePrefix(int a, size_t n) {
    if (n == 0) return a;
    // only the suffix is walked (the reference converts the whole string, determinize-lattice-inl.h:117-126)
    tmp_.clear();
    for (int x = a; (size_t)n_[x].size > n; x = n_[x].parent) tmp_.push_back(n_[x].label);
    int e = 0;
    for (size_t i = tmp_.size(); i-- > 0;) e = Successor(e, tmp_[i]);
    return e;
  }
 private:
  struct Node { int parent; int32_t label; int size; uint32_t gen; int32_t first; int jump; int child; };   // first: first label of the string; jump: a second ancestor (level-ancestor queries in O(log)); child: first child (0: none)
  std::vector<Node> n_;
  std::vector<int32_t> tmp_;
  uint32_t cur_gen_ = 0;
  size_t live_count_ = 0;
  struct Slot { uint64_t key; int id; };
  std::vector<Slot> table_;
  size_t n_hashed_ = 0;                   // nodes reachable through the table only (second and later children)
  static size_t Hash(uint64_t k) { k ^= k >> 29; k *= 0x9E3779B97F4A7C15ull; return (size_t)(k >> 24); }
  void Grow() {
    const size_t n = table_.empty() ? (size_t)1 << 12 : table_.size() * 4;
    table_.assign(n, Slot{0, -1});
    const size_t mask = n - 1;
    for (int id = 1; id < (int)n_.size(); id++) {
      if (n_[n_[id].parent].child == id) continue;          // first children are not in the table
      const uint64_t key = ((uint64_t)(uint32_t)n_[id].parent << 32) | (uint32_t)n_[id].label;
      size_t h = Hash(key) & mask;
      while (table_[h].id >= 0) h = (h + 1) & mask;
      table_[h] = Slot{key, id};
    }
  }
};

// ------------------------------------------------------------------ the determinizer
// LatticeDeterminizerPruned (determinize-lattice-pruned.cc:44-1190).  Same procedure, same
// processing order (the task queue is a std::priority_queue fed in the same order, so equal
// priorities pop in the same order as in the reference); the data structures are this file's own.
class Determinizer {
 public:
  Determinizer(const Lattice &ifst, double beam, const DetOptions &opts)
      : ifst_(ifst), beam_(beam), opts_(opts), minimal_hash_(3, SubsetKey(), SubsetEqual{opts.delta}),
        initial_hash_(3, SubsetKey(), SubsetEqual{opts.delta}) {}
  ~Determinizer() {
    for (auto &kv : initial_hash_) delete kv.first;
    for (OutputState *s : out_) delete s;
    while (!queue_.empty()) { delete queue_.top(); queue_.pop(); }
  }

  bool Determinize(double *effective_beam) {
    Initialize();
    while (!queue_.empty()) {
      Task *task = queue_.top();
      const size_t num_states = out_.size();
      if ((opts_.max_states > 0 && num_states > (size_t)opts_.max_states) ||
          (opts_.max_arcs > 0 && num_arcs_ > opts_.max_arcs) ||
          (num_states % 10 == 0 && !CheckMemoryUsage()))
        break;
      queue_.pop();
      ProcessTransition(task->state, task->label, &task->subset);
      delete task;
    }
    if (effective_beam) {
      if (queue_.empty()) *effective_beam = beam_;
      else *effective_beam = queue_.top()->priority_cost - backward_[ifst_.start];
    }
    return queue_.empty();
  }

  void Output(CompactLattice *ofst) {      // :54-105
    ofst->st.clear(); ofst->start = -1;
    const int n = (int)out_.size();
    if (n == 0) return;
    ofst->st.resize(n);
    ofst->start = 0;
    std::vector<int32_t> seq_;
    for (int s = 0; s < n; s++) {
      for (const TempArc &t : out_[s]->arcs) {
        CWeight w; w.w = t.weight; repo_.ToVector(t.string, &seq_); w.str.assign(seq_.begin(), seq_.end());
        if (t.next < 0) ofst->st[s].final = std::move(w);
        else ofst->st[s].arcs.push_back(CArc{t.ilabel, std::move(w), t.next});
      }
    }
  }
  void Output(Lattice *ofst) {             // :110-180: strings become chains of extra states
    ofst->st.clear(); ofst->start = -1;
    const int n = (int)out_.size();
    if (n == 0) return;
    ofst->st.resize(n);
    ofst->start = 0;
    std::vector<int32_t> seq;
    for (int s = 0; s < n; s++) {
      for (const TempArc &t : out_[s]->arcs) {
        repo_.ToVector(t.string, &seq);
        if (t.next < 0) {
          int cur = s;
          for (size_t i = 0; i < seq.size(); i++) {
            const int nx = (int)ofst->st.size();
            ofst->st.emplace_back();
            ofst->st[cur].arcs.push_back(Arc{0, seq[i], i == 0 ? t.weight : One(), nx});
            cur = nx;
          }
          ofst->st[cur].final = seq.empty() ? t.weight : One();
        } else {
          int cur = s;
          for (size_t i = 0; i + 1 < seq.size(); i++) {
            const int nx = (int)ofst->st.size();
            ofst->st.emplace_back();
            ofst->st[cur].arcs.push_back(Arc{i == 0 ? t.ilabel : 0, seq[i], i == 0 ? t.weight : One(), nx});
            cur = nx;
          }
          ofst->st[cur].arcs.push_back(Arc{seq.size() <= 1 ? t.ilabel : 0, seq.empty() ? 0 : seq.back(),
                                           seq.size() <= 1 ? t.weight : One(), t.next});
        }
      }
    }
  }

 private:
  struct Element {
    int32_t state; int string; Weight weight;
    bool Differs(const Element &o) const { return state != o.state || string != o.string || !Same(weight, o.weight); }
  };
  struct ElemGreaterState { bool operator()(const Element &a, const Element &b) const { return a.state > b.state; } };
  struct TempArc { int32_t ilabel; int string; int32_t next; Weight weight; };
  struct OutputState {
    std::vector<Element> minimal_subset;
    std::vector<TempArc> arcs;
    double forward_cost;
  };
  struct Task { int32_t state; int32_t label; std::vector<Element> subset; double priority_cost; };
  struct TaskWorse { bool operator()(const Task *a, const Task *b) const { return a->priority_cost > b->priority_cost; } };
  struct SubsetKey {   // states and strings only; weights are compared with a tolerance (:428-441)
    size_t operator()(const std::vector<Element> *s) const {
      uint64_t h = 0x9E3779B97F4A7C15ull;        // (order-dependent mix of the (state, string) pairs; any hash would do: equality decides)
      for (const Element &e : *s) { h ^= (uint64_t)(uint32_t)e.state | ((uint64_t)(uint32_t)e.string << 32); h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 31; }
      return (size_t)h;
    }
  };
  struct SubsetEqual {
    float delta;
    bool operator()(const std::vector<Element> *a, const std::vector<Element> *b) const {
      if (a->size() != b->size()) return false;
      for (size_t i = 0; i < a->size(); i++)
        if ((*a)[i].state != (*b)[i].state || (*a)[i].string != (*b)[i].string || !ApproxEqual((*a)[i].weight, (*b)[i].weight, delta)) return false;
      return true;
    }
  };
  typedef std::unordered_map<const std::vector<Element> *, int32_t, SubsetKey, SubsetEqual> MinimalHash;
  typedef std::unordered_map<const std::vector<Element> *, Element, SubsetKey, SubsetEqual> InitialHash;

  int CompareWS(const Weight &aw, int as, const Weight &bw, int bs) const {   // :583-608
    const int c = Compare(aw, bw);
    if (c != 0) return c;
    if (as == bs) return 0;
    std::vector<int32_t> av, bv;
    repo_.ToVector(as, &av); repo_.ToVector(bs, &bv);
    return CompareStr(av, bv);
  }

  // :252-340.  Sizes as the reference counts them on x86-64: repository entry 2 x 16 bytes,
  // TempArc 32, Element 24.
  bool CheckMemoryUsage() {
    if (opts_.max_mem <= 0) return true;
    const int32_t repo = (int32_t)(repo_.NumEntries() * 32), arcs = num_arcs_ * 32, elems = num_elems_ * 24;
    if (repo + arcs + elems <= opts_.max_mem) return true;
    // RebuildRepository: everything still referenced -- subsets and arcs of the output states, the
    // queued tasks, keys and values of the initial-subset hash
    std::vector<int> needed;
    for (const OutputState *os : out_) {
      for (const Element &e : os->minimal_subset) needed.push_back(e.string);
      for (const TempArc &a : os->arcs) needed.push_back(a.string);
    }
    {
      std::vector<Task *> tasks;
      while (!queue_.empty()) { tasks.push_back(queue_.top()); queue_.pop(); }
      for (Task *t : tasks) { for (const Element &e : t->subset) needed.push_back(e.string); }
      for (Task *t : tasks) queue_.push(t);
    }
    for (const auto &kv : initial_hash_) {
      for (const Element &e : *kv.first) needed.push_back(e.string);
      needed.push_back(kv.second.string);
    }
    repo_.Rebuild(needed);
    const int32_t new_repo = (int32_t)(repo_.NumEntries() * 32);
    return new_repo + arcs + elems <= (int32_t)(opts_.max_mem * 0.8);
  }

  void ComputeBackward() {        // :1013-1036
    const int n = (int)ifst_.st.size();
    backward_.assign(n, 0.0);
    for (int s = n - 1; s >= 0; s--) {
      double cost = Cost(ifst_.st[s].final);
      for (const Arc &a : ifst_.st[s].arcs) cost = std::min(cost, Cost(a.w) + backward_[a.next]);
      backward_[s] = cost;
    }
    if (ifst_.start < 0) return;
    cutoff_ = backward_[ifst_.start] + beam_;
  }

  bool IsIsymbolOrFinal(int32_t s) {    // :985-1010
    if (isym_or_final_.size() <= (size_t)s) isym_or_final_.resize(s + 1, 0);
    if (isym_or_final_[s] == 1) return false;
    if (isym_or_final_[s] == 2) return true;
    isym_or_final_[s] = 1;
    if (!IsZero(ifst_.st[s].final)) isym_or_final_[s] = 2;
    for (const Arc &a : ifst_.st[s].arcs) if (a.ilabel != 0 && !IsZero(a.w)) { isym_or_final_[s] = 2; break; }
    return isym_or_final_[s] == 2;
  }
  void ConvertToMinimal(std::vector<Element> *subset) {
    size_t o = 0;
    for (size_t i = 0; i < subset->size(); i++) if (IsIsymbolOrFinal((*subset)[i].state)) (*subset)[o++] = (*subset)[i];
    subset->resize(o);
  }

  void EpsilonClosure(std::vector<Element> *subset) {    // :614-725
    std::priority_queue<Element, std::vector<Element>, ElemGreaterState> queue;
    // state -> position in `cur` (the reference keeps an unordered_map; only membership and the current value matter)
    if (ec_pos_.size() < ifst_.st.size()) { ec_pos_.assign(ifst_.st.size(), 0); ec_stamp_.assign(ifst_.st.size(), 0); }
    ec_gen_++;
    std::vector<Element> &cur = ec_cur_;
    cur.clear();
    auto find = [&](int32_t st) -> Element * { return ec_stamp_[st] == ec_gen_ ? &cur[ec_pos_[st]] : nullptr; };
    auto insert = [&](const Element &e) { ec_stamp_[e.state] = ec_gen_; ec_pos_[e.state] = (int)cur.size(); cur.push_back(e); };
    for (const Element &e : *subset) { queue.push(e); insert(e); }
    bool replaced = false;
    int counter = 0;
    while (!queue.empty()) {
      const Element elem = queue.top();
      queue.pop();
      if (replaced && find(elem.state)->Differs(elem)) continue;
      if (opts_.max_loop > 0 && counter++ > opts_.max_loop)
        throw std::runtime_error("lattice determinization: looped more than max-loop times during epsilon closure");
      for (const Arc &arc : ifst_.st[elem.state].arcs) {
        if (arc.ilabel != 0 || IsZero(arc.w)) continue;
        Element next;
        next.state = arc.next;
        next.weight = Times(elem.weight, arc.w);
        Element *it = find(next.state);
        if (it == nullptr) {
          next.string = arc.olabel == 0 ? elem.string : repo_.Successor(elem.string, arc.olabel);
          insert(next);
          queue.push(next);
        } else {
          int comp = Compare(next.weight, it->weight);
          if (comp == 0) {
            next.string = arc.olabel == 0 ? elem.string : repo_.Successor(elem.string, arc.olabel);
            comp = CompareWS(next.weight, next.string, it->weight, it->string);
          }
          if (comp == 1) {
            next.string = arc.olabel == 0 ? elem.string : repo_.Successor(elem.string, arc.olabel);
            it->string = next.string;
            it->weight = next.weight;
            queue.push(next);
            replaced = true;
          }
        }
      }
    }
    *subset = cur;
    std::sort(subset->begin(), subset->end(), [](const Element &a, const Element &b) { return a.state < b.state; });
  }

  void ProcessFinal(int32_t oid) {      // :733-770
    OutputState &st = *out_[oid];
    int final_string = repo_.Empty();
    Weight final_weight = Zero();
    bool is_final = false;
    for (const Element &e : st.minimal_subset) {
      const Weight w = Times(e.weight, ifst_.st[e.state].final);
      if (!IsZero(w) && (!is_final || CompareWS(w, e.string, final_weight, final_string) == 1)) {
        is_final = true; final_weight = w; final_string = e.string;
      }
    }
    if (is_final && Cost(final_weight) + st.forward_cost <= cutoff_) {
      st.arcs.push_back(TempArc{0, final_string, -1, final_weight});
      num_arcs_++;
    }
  }

  void NormalizeSubset(std::vector<Element> *elems, Weight *tot, int *common) {   // :775-803
    if (elems->empty()) { *common = repo_.Empty(); *tot = Zero(); return; }
    // the common prefix of the elements' strings is the lowest common ancestor of their trie nodes (the reference
    // shrinks a vector against every string, determinize-lattice-inl.h:96-115: the same prefix, found without
    // walking the strings)
    int pre = (*elems)[0].string;
    Weight w = (*elems)[0].weight;
    for (size_t i = 1; i < elems->size(); i++) {
      w = Plus(w, (*elems)[i].weight);
      if (pre != 0) pre = repo_.CommonPrefix(pre, (*elems)[i].string);
    }
    const size_t plen = (size_t)repo_.Size(pre);
    for (Element &e : *elems) {
      e.weight = Divide(e.weight, w);
      e.string = repo_.RemovePrefix(e.string, plen);
    }
    *common = pre;
    *tot = w;
  }

  void MakeSubsetUnique(std::vector<Element> *subset) {     // :808-835
    size_t in = 0, out = 0;
    const size_t end = subset->size();
    while (in != end) {
      if (in != out) (*subset)[out] = (*subset)[in];
      in++;
      while (in != end && (*subset)[in].state == (*subset)[out].state) {
        if (CompareWS((*subset)[in].weight, (*subset)[in].string, (*subset)[out].weight, (*subset)[out].string) == 1) {
          (*subset)[out].string = (*subset)[in].string;
          (*subset)[out].weight = (*subset)[in].weight;
        }
        in++;
      }
      out++;
    }
    subset->resize(out);
  }

  int32_t MinimalToStateId(const std::vector<Element> &subset, double forward_cost) {    // :505-530
    auto it = minimal_hash_.find(&subset);
    if (it != minimal_hash_.end()) return it->second;
    const int32_t id = (int32_t)out_.size();
    OutputState *ns = new OutputState{subset, {}, forward_cost};
    minimal_hash_[&ns->minimal_subset] = id;
    out_.push_back(ns);
    num_elems_ += (int)subset.size();
    ProcessFinal(id);
    ProcessTransitions(id);
    return id;
  }

  int32_t InitialToStateId(const std::vector<Element> &subset_in, double forward_cost, Weight *remaining, int *common_prefix) {   // :535-578
    auto it = initial_hash_.find(&subset_in);
    if (it != initial_hash_.end()) {
      *remaining = it->second.weight; *common_prefix = it->second.string;
      return it->second.state;
    }
    std::vector<Element> subset(subset_in);
    EpsilonClosure(&subset);
    ConvertToMinimal(&subset);
    Element elem;
    NormalizeSubset(&subset, &elem.weight, &elem.string);
    forward_cost += Cost(elem.weight);
    const int32_t ans = MinimalToStateId(subset, forward_cost);
    *remaining = elem.weight; *common_prefix = elem.string;
    std::vector<Element> *key = new std::vector<Element>(subset_in);
    elem.state = ans;
    initial_hash_[key] = elem;
    num_elems_ += (int)key->size();
    return ans;
  }

  void ProcessTransition(int32_t oid, int32_t ilabel, std::vector<Element> *subset) {    // :845-875
    double forward_cost = out_[oid]->forward_cost;
    int common; Weight tot;
    NormalizeSubset(subset, &tot, &common);
    forward_cost += Cost(tot);
    Weight next_tot; int next_common;
    const int32_t next = InitialToStateId(*subset, forward_cost, &next_tot, &next_common);
    common = repo_.Concatenate(common, next_common);
    tot = Times(tot, next_tot);
    out_[oid]->arcs.push_back(TempArc{ilabel, common, next, tot});
    num_arcs_++;
  }

  void ProcessTransitions(int32_t oid) {     // :903-980
    std::vector<std::pair<int32_t, Element>> &all = all_elems_tmp_;
    // (the subset is copied: the recursion below may grow out_)
    {
      const std::vector<Element> &ms = out_[oid]->minimal_subset;
      for (const Element &e : ms) {
        for (const Arc &arc : ifst_.st[e.state].arcs) {
          if (arc.ilabel == 0 || IsZero(arc.w)) continue;
          Element ne;
          ne.state = arc.next;
          ne.weight = Times(e.weight, arc.w);
          ne.string = arc.olabel == 0 ? e.string : repo_.Successor(e.string, arc.olabel);
          all.emplace_back(arc.ilabel, ne);
        }
      }
    }
    std::sort(all.begin(), all.end(), [](const std::pair<int32_t, Element> &a, const std::pair<int32_t, Element> &b) {
      if (a.first < b.first) return true;
      if (a.first > b.first) return false;
      return a.second.state < b.second.state;
    });
    size_t cur = 0;
    const size_t end = all.size();
    while (cur != end) {
      Task *task = new Task;
      const int32_t ilabel = all[cur].first;
      task->state = oid;
      task->priority_cost = std::numeric_limits<double>::infinity();
      task->label = ilabel;
      while (cur != end && all[cur].first == ilabel) {
        const Element &e = all[cur].second;
        task->subset.push_back(e);
        task->priority_cost = std::min(task->priority_cost, Cost(e.weight) + backward_[e.state]);
        cur++;
      }
      task->priority_cost += out_[oid]->forward_cost;
      if (task->priority_cost > cutoff_) delete task;
      else { MakeSubsetUnique(&task->subset); queue_.push(task); }
    }
    all.clear();
  }

  void Initialize() {       // :1040-1085
    ComputeBackward();
    minimal_hash_.rehash(ifst_.st.size() / 2 + 3);
    initial_hash_.rehash(ifst_.st.size() / 2 + 3);
    if (ifst_.start < 0) return;
    std::vector<Element> subset(1);
    subset[0].state = ifst_.start; subset[0].weight = One(); subset[0].string = repo_.Empty();
    EpsilonClosure(&subset);
    ConvertToMinimal(&subset);
    OutputState *init = new OutputState{subset, {}, 0.0};
    out_.push_back(init);
    num_elems_ += (int)subset.size();
    minimal_hash_[&init->minimal_subset] = 0;
    ProcessFinal(0);
    ProcessTransitions(0);
  }

  const Lattice &ifst_;
  double beam_, cutoff_ = 0.0;
  DetOptions opts_;
  std::vector<OutputState *> out_;
  int num_arcs_ = 0, num_elems_ = 0;
  std::vector<double> backward_;
  MinimalHash minimal_hash_;
  InitialHash initial_hash_;
  std::priority_queue<Task *, std::vector<Task *>, TaskWorse> queue_;
  std::vector<std::pair<int32_t, Element>> all_elems_tmp_;
  std::vector<uint8_t> isym_or_final_;
  std::vector<int> ec_pos_; std::vector<uint32_t> ec_stamp_; uint32_t ec_gen_ = 0; std::vector<Element> ec_cur_;      // EpsilonClosure's index
  StringRepo repo_;
};

template <class Out>
bool DeterminizePrunedImpl(const Lattice &ifst, double beam, Out *ofst, DetOptions opts, bool infinite_beam_skips_retry) {
  if (ifst.st.empty()) { ofst->st.clear(); ofst->start = -1; return true; }
  const int max_iters = 10;
  Lattice temp;
  for (int iter = 0; iter < max_iters; iter++) {
    Determinizer det(iter == 0 ? ifst : temp, beam, opts);
    double effective_beam;
    const bool ans = det.Determinize(&effective_beam);
    if (effective_beam >= beam * opts.retry_cutoff || (infinite_beam_skips_retry && beam == std::numeric_limits<double>::infinity()) ||
        iter + 1 == max_iters) {
      det.Output(ofst);
      return ans;
    }
    if (effective_beam < 0.0) effective_beam = 0.0;
    double new_beam = beam * std::sqrt(effective_beam / beam);
    if (new_beam < 0.5 * beam) new_beam = 0.5 * beam;
    beam = new_beam;
    if (iter == 0) temp = ifst;
    PruneLattice((float)beam, &temp);
  }
  return false;
}

}  // namespace

bool DeterminizeLatticePruned(const Lattice &ifst, double beam, CompactLattice *ofst, DetOptions opts) {
  return DeterminizePrunedImpl(ifst, beam, ofst, opts, true);
}
bool DeterminizeLatticePruned(const Lattice &ifst, double beam, Lattice *ofst, DetOptions opts) {
  // (ifst and ofst may be the same object in the reference's first pass; here the result is built aside)
  Lattice out;
  const bool ans = DeterminizePrunedImpl(ifst, beam, &out, opts, false);
  *ofst = std::move(out);
  return ans;
}

// determinize-lattice-pruned.cc:1283-1340 (insert), :1342-1365 (delete), :1385-1404 (first pass),
// :1408-1470 (both passes), :1487-1507 (wrapper)
bool DeterminizeLatticePhonePrunedWrapper(const TransInfo &tm, Lattice *ifst, double beam, CompactLattice *ofst, PhoneDetOptions opts) {
  bool ans = true;
  Invert(ifst);
  if (!IsTopSorted(*ifst)) {
    if (!TopSort(ifst))
      throw std::runtime_error("Topological sorting of state-level lattice failed (probably your lexicon has empty words or your LM has epsilon cycles).");
  }
  ArcSortILabel(ifst);
  // words are on the input side here; the conversion to a compact lattice wants the transition-ids there (ConvertLattice with
  // invert = false inverts a copy, K/fstext/lattice-utils-inl.h:41-49)
  auto convert_as_is = [&]() { Lattice back = *ifst; Invert(&back); ConvertLattice(back, ofst); Connect(ofst); };
  if (!opts.phone_determinize && !opts.word_determinize) { convert_as_is(); return ans; }      // (:1423-1429: copied without determinization)
  DetOptions det;
  det.delta = opts.delta;
  det.max_mem = opts.max_mem;
  if (opts.phone_determinize) {
    // phone symbols on the word side at every phone boundary (first transition-id of a phone)
    int32_t highest = 0;
    for (const State &s : ifst->st) for (const Arc &a : s.arcs) highest = std::max(highest, a.ilabel);
    const int32_t first_phone_label = highest + 1;
    const int32_t n0 = (int32_t)ifst->st.size();
    for (int32_t s = 0; s < n0; s++) {
      if (s == ifst->start) continue;
      const size_t na = ifst->st[s].arcs.size();
      for (size_t i = 0; i < na; i++) {
        Arc arc = ifst->st[s].arcs[i];
        if (arc.olabel != 0) {
          if (arc.olabel < 0 || (size_t)arc.olabel >= tm.phone.size())
            throw std::runtime_error("lattice determinization: transition-id out of the model's range");
          if (tm.hmm_state[arc.olabel] == 0 && !tm.self_loop[arc.olabel]) {
            const int32_t phone = tm.phone[arc.olabel];
            if (arc.ilabel == 0) arc.ilabel = first_phone_label + phone;
            else {
              const int32_t extra = (int32_t)ifst->st.size();
              ifst->st.emplace_back();
              ifst->st[extra].arcs.push_back(Arc{first_phone_label + phone, 0, One(), arc.next});
              arc.next = extra;
            }
          }
        }
        ifst->st[s].arcs[i] = arc;
      }
    }
    TopSort(ifst);
    ans = DeterminizeLatticePruned(*ifst, beam, ifst, det) && ans;
    for (State &s : ifst->st) for (Arc &a : s.arcs) if (a.ilabel >= first_phone_label) a.ilabel = 0;
    TopSort(ifst);
    if (!opts.word_determinize) { convert_as_is(); return ans; }        // (:1445-1450)
  }
  ans = DeterminizeLatticePruned(*ifst, beam, ofst, det) && ans;
  if (opts.minimize) {
    ans = PushCompactLatticeStrings(ofst) && ans;
    ans = PushCompactLatticeWeights(ofst) && ans;
    ans = MinimizeCompactLattice(ofst, kDelta) && ans;
  }
  Connect(ofst);
  return ans;
}

// ---- push and minimize (what --minimize adds after determinization: K/lat/determinize-lattice-pruned.cc:1460-1466,
//      K/latbin/lattice-determinize-pruned.cc:111-115)
static bool IsZeroC(const CWeight &w) { return IsZero(w.w) && w.str.empty(); }
static bool SortedOrSort(CompactLattice *clat) { return IsTopSorted(*clat) || TopSort(clat); }

namespace {
// The first `len` transition-ids met from `state` on: along arc `arc_idx`, or (arc_idx < 0) along "any" path, which is the
// state's final string when it is final and its first arc otherwise (CompactLatticePusher::GetString, K/lat/push-lattice.cc:53-86)
void StringFrom(const CompactLattice &clat, int32_t state, int arc_idx, int32_t *out, size_t len) {
  while (len != 0) {
    const CState &st = clat.st[state];
    if (arc_idx < 0 && !IsZeroC(st.final)) {
      if (st.final.str.size() < len) throw std::runtime_error("push: paths in the lattice have inconsistent lengths");
      std::copy(st.final.str.begin(), st.final.str.begin() + len, out);
      return;
    }
    const size_t a = arc_idx < 0 ? 0 : (size_t)arc_idx;
    if (a >= st.arcs.size()) throw std::runtime_error("push: paths in the lattice are inconsistent in length");
    const LabelStr &str = st.arcs[a].w.str;
    if (str.size() >= len) { std::copy(str.begin(), str.begin() + len, out); return; }
    std::copy(str.begin(), str.end(), out);
    out += str.size(); len -= str.size();
    state = st.arcs[a].next; arc_idx = -1;
  }
}
}  // namespace

// K/lat/push-lattice.cc:30-228: every state hands back to its incoming arcs the transition-ids that all of its continuations
// start with (the longest common prefix, found from the last state to the first), so that equal futures look equal
bool PushCompactLatticeStrings(CompactLattice *clat) {
  if (!SortedOrSort(clat)) return false;
  const int32_t n = (int32_t)clat->st.size();
  std::vector<int32_t> shift(n, 0);
  std::vector<int32_t> a, b;
  for (int32_t s = n - 1; s > clat->start; s--) {
    const CState &st = clat->st[s];
    if (st.arcs.empty()) { shift[s] = (int32_t)st.final.str.size(); continue; }
    const bool is_final = !IsZeroC(st.final);
    int32_t sh = std::numeric_limits<int32_t>::max();
    if (is_final) sh = std::min(sh, (int32_t)st.final.str.size());
    for (const CArc &arc : st.arcs) sh = std::min(sh, shift[arc.next] + (int32_t)arc.w.str.size());
    if (st.arcs.size() + (is_final ? 1 : 0) > 1 && sh > 0) {      // CheckForConflict (:88-131)
      a.assign(sh, 0); b.assign(sh, 0);
      size_t arc;
      if (is_final) { std::copy(st.final.str.begin(), st.final.str.begin() + sh, a.begin()); arc = 0; }
      else { StringFrom(*clat, s, 0, a.data(), a.size()); arc = 1; }
      for (; arc < st.arcs.size(); arc++) {
        StringFrom(*clat, s, (int)arc, b.data(), b.size());
        const size_t same = std::mismatch(a.begin(), a.end(), b.begin()).first - a.begin();
        if (same != a.size()) { sh = (int32_t)same; a.resize(same); b.resize(same); }
      }
    }
    shift[s] = sh;
  }
  std::vector<int32_t> str;
  for (int32_t s = 0; s < n; s++) {        // ApplyShifts (:170-201)
    const int32_t sh = shift[s];
    CState &st = clat->st[s];
    for (CArc &arc : st.arcs) {
      if (arc.next <= s) throw std::runtime_error("push: cyclic lattice");
      str.assign(arc.w.str.begin(), arc.w.str.end());
      const size_t orig = str.size(), more = (size_t)shift[arc.next];
      str.resize(orig + more);
      StringFrom(*clat, arc.next, -1, str.data() + orig, more);
      arc.w.str.assign(str.begin() + sh, str.end());
    }
    if (!IsZeroC(st.final)) st.final.str.erase(st.final.str.begin(), st.final.str.begin() + sh);
  }
  return true;
}

// K/lat/push-lattice.cc:239-289: weights pushed towards the start (each state's best cost to the end becomes zero)
bool PushCompactLatticeWeights(CompactLattice *clat) {
  if (!SortedOrSort(clat)) return false;
  const int32_t n = (int32_t)clat->st.size();
  if (n == 0) return true;
  std::vector<Weight> to_end(n);
  for (int32_t s = n - 1; s >= 0; s--) {
    Weight w = clat->st[s].final.w;
    for (const CArc &arc : clat->st[s].arcs) {
      if (arc.next <= s) throw std::runtime_error("push: cyclic lattices not allowed");
      w = Plus(w, Times(arc.w.w, to_end[arc.next]));
    }
    to_end[s] = w;
  }
  to_end[0] = One();       // the leftover stays on the start state
  for (int32_t s = 0; s < n; s++) {
    const Weight here = to_end[s];
    if (IsZero(here)) continue;
    for (CArc &arc : clat->st[s].arcs)
      if (!IsZero(to_end[arc.next])) arc.w.w = Times(arc.w.w, Divide(to_end[arc.next], here));
    if (!IsZeroC(clat->st[s].final)) clat->st[s].final.w = Divide(clat->st[s].final.w, here);
  }
  return true;
}

// K/lat/minimize-lattice.cc:38-270: from the last state to the first, a state joins the first later state (among those with
// the same hash of final string, labels, strings and successor classes) that has approximately the same final weight and, arc
// by arc in label order, the same label, string, successor class and approximately the same weight.  The hash only narrows the
// search -- which states merge does not depend on it -- so it need not be the reference's.
bool MinimizeCompactLattice(CompactLattice *clat, float delta) {
  if (!SortedOrSort(clat)) return false;
  const int32_t n = (int32_t)clat->st.size();
  auto hash_str = [](const LabelStr &v) { uint64_t h = 1469598103934665603ull; for (int32_t x : v) h = (h ^ (uint32_t)x) * 1099511628211ull; return h | 1ull; };
  std::vector<uint64_t> hash(n);
  for (int32_t s = n - 1; s >= 0; s--) {
    const CState &st = clat->st[s];
    // (mixing: splitmix64's multipliers; a sum over the arcs, so that their order does not matter)
    uint64_t h = IsZeroC(st.final) ? 0x632BE59BD9B4E019ull : 0xBF58476D1CE4E5B9ull * hash_str(st.final.str);
    for (const CArc &arc : st.arcs)
      h += 0x94D049BB133111EBull * ((uint64_t)(uint32_t)arc.label + 0x9E3779B9ull) * (1ull + hash_str(arc.w.str) * (arc.next > s ? hash[arc.next] : 1ull));
    hash[s] = h;
  }
  std::unordered_map<uint64_t, std::vector<int32_t>> groups;
  for (int32_t s = 0; s < n; s++) groups[hash[s]].push_back(s);
  std::vector<int32_t> map(n);
  for (int32_t s = 0; s < n; s++) map[s] = s;
  struct Key { int32_t label, next; const CArc *arc; };
  auto sorted_arcs = [&](int32_t state, std::vector<Key> *out) {
    out->clear();
    for (const CArc &arc : clat->st[state].arcs) {
      if (arc.next == state) continue;                  // (self-loops are left out of the comparison, :163-167)
      if (arc.next < state) throw std::runtime_error("minimize: lattice not topologically sorted");
      out->push_back(Key{arc.label, map[arc.next], &arc});
    }
    std::sort(out->begin(), out->end(), [](const Key &x, const Key &y) { return x.label != y.label ? x.label < y.label : x.next < y.next; });
  };
  std::vector<Key> ka, kb;
  auto equivalent = [&](int32_t s, int32_t t) {
    const CState &a = clat->st[s], &b = clat->st[t];
    if (!(ApproxEqual(a.final.w, b.final.w, delta) && a.final.str == b.final.str)) return false;
    if (a.arcs.size() != b.arcs.size()) return false;
    sorted_arcs(s, &ka); sorted_arcs(t, &kb);
    if (ka.size() != kb.size()) throw std::runtime_error("minimize: self-loops in the lattice");
    for (size_t i = 0; i < ka.size(); i++) {
      if (ka[i].next != kb[i].next || ka[i].label != kb[i].label) return false;
      if (!(ApproxEqual(ka[i].arc->w.w, kb[i].arc->w.w) && ka[i].arc->w.str == kb[i].arc->w.str)) return false;
    }
    return true;
  };
  int32_t removed = 0;
  for (int32_t s = n - 1; s >= 0; s--)
    for (int32_t t : groups[hash[s]])
      if (t > s && map[t] == t && equivalent(s, t)) { map[s] = t; removed++; break; }
  if (removed == 0) return true;
  if (clat->start >= 0) clat->start = map[clat->start];
  for (int32_t s = 0; s < n; s++) {
    if (map[s] != s) continue;
    for (CArc &arc : clat->st[s].arcs) arc.next = map[arc.next];
  }
  Connect(clat);
  return true;
}

bool PruneLattice(float beam, Lattice *lat) {
  if (!IsTopSorted(*lat)) { if (!TopSort(lat)) return false; }
  const int32_t n = (int32_t)lat->st.size();
  if (n == 0) return false;
  const double inf = std::numeric_limits<double>::infinity();
  std::vector<double> fwd(n, inf);
  fwd[lat->start] = 0.0;
  double best_final = inf;
  for (int32_t s = 0; s < n; s++) {
    const double f = fwd[s];
    for (const Arc &a : lat->st[s].arcs) { const double nf = f + Cost(a.w); if (fwd[a.next] > nf) fwd[a.next] = nf; }
    const double tf = f + Cost(lat->st[s].final);
    if (tf < best_final) best_final = tf;
  }
  const int32_t bad = n;
  lat->st.emplace_back();
  const double cutoff = best_final + beam;
  std::vector<double> &bwd = fwd;
  for (int32_t s = n - 1; s >= 0; s--) {
    const double f = fwd[s];
    double b = Cost(lat->st[s].final);
    if (b + f > cutoff && b != inf) lat->st[s].final = Zero();
    for (Arc &a : lat->st[s].arcs) {
      const double ab = Cost(a.w) + bwd[a.next];
      if (ab < b) b = ab;
      if (f + ab > cutoff) a.next = bad;
    }
    bwd[s] = b;
  }
  Connect(lat);
  return !lat->st.empty();
}

// ------------------------------------------------------------------ scoring
static inline Weight ScaleW(const Weight &w, double s00, double s01, double s10, double s11) {   // lattice-weight.h:232-242
  if (w.v1 == kInf) return Zero();
  return Weight{(float)(s00 * w.v1 + s01 * w.v2), (float)(s10 * w.v1 + s11 * w.v2)};
}
void ScaleLattice(double s00, double s01, double s10, double s11, Lattice *lat) {
  if (s00 == 1.0 && s01 == 0.0 && s10 == 0.0 && s11 == 1.0) return;      // lattice-utils-inl.h: ScaleLattice returns early on the identity
  for (State &s : lat->st) {
    for (Arc &a : s.arcs) a.w = ScaleW(a.w, s00, s01, s10, s11);
    if (!IsZero(s.final)) s.final = ScaleW(s.final, s00, s01, s10, s11);
  }
}
void ScaleLattice(double s00, double s01, double s10, double s11, CompactLattice *clat) {
  if (s00 == 1.0 && s01 == 0.0 && s10 == 0.0 && s11 == 1.0) return;
  for (CState &s : clat->st) {
    for (CArc &a : s.arcs) a.w.w = ScaleW(a.w.w, s00, s01, s10, s11);
    if (!IsZero(s.final.w)) s.final.w = ScaleW(s.final.w, s00, s01, s10, s11);
  }
}
void AddWordInsPen(float pen, CompactLattice *clat) {
  for (CState &s : clat->st) for (CArc &a : s.arcs) if (a.label != 0) a.w.w.v1 = a.w.w.v1 + pen;
}

void ConvertLattice(const CompactLattice &clat, Lattice *lat, bool invert) {
  lat->st.clear();
  const int32_t n = (int32_t)clat.st.size();
  lat->st.resize(n);
  lat->start = clat.start;
  auto add = [&](int32_t from, int32_t il, int32_t ol, Weight w, int32_t to) {
    if (invert) std::swap(il, ol);
    lat->st[from].arcs.push_back(Arc{il, ol, w, to});
  };
  for (int32_t s = 0; s < n; s++) {
    const CWeight &fw = clat.st[s].final;
    if (!(IsZero(fw.w) && fw.str.empty())) {
      int32_t cur = s;
      for (size_t k = 0; k < fw.str.size(); k++) {
        const int32_t nx = (int32_t)lat->st.size();
        lat->st.emplace_back();
        add(cur, 0, fw.str[k], k == 0 ? fw.w : One(), nx);
        cur = nx;
      }
      lat->st[cur].final = fw.str.empty() ? fw.w : One();
    }
    for (const CArc &a : clat.st[s].arcs) {
      const size_t len = a.w.str.size();
      int32_t cur = s;
      for (size_t k = 0; k + 1 < len; k++) {
        const int32_t nx = (int32_t)lat->st.size();
        lat->st.emplace_back();
        add(cur, k == 0 ? a.label : 0, a.w.str[k], k == 0 ? a.w.w : One(), nx);
        cur = nx;
      }
      add(cur, len <= 1 ? a.label : 0, len > 0 ? a.w.str[len - 1] : 0, len <= 1 ? a.w.w : One(), a.next);
    }
  }
}

// Factor (fstext/factor-inl.h:64-140) + ConvertLattice (lattice-utils-inl.h:27-100): states with
// exactly one arc in and one arc out (not initial, not final, no word on the arc out) disappear
// into the arc that enters them; the transition-ids met on the way become the arc's string.
void ConvertLattice(const Lattice &lat, CompactLattice *clat) {
  clat->st.clear(); clat->start = -1;
  if (lat.start < 0) return;
  const int32_t n = (int32_t)lat.st.size();
  // depth-first discovery order (DfsOrderVisitor::InitState), OpenFst's visiting order
  std::vector<int32_t> order;
  {
    std::vector<uint8_t> color(n, 0);
    std::vector<std::pair<int32_t, size_t>> stack;
    for (int32_t root = lat.start; root < n;) {
      color[root] = 1; order.push_back(root);
      stack.emplace_back(root, 0);
      while (!stack.empty()) {
        const int32_t s = stack.back().first;
        const size_t ai = stack.back().second;
        if (ai >= lat.st[s].arcs.size()) { color[s] = 2; stack.pop_back(); if (!stack.empty()) stack.back().second++; continue; }
        const int32_t nx = lat.st[s].arcs[ai].next;
        if (color[nx] == 0) { color[nx] = 1; order.push_back(nx); stack.emplace_back(nx, 0); }
        else stack.back().second++;
      }
      for (root = (root == lat.start) ? 0 : root + 1; root < n && color[root] != 0; ++root) {}
    }
  }
  enum { kInitial = 1, kFinal = 2, kArcsIn = 4, kMultIn = 8, kArcsOut = 16, kMultOut = 32, kOlabelsOut = 64, kIlabelsOut = 128 };
  std::vector<int> props(n, 0);
  props[lat.start] |= kInitial;
  for (int32_t s = 0; s < n; s++) {
    for (const Arc &a : lat.st[s].arcs) {
      if (a.ilabel != 0) props[s] |= kIlabelsOut;
      if (a.olabel != 0) props[s] |= kOlabelsOut;
      if (props[s] & kArcsOut) props[s] |= kMultOut;
      props[s] |= kArcsOut;
      if (props[a.next] & kArcsIn) props[a.next] |= kMultIn;
      props[a.next] |= kArcsIn;
    }
    if (!IsZero(lat.st[s].final)) props[s] |= kFinal;
  }
  std::vector<uint8_t> remove(n);
  for (int32_t s = 0; s < n; s++) remove[s] = props[s] == (kArcsIn | kArcsOut) || props[s] == (kArcsIn | kArcsOut | kIlabelsOut);
  std::vector<int32_t> mapping(n, -1);
  CompactLattice f;     // arcs carry their strings directly (the symbol table of Factor is only an indirection)
  { size_t keep = 0; for (int32_t s = 0; s < n; s++) keep += !remove[s]; f.st.reserve(keep); }      // (no regrowth: a state moves its arcs' strings along)
  auto state_of = [&](int32_t s) { if (mapping[s] < 0) { mapping[s] = (int32_t)f.st.size(); f.st.emplace_back(); } return mapping[s]; };
  for (int32_t s : order) {
    if (remove[s]) continue;
    const int32_t ns = state_of(s);
    f.st[ns].arcs.reserve(lat.st[s].arcs.size());
    for (const Arc &a0 : lat.st[s].arcs) {
      Arc arc = a0;
      CArc out;
      if (arc.ilabel != 0) out.w.str.push_back(arc.ilabel);
      while (remove[arc.next]) {
        const Arc &nx = lat.st[arc.next].arcs[0];
        arc.w = Times(arc.w, nx.w);
        if (nx.ilabel != 0) out.w.str.push_back(nx.ilabel);
        arc.next = nx.next;
      }
      out.label = arc.olabel; out.w.w = arc.w; out.next = state_of(arc.next);
      f.st[ns].arcs.push_back(std::move(out));
    }
    if (!IsZero(lat.st[s].final)) f.st[ns].final.w = lat.st[s].final;
  }
  f.start = mapping[lat.start];
  TopSort(&f);
  *clat = std::move(f);
}

int32_t LatticeStateTimes(const Lattice &lat, std::vector<int32_t> *times) {
  if (!IsTopSorted(lat)) throw std::runtime_error("Input lattice must be topologically sorted.");
  const int32_t n = (int32_t)lat.st.size();
  times->assign(n, -1);
  if (n == 0) return 0;
  (*times)[0] = 0;
  for (int32_t s = 0; s < n; s++) {
    const int32_t t = (*times)[s];
    for (const Arc &a : lat.st[s].arcs) {
      const int32_t nt = a.ilabel != 0 ? t + 1 : t;
      if ((*times)[a.next] == -1) (*times)[a.next] = nt;
    }
  }
  return *std::max_element(times->begin(), times->end());
}

int32_t CompactLatticeStateTimes(const CompactLattice &clat, std::vector<int32_t> *times) {
  if (!IsTopSorted(clat)) throw std::runtime_error("Input lattice must be topologically sorted.");
  const int32_t n = (int32_t)clat.st.size();
  times->assign(n, -1);
  if (n == 0) return 0;
  (*times)[0] = 0;
  int32_t utt_len = -1;
  for (int32_t s = 0; s < n; s++) {
    const int32_t t = (*times)[s];
    for (const CArc &a : clat.st[s].arcs) {
      const int32_t nt = t + (int32_t)a.w.str.size();
      if ((*times)[a.next] == -1) (*times)[a.next] = nt;
    }
    if (!(IsZero(clat.st[s].final.w) && clat.st[s].final.str.empty())) {
      const int32_t len = t + (int32_t)clat.st[s].final.str.size();
      utt_len = utt_len == -1 ? len : std::max(utt_len, len);
    }
  }
  return utt_len == -1 ? 0 : utt_len;    // (no final state: the reference warns and returns 0)
}

static inline double LogAddD(double x, double y) {     // base/kaldi-math.h:186-205
  double diff;
  if (x < y) { diff = x - y; x = y; } else diff = y - x;
  if (diff >= -36.04365338911715) return x + std::log1p(std::exp(diff));     // kMinLogDiffDouble = log(DBL_EPSILON)
  return x;
}

double LatticeForwardBackward(const Lattice &lat, Posterior *post, double *acoustic_like_sum) {
  const double kLogZero = -std::numeric_limits<double>::infinity();
  if (acoustic_like_sum) *acoustic_like_sum = 0.0;
  std::vector<int32_t> times;
  const int32_t max_time = LatticeStateTimes(lat, &times);
  const int32_t n = (int32_t)lat.st.size();
  std::vector<double> alpha(n, kLogZero);
  std::vector<double> &beta = alpha;
  double tot_forward = kLogZero;
  post->assign(max_time, {});
  if (n == 0) return kLogZero;
  alpha[0] = 0.0;
  for (int32_t s = 0; s < n; s++) {
    const double a = alpha[s];
    for (const Arc &arc : lat.st[s].arcs) alpha[arc.next] = LogAddD(alpha[arc.next], a - Cost(arc.w));
    const Weight &f = lat.st[s].final;
    if (!IsZero(f)) {
      tot_forward = LogAddD(tot_forward, a - (double)(f.v1 + f.v2));
      if (times[s] != max_time) throw std::runtime_error("Lattice is inconsistent (final-prob not at max_time)");
    }
  }
  for (int32_t s = n - 1; s >= 0; s--) {
    const Weight &f = lat.st[s].final;
    double this_beta = -(double)(f.v1 + f.v2);
    for (const Arc &arc : lat.st[s].arcs) {
      const double arc_beta = beta[arc.next] - Cost(arc.w);
      this_beta = LogAddD(this_beta, arc_beta);
      if (arc.ilabel != 0 || acoustic_like_sum) {
        const double p = std::exp(alpha[s] + arc_beta - tot_forward);
        if (arc.ilabel != 0) (*post)[times[s]].push_back(std::make_pair(arc.ilabel, (float)p));
        if (acoustic_like_sum) *acoustic_like_sum -= p * arc.w.v2;
      }
    }
    if (acoustic_like_sum && !IsZero(f)) *acoustic_like_sum -= std::exp(alpha[s] - Cost(f) - tot_forward) * f.v2;
    beta[s] = this_beta;
  }
  const double tot_backward = beta[0];
  // MergePairVectorSumming (util/stl-utils.h:289-315): sort on the transition-id, add up equal ones, drop zeros
  for (auto &v : *post) {
    std::sort(v.begin(), v.end(), [](const std::pair<int32_t, float> &a, const std::pair<int32_t, float> &b) { return a.first < b.first; });
    size_t out = 0, in = 0;
    while (in < v.size()) {
      v[out] = v[in++];
      while (in < v.size() && v[in].first == v[out].first) v[out].second += v[in++].second;
      if (v[out].second != 0.0f) out++;
    }
    v.resize(out);
  }
  return tot_backward;
}

bool RescoreCompactLattice(const std::function<float(int32_t, int32_t)> &loglike, int32_t num_frames, CompactLattice *clat) {
  if (clat->st.empty()) return false;
  if (!IsTopSorted(*clat) && !TopSort(clat)) return false;
  std::vector<int32_t> times;
  const int32_t utt_len = CompactLatticeStateTimes(*clat, &times);
  // The reference buckets (state, arc, transition-id) by time and walks the times in order (lattice-functions.cc:1236-1300); a weight
  // receives its transition-ids' scores in increasing time, i.e. in the order of its own string, and no two weights share a sum: walking
  // the weights and their strings directly adds the same floats in the same order -- without three thousand vectors per utterance.
  // First the checks the bucketed walk would have failed on (nothing is modified when it returns false)
  if (utt_len > num_frames) return false;          // features are too short for the lattice
  for (int32_t s = 0; s < (int32_t)clat->st.size(); s++) {
    const int32_t t = times[s];
    for (const CArc &a : clat->st[s].arcs)
      if (!a.w.str.empty() && t >= utt_len && t != utt_len) return false;
  }
  for (int32_t s = 0; s < (int32_t)clat->st.size(); s++) {
    const int32_t t = times[s];
    for (CArc &a : clat->st[s].arcs) {
      if (t >= utt_len) continue;                  // (t == utt_len: the reference adds nothing for such a string either)
      for (size_t off = 0; off < a.w.str.size(); off++) a.w.w.v2 = -loglike(t + (int32_t)off, a.w.str[off]) + a.w.w.v2;
    }
    CWeight &f = clat->st[s].final;
    if (!(IsZero(f.w) && f.str.empty()))
      for (size_t off = 0; off < f.str.size(); off++) f.w.v2 = -loglike(t + (int32_t)off, f.str[off]) + f.w.v2;
  }
  return true;
}

void CompactLatticeShortestPath(const CompactLattice &clat_in, CompactLattice *shortest) {     // lattice-functions.cc:1094-1177
  const CompactLattice *pc = &clat_in;
  CompactLattice copy;
  if (!IsTopSorted(clat_in)) {
    copy = clat_in;
    if (!TopSort(&copy)) throw std::runtime_error("Was not able to topologically sort lattice (cycles found?)");
    pc = &copy;
  }
  const CompactLattice &clat = *pc;
  shortest->st.clear(); shortest->start = -1;
  if (clat.start < 0) return;
  const int32_t n = (int32_t)clat.st.size(), superfinal = n;
  const double inf = std::numeric_limits<double>::infinity();
  std::vector<std::pair<double, int32_t>> best(n + 1, std::make_pair(inf, -1));
  best[0].first = 0;
  for (int32_t s = 0; s < n; s++) {
    const double my = best[s].first;
    for (const CArc &a : clat.st[s].arcs) {
      const double nc = my + Cost(a.w.w);
      if (nc < best[a.next].first) { best[a.next].first = nc; best[a.next].second = s; }
    }
    const double tf = my + Cost(clat.st[s].final.w);
    if (tf < best[superfinal].first) { best[superfinal].first = tf; best[superfinal].second = s; }
  }
  std::vector<int32_t> states;
  int32_t cur = superfinal;
  while (cur != 0) {
    const int32_t prev = best[cur].second;
    if (prev < 0) return;       // no path: empty result
    states.push_back(prev);
    cur = prev;
  }
  std::reverse(states.begin(), states.end());
  shortest->st.resize(states.size());
  shortest->start = 0;
  for (size_t s = 0; s < states.size(); s++) {
    if (s + 1 < states.size()) {
      bool have = false;
      const CArc *pick = nullptr;
      for (const CArc &a : clat.st[states[s]].arcs)
        if (a.next == states[s + 1] && (!have || Cost(a.w.w) < Cost(pick->w.w))) { pick = &a; have = true; }
      CArc na = *pick;
      na.next = (int32_t)s + 1;
      shortest->st[s].arcs.push_back(std::move(na));
    } else {
      shortest->st[s].final = clat.st[states[s]].final;
    }
  }
}

bool GetLinearSymbolSequence(const CompactLattice &lin, std::vector<int32_t> *ali, std::vector<int32_t> *words, Weight *tot) {
  // CompactLatticeToWordAlignment / GetLinearSymbolSequence on the converted lattice: strings in
  // path order, words where the arc label is not 0, weights multiplied along the path
  if (ali) ali->clear();
  if (words) words->clear();
  Weight w = One();
  if (lin.start < 0) { if (tot) *tot = Zero(); return true; }   // the reference accepts the empty FST as the empty sequence
  int32_t s = lin.start;
  for (;;) {
    const CState &st = lin.st[s];
    if (!IsZero(st.final.w) || !st.final.str.empty()) {
      if (!st.arcs.empty()) return false;
      w = Times(w, st.final.w);
      if (ali) ali->insert(ali->end(), st.final.str.begin(), st.final.str.end());
      break;
    }
    if (st.arcs.size() != 1) return false;
    const CArc &a = st.arcs[0];
    w = Times(w, a.w.w);
    if (ali) ali->insert(ali->end(), a.w.str.begin(), a.w.str.end());
    if (words && a.label != 0) words->push_back(a.label);
    s = a.next;
  }
  if (tot) *tot = w;
  return true;
}

}  // namespace vblat
