// vbh_faster.hip -- FasterDecoder (1-best Viterbi beam search with back pointers) on gfx950.
//
// Replaces (paths relative to /root/reference/kaldi-master/src/):
//   decoder/faster-decoder.cc:36-54    InitDecoding / Decode (frame loop)
//   :151-213  GetCutoff (double costs, nth_element for max-/min-active, adaptive beam)
//   :224-303  ProcessEmitting          :306-346  ProcessNonemitting
//   :78-147   ReachedFinal / GetBestPath (back-pointer trace, (graph, acoustic) totals)
//   decoder/faster-decoder.h:116-133   Token: cost = prev->cost + arc.weight [+ ac_cost], in double
//
// Same execution model as vbh_decode.hip (one 1024-thread workgroup per utterance, persistent
// grid, per-frame state map in LDS, arcs spread over lanes by wave_expand); no lattice, so no
// links, pruning passes or compaction: a token is {previous token, arc, state} + a double cost,
// kept for the whole utterance for the back trace.
//
// Semantics = oracle/vb_oracle_decode.c's order-independent FasterDecoder (bit-exact, tests/test_gpu_faster.py; both modes reproduce
// the reference's best paths on all its golden cases, tests/test_oracle_decode.py): per state the cheapest incoming arc wins and ties go
// to the smaller arc index; the epsilon closure runs to its fixed point before tokens are made; ties between final tokens go to the
// smaller state id (the reference's own choices in those three places follow its hash-list order).  ADMISSION of emitting arcs
// (vbh_faster_cfg::admission; the kernel is a template over it, as decode_kernel is):
//   FADM_ORDERED (the default; oracle exact_cutoff = 3): the reference's rule -- an arc is admitted iff new_weight < next_weight_cutoff
//     as it stands when the arc is visited, every admitted arc tightening it (faster-decoder.cc:266-273), from the bound the best token's
//     arcs give (:241-255) -- applied in a canonical order: arcs leaving tokens whose state hashes to a lower one of 64 buckets count as
//     visited earlier (the prefix minimum of vbh_decode.hip's ordered admission, here on 64-bit keys);
//   FADM_EXACT (rounds 1-4; oracle exact_cutoff = 1): next_cutoff takes its final value (min over all candidate arcs of new_weight +
//     adaptive_beam) before anything is admitted -- the order-independent subset of what the reference admits.
#include "vbh_internal.h"

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstring>

namespace {

constexpr int NT = 1024;
constexpr int NW = NT / 64;
constexpr int EPSBIT = (int)0x80000000;     // on an arc's next state: that state has epsilon arcs
constexpr int STMASK = 0x3FFFFFFF;
constexpr int HBITS = 13;
constexpr int HCAP = 1 << HBITS;            // states per frame the LDS map can hold
#ifndef VBH_FASTER_HPROBE
#define VBH_FASTER_HPROBE 16      // (round 5, same box: 64 / 32 / 16 / 8 probes: 2.06 / 1.99 / 1.97 / 2.00 s per 1024 x 30 s: a full map is given up on sooner)
#endif
constexpr int HPROBE = VBH_FASTER_HPROBE;
constexpr unsigned HEMPTY = 0xFFFFFFFFu;
constexpr int MAX_EPS_ITERS = 4096;
#ifndef VBH_FASTER_UA
#define VBH_FASTER_UA 4         // x 64 arcs a wave has in flight in pass 1 (round 5, same box, s per 1024 x 30 s: 1 / 2 / 4: 1.68 / 1.65 / 1.56)
#endif
constexpr int UA = VBH_FASTER_UA;
static_assert(UA >= 1 && UA <= 4, "Shared::wx holds 64 x 4 marker bytes per wave");
#ifndef VBH_FASTER_SLOW
#define VBH_FASTER_SLOW 0      // 1: every frame takes the slow winner path (rounds 1-4's; A/B and tests)
#endif

typedef int i32x4 __attribute__((ext_vector_type(4)));
#define VBH_GLOBAL __attribute__((address_space(1)))
template <class T> using gp = VBH_GLOBAL T *;

__device__ __forceinline__ float asf(int x) { return __int_as_float(x); }

// The packed graph of vbh_decode.hip (DevGraph): st and arcs are ONE array of 16-byte records, a state's record {the graph's own state id,
// #epsilon arcs, #arcs, final cost} directly in front of its arcs (epsilon arcs first), a state's id = the index of its record, an arc's
// next state = that state's record.  Ids are monotone in the graph's own, arc positions in the plain arc numbering: ties fall as before.
struct FGraph { int S, start; const i32x4 *st; const i32x4 *arcs; const int *arc_il; };
struct FCfg { float beam, beam_delta, acwt; int max_active, min_active; };

struct FSlot {
  i32x4 *tok;              // [tok_cap] {prev token, arc (-1: start), state | EPSBIT, the arc's cost: float(cost - prev's cost)}
  double *cost;            // [2 x frame_cap] the costs of the current frame's tokens and of the frame being made, in alternating halves
                           //   (only those two frames' costs are ever read: the history keeps {previous token, arc, state} alone, round 5)
  unsigned long long *win; // [HCAP + g_cap] winner key (arc << 32 | source token) per state of the frame being built, by map slot (LDS slots, then
                           // the second level's); ~0 = none: make_tokens leaves every entry it read that way.  Used by the SLOW path only (below)
  // Round 5: the way into a state is recorded WHEN its cost is lowered.  Whoever lowers a state's cost (map_relax: improved) stores {its cost key,
  // arc, source} here, by map slot; the last lowering of a frame is the state's final cost, so its record is the winner -- unless (a) two
  // lowerings' stores landed in the other order (the record's key then differs from the state's final key: seen by number_tokens), or (b) another
  // arc reaches the state at exactly its final cost (ties go to the smaller arc index; map_relax sees the equal cost).  A frame in which either
  // happens takes the slow path -- the passes of rounds 1-4 that re-read every candidate against the final costs (winners, epsilon_winners) --
  // all others (nearly all) skip them: no second pass over the candidates, no 64-bit atomics in HBM, no gather of winner keys.
  // source: >= 0 a token of the previous frame; -1 none (the start token); <= -2: the state at map entry -2 - source of THIS frame (epsilon arcs)
  i32x4 *wrec;             // [HCAP + g_cap] {key lo, key hi, arc, source}
  double *c_w;             // [cand_cap] candidate new_weight
  i32x4 *c_r;              // [cand_cap] {arc, source token, next state | EPSBIT, 0}
  int *front[3];           // epsilon work lists (states), rotated
  int *elist;              // [frame_cap] the frame's states that have epsilon arcs, each once (pushed by its creator)
  unsigned *g_hs; unsigned long long *g_hc; int *g_ht;   // second level of the state map, [g_cap] (power of two)
  int *g_list;             // [frame_cap] the second-level slots the frame being built has claimed (what is walked, numbered and cleared)
  int *remap;              // [tok_cap] garbage collection: mark, then new index
  i32x4 *stage_tok;        // [stage_cap] survivors on their way down
  int tok_cap, frame_cap, cand_cap, front_cap, g_cap, stage_cap;
};

struct FUtt { long long row0; int T; int start; };      // start: the utterance's own start state (alignment: one graph per utterance inside a union graph), -1: the graph's
struct FResult { int ok, error, reached_final, num_frames, path_len, final_state; float final_cost; int pad; long long n_tokens; };

// (FERR_TOK: even after garbage collection the history does not fit)
enum { FERR_NONE = 0, FERR_TOK = 1, FERR_FRAME = 2, FERR_CAND = 3, FERR_FRONT = 4, FERR_MAP = 5, FERR_EPS = 6, FERR_PATH = 7 };

// order-preserving map double -> uint64
__device__ __forceinline__ unsigned long long d2o(double d) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(d);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double o2d(unsigned long long o) {
  const unsigned long long u = (o >> 63) ? (o & 0x7FFFFFFFFFFFFFFFull) : ~o;
  return __longlong_as_double((long long)u);
}
template <class T>
__device__ __forceinline__ gp<T> uptr(T *p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (gp<T>)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ int wave_alloc(int *counter, bool want) {
  const unsigned long long m = __ballot(want);
  if (m == 0) return -1;
  const int lane = threadIdx.x & 63;
  const int leader = __ffsll((long long)m) - 1;
  int base = 0;
  if (lane == leader) base = atomicAdd(counter, __popcll(m));
  base = __builtin_amdgcn_readlane(base, leader);
  return want ? base + __popcll(m & ((1ull << lane) - 1ull)) : -1;
}
__device__ __forceinline__ int wave_incl_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
  return v;
}
// reductions and a running maximum over the wave on the VALU (the DPP ladder of the scan above), see vbh_decode.hip
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
  const int id = -1;
#define VBH_DPP_MIN(ctrl, rows) { const unsigned o_ = (unsigned)__builtin_amdgcn_update_dpp(id, (int)v, ctrl, rows, 0xf, false); v = o_ < v ? o_ : v; }
  VBH_DPP_MIN(0x111, 0xf) VBH_DPP_MIN(0x112, 0xf) VBH_DPP_MIN(0x114, 0xf) VBH_DPP_MIN(0x118, 0xf) VBH_DPP_MIN(0x142, 0xa) VBH_DPP_MIN(0x143, 0xc)
#undef VBH_DPP_MIN
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) { return (unsigned)__builtin_amdgcn_readlane(wave_incl_scan((int)v), 63); }
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
  const unsigned hi = (unsigned)(v >> 32), lo = (unsigned)v;
  const unsigned mh = wave_min_u32(hi);
  const unsigned ml = wave_min_u32(hi == mh ? lo : 0xFFFFFFFFu);
  return ((unsigned long long)mh << 32) | ml;
}
// inclusive running minimum of 64-bit keys over the lanes (the DPP ladder of wave_min_u32 on both halves; lanes a step does not reach
// keep their value: the identity is all ones)
__device__ __forceinline__ unsigned long long wave_min_scan_u64(unsigned long long v) {
  unsigned hi = (unsigned)(v >> 32), lo = (unsigned)v;
  const int id = -1;
#define VBH_DPP_MIN64(ctrl, rows) { const unsigned oh_ = (unsigned)__builtin_amdgcn_update_dpp(id, (int)hi, ctrl, rows, 0xf, false), \
                                                   ol_ = (unsigned)__builtin_amdgcn_update_dpp(id, (int)lo, ctrl, rows, 0xf, false); \
                                    const bool lt_ = oh_ < hi || (oh_ == hi && ol_ < lo); hi = lt_ ? oh_ : hi; lo = lt_ ? ol_ : lo; }
  VBH_DPP_MIN64(0x111, 0xf) VBH_DPP_MIN64(0x112, 0xf) VBH_DPP_MIN64(0x114, 0xf) VBH_DPP_MIN64(0x118, 0xf) VBH_DPP_MIN64(0x142, 0xa) VBH_DPP_MIN64(0x143, 0xc)
#undef VBH_DPP_MIN64
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long shfl_u64(unsigned long long v, int src) {
  return ((unsigned long long)(unsigned)__shfl((int)(unsigned)(v >> 32), src, 64) << 32) | (unsigned)__shfl((int)(unsigned)v, src, 64);
}
enum { FADM_ORDERED = 0, FADM_EXACT = 1 };
// ordered admission: bucket of a token's state, from the graph's own state id (oracle/vb_oracle_decode.c VBO_FADM_BUCKET, adm_bucket_bits = 6)
__device__ __forceinline__ int adm_bucket(int st) { return (int)(((unsigned)st * 2654435761u) >> 26); }
__device__ __forceinline__ int wave_max_scan(int v) {      // inclusive running maximum over the lanes, values >= 0
#define VBH_DPP_MAX(ctrl, rows) { const int o_ = __builtin_amdgcn_update_dpp(0, v, ctrl, rows, 0xf, false); v = o_ > v ? o_ : v; }
  VBH_DPP_MAX(0x111, 0xf) VBH_DPP_MAX(0x112, 0xf) VBH_DPP_MAX(0x114, 0xf) VBH_DPP_MAX(0x118, 0xf) VBH_DPP_MAX(0x142, 0xa) VBH_DPP_MAX(0x143, 0xc)
#undef VBH_DPP_MAX
  return v;
}
// see vbh_decode.hip: sub-items of up to 64 items dealt out evenly over the lanes of a wave.  The owner of sub-item k is the last
// lane whose item starts at or before k: every such lane leaves its number at that position in the wave's strip of 64 LDS bytes
// (`mk`), and a running maximum over the window, plus what is carried in from before it, is the owner (no search by shuffles).
template <class F>
__device__ __forceinline__ void wave_expand(unsigned char *mk, int cnt, F body) {
  const int lane = threadIdx.x & 63;
  const int incl = wave_incl_scan(cnt);
  const int total = __builtin_amdgcn_readlane(incl, 63);
  const int excl = incl - cnt;
  const bool has = cnt > 0;
  for (int k0 = 0; k0 < total; k0 += 64) {
    if (lane < 16) reinterpret_cast<unsigned *>(mk)[lane] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (has && excl >= k0 && excl < k0 + 64) mk[excl - k0] = (unsigned char)(lane + 1);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const unsigned long long before = __ballot(has && excl < k0);
    int c_own = 0, c_start = 0;
    if (before) {
      const int l = __builtin_amdgcn_readfirstlane(63 - __clzll((long long)before));
      c_own = l + 1;
      c_start = __builtin_amdgcn_readlane(excl, l) + 1;
    }
    const int k = k0 + lane;
    const int m = (int)mk[lane];
    int so = wave_max_scan(m), ss = wave_max_scan(m ? k + 1 : 0);
    so = so > c_own ? so : c_own; ss = ss > c_start ? ss : c_start;
    const bool valid = k < total;
    const int owner = valid ? so - 1 : 0;
    const int j = k - (ss - 1);
    body(valid, owner, j);
  }
}
// the same UA x 64 sub-items per round (vbh_decode.hip wave_owners / wave_expand_n): body(valid[UA], owner[UA], j[UA]) issues the loads
// of all of them before it uses any.  mk: 64 x UA marker bytes.
template <int UA, class F>
__device__ __forceinline__ void wave_expand_n(unsigned char *mk, int cnt, F body) {
  const int lane = threadIdx.x & 63;
  const int incl = wave_incl_scan(cnt);
  const int total = __builtin_amdgcn_readlane(incl, 63);
  const int excl = incl - cnt;
  const bool has = cnt > 0;
  for (int k0 = 0; k0 < total; k0 += 64 * UA) {
    if (lane < 16 * UA) reinterpret_cast<unsigned *>(mk)[lane] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (has && excl >= k0 && excl < k0 + 64 * UA) mk[excl - k0] = (unsigned char)(lane + 1);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const unsigned long long before = __ballot(has && excl < k0);
    int c_own = 0, c_start = 0;
    if (before) {
      const int l = __builtin_amdgcn_readfirstlane(63 - __clzll((long long)before));
      c_own = l + 1;
      c_start = __builtin_amdgcn_readlane(excl, l) + 1;
    }
    bool valid[UA]; int owner[UA], j[UA], m[UA];
#pragma unroll
    for (int u = 0; u < UA; u++) m[u] = (k0 + u * 64 < total) ? (int)mk[u * 64 + lane] : 0;
#pragma unroll
    for (int u = 0; u < UA; u++) {
      valid[u] = false; owner[u] = 0; j[u] = 0;
      if (k0 + u * 64 < total) {      // (uniform)
        const int k = k0 + u * 64 + lane;
        int so = wave_max_scan(m[u]), ss = wave_max_scan(m[u] ? k + 1 : 0);
        so = so > c_own ? so : c_own; ss = ss > c_start ? ss : c_start;
        valid[u] = k < total;
        owner[u] = valid[u] ? so - 1 : 0;
        j[u] = k - (ss - 1);
        c_own = __builtin_amdgcn_readlane(so, 63); c_start = __builtin_amdgcn_readlane(ss, 63);
      }
    }
    body(valid, owner, j);
  }
}
// N requests per lane with ONE LDS atomic; all 64 lanes must call it together
template <int N>
__device__ __forceinline__ void wave_alloc_n(int *counter, const bool (&want)[N], int (&idx)[N]) {
  const int lane = threadIdx.x & 63;
  const unsigned long long lt = (1ull << lane) - 1ull;
  unsigned long long m[N]; int off[N]; int total = 0;
#pragma unroll
  for (int u = 0; u < N; u++) { m[u] = __ballot(want[u]); off[u] = total; total += __popcll(m[u]); }
  int base = 0;
  if (total) {
    if (lane == 0) base = atomicAdd(counter, total);
    base = __builtin_amdgcn_readfirstlane(base);
  }
#pragma unroll
  for (int u = 0; u < N; u++) idx[u] = want[u] ? base + off[u] + __popcll(m[u] & lt) : -1;
}
__device__ __forceinline__ int wave_chunk(int n) {
  const int c = (n + NW - 1) / NW;
  return c < 1 ? 1 : (c > 64 ? 64 : c);
}

struct Shared {
  unsigned long long red[2][NW], red2[2][NW];
  unsigned red3[2][NW];            // get_cutoff: the state of a wave's cheapest token
  // ordered admission (FADM_ORDERED): bmin[b] = key of the smallest new_weight + adaptive_beam over the frame's arcs that leave a token of
  // bucket b and got below b's own bound (others cannot change a later bucket's), init_bound = the same over the best token's arcs
  unsigned long long bmin[64], init_bound;
  unsigned hist[256];
  unsigned vhist[1024];            // get_cutoff: the frame's costs by value, 1024 equal steps from the smallest to the largest
  unsigned long long vlist[2][64]; // ... the costs of the step an order statistic falls in ([which]: the max-active, the min-active one --
  unsigned long long vmin[2], vmax[2];   // ... the range of the step's costs beyond the list's 64 (reset by their reader; ~0 / 0 between uses)
  int vb[2], vr[2], vn[2];         //     a slow wave may still be reading the first one's when the second one starts), that step, the
                                   //     statistic's rank within it, the fill of vlist
  alignas(4) unsigned char wx[NW][64 * 4];   // wave_expand / wave_expand_n: a strip of 64 x UA marker bytes per wave (UA <= 4)
  unsigned hs[HCAP];               // state map of the frame being built: tag,
  unsigned long long hc[HCAP];     // best cost (order-preserving key),
  int ht[HCAP];                    // index of the state's token within the frame
  int n_new, n_cand, fcnt[3], error, utt;
  unsigned long long cand_bound;   // pass 1: key of the smallest new_weight + adaptive_beam seen so far (>= the final next_cutoff)
  int n_eps;                       // fill of FSlot::elist
  int grp[(HCAP / NT) * NW + 1];   // number_tokens: occupied slots per group of 64 map slots, then their exclusive prefix
  int u_tie, u_bad;                // (diagnostic: frames of the utterance with the flag set)
  int tie, bad, n_slow;            // FSlot::wrec: an equal cost was seen / a record does not carry its state's final key (this frame); frames that took the slow path (this utterance)
  int g_used;                      // the second level holds entries of the frame being built
  int n_g;                         // ... how many (fill of FSlot::g_list); -1: unknown, the whole table is cleared
  unsigned long long sel_prefix, sel_mask; int sel_k, sel_single;
  unsigned long long u0;
#ifdef VBH_FASTER_PROF      // build-time diagnostic: shader-clock cycles per phase, printed by thread 0 for the first utterances
  long long tph[16], t_last;
  int n_sel, n_sel_over, n_sel_radix;
  long long s_cand, s_new, s_g, s_tok; int s_gframes;
#endif
};
#ifdef VBH_FASTER_PROF
#define FPHASE(i) { if (threadIdx.x == 0) { const long long now_ = (long long)__builtin_amdgcn_s_memtime(); sh.tph[i] += now_ - sh.t_last; sh.t_last = now_; } }
#else
#define FPHASE(i)
#endif

__device__ __forceinline__ unsigned long long block_min_u64(unsigned long long v, Shared &sh, int &par) {
  v = wave_min_u64(v);
  if ((threadIdx.x & 63) == 0) sh.red[par][threadIdx.x >> 6] = v;
  __syncthreads();
  unsigned long long r = sh.red[par][0];
#pragma unroll
  for (int i = 1; i < NW; i++) r = sh.red[par][i] < r ? sh.red[par][i] : r;
  par ^= 1;
  return r;
}
// smallest and largest of the threads' (lo, hi) in one barrier; st: the thread's state at its lo -> the smallest state among the
// tokens at the overall lo (GetCutoff's best_elem, faster-decoder.cc:161-166, with ties to the smaller state id instead of list order)
__device__ __forceinline__ void block_minmax_u64(unsigned long long &lo, unsigned long long &hi, unsigned &st, Shared &sh, int &par) {
  const unsigned long long mylo = lo;
  lo = wave_min_u64(lo); hi = ~wave_min_u64(~hi);
  st = wave_min_u32(mylo == lo ? st : 0xFFFFFFFFu);
  if ((threadIdx.x & 63) == 0) { sh.red[par][threadIdx.x >> 6] = lo; sh.red2[par][threadIdx.x >> 6] = hi; sh.red3[par][threadIdx.x >> 6] = st; }
  __syncthreads();
  lo = sh.red[par][0]; hi = sh.red2[par][0]; st = sh.red3[par][0];
#pragma unroll
  for (int i = 1; i < NW; i++) {
    const unsigned long long l_ = sh.red[par][i]; const unsigned s_ = sh.red3[par][i];
    if (l_ < lo || (l_ == lo && s_ < st)) { lo = l_; st = s_; }
    hi = sh.red2[par][i] > hi ? sh.red2[par][i] : hi;
  }
  par ^= 1;
}
__device__ __forceinline__ unsigned long long block_sum_u64(unsigned long long v, Shared &sh, int &par) {
  v = ((unsigned long long)wave_sum_u32((unsigned)(v >> 32)) << 32) | wave_sum_u32((unsigned)v);      // (both halves are counts of a frame's tokens)
  if ((threadIdx.x & 63) == 0) sh.red[par][threadIdx.x >> 6] = v;
  __syncthreads();
  unsigned long long r = 0;
#pragma unroll
  for (int i = 0; i < NW; i++) r += sh.red[par][i];
  par ^= 1;
  return r;
}

// the step of a cost in the frame's histogram by value (monotone in c; c >= best)
__device__ __forceinline__ int step_of(double c, double best, double scale) {
  const double x = (c - best) * scale;
  return x >= 1023.0 ? 1023 : (int)x;      // (also the costs at infinity; a NaN goes to step 0 like the comparison says)
}

// k-th smallest (0-based) of a frame's costs: radix select over the 64-bit order-preserving keys (the std::nth_element value in
// GetCutoff, faster-decoder.cc:171-196).  each(f) calls f(key) for the calling thread's share of the keys.  The keys of a frame
// agree in their leading bytes (sign, exponent, the top of the mantissa: they lie within a beam of each other): lo and hi, the
// smallest and the largest key, say in how many, and the select starts below them.
template <class Each>
__device__ double select_kth(Each each, unsigned long long lo, unsigned long long hi, int k, Shared &sh) {
  const int tid = threadIdx.x;
  int shift0 = 56;
  while (shift0 > 0 && (lo >> shift0) == (hi >> shift0)) shift0 -= 8;
  __syncthreads();
  if (tid == 0) {
    const unsigned long long m0 = shift0 == 56 ? 0ull : ~0ull << (shift0 + 8);
    sh.sel_prefix = lo & m0; sh.sel_mask = m0; sh.sel_k = k;
  }
  for (int shift = shift0; shift >= 0; shift -= 8) {
    for (int i = tid; i < 256; i += NT) sh.hist[i] = 0;
    __syncthreads();
    const unsigned long long prefix = sh.sel_prefix, mask = sh.sel_mask;
    each([&](unsigned long long key) {
      if ((key & mask) == prefix) atomicAdd(&sh.hist[(unsigned)(key >> shift) & 255u], 1u);
    });
    __syncthreads();
    if (tid < 64) {
      const int kk = sh.sel_k;
      const int h0 = (int)sh.hist[4 * tid], h1 = (int)sh.hist[4 * tid + 1], h2 = (int)sh.hist[4 * tid + 2], h3 = (int)sh.hist[4 * tid + 3];
      const int mine = h0 + h1 + h2 + h3;
      const int incl = wave_incl_scan(mine);
      const unsigned long long m = __ballot(incl > kk);
      const int owner = m ? __ffsll((long long)m) - 1 : 63;
      if (tid == owner) {
        int r = kk - (incl - mine), b = 0, hb = h0;
        if (r >= h0) { r -= h0; b = 1; hb = h1; if (r >= h1) { r -= h1; b = 2; hb = h2; if (r >= h2) { r -= h2; b = 3; hb = h3; } } }
        sh.sel_k = r;
        sh.sel_prefix = prefix | ((unsigned long long)(4 * tid + b) << shift);
        sh.sel_mask = mask | (255ull << shift);
        sh.sel_single = hb == 1;      // one key left under the prefix: it is the answer, whatever its remaining bytes
      }
    }
    __syncthreads();
    if (sh.sel_single && shift > 0) {
      const unsigned long long prefix1 = sh.sel_prefix, mask1 = sh.sel_mask;
      __syncthreads();
      each([&](unsigned long long key) { if ((key & mask1) == prefix1) sh.sel_prefix = key; });
      __syncthreads();
      break;
    }
  }
  return o2d(sh.sel_prefix);
}

// The k-th smallest cost of a frame (0-based), given its histogram by value (vhist: step = (int)((cost - best) * scale), a
// monotone function of the cost, so the steps are ordered like the costs): the step the statistic falls in, then the few costs of
// that step ranked against each other by every wave for itself.  Two barriers; more than 64 costs in the step (many equal ones, a
// degenerate range): the radix select.
template <class Each>
__device__ double select_by_value(Each each, double best, double scale, unsigned long long lo, unsigned long long hi, int k, int which, Shared &sh) {
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid < 64) {
    int c[16], mine = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) { c[i] = (int)sh.vhist[16 * tid + i]; mine += c[i]; }
    const int incl = wave_incl_scan(mine);
    const unsigned long long m = __ballot(incl > k);
    const int owner = m ? __ffsll((long long)m) - 1 : 63;
    if (tid == owner) {
      int r = k - (incl - mine), bb = 0;
#pragma unroll
      for (int i = 0; i < 15; i++) if (bb == i && r >= c[i]) { r -= c[i]; bb = i + 1; }
      sh.vb[which] = 16 * tid + bb; sh.vr[which] = r; sh.vn[which] = 0;
    }
  }
  __syncthreads();
  const int vb = sh.vb[which], vr = sh.vr[which];
  each([&](unsigned long long key) {
    if (step_of(o2d(key), best, scale) == vb) {
      const int q = atomicAdd(&sh.vn[which], 1);
      if (q < 64) sh.vlist[which][q] = key;
      else { atomicMin(&sh.vmin[which], key); atomicMax(&sh.vmax[which], key); }      // (the overflow's range: see below)
    }
  });
  __syncthreads();
  const int vn = sh.vn[which];
#ifdef VBH_FASTER_PROF
  if (threadIdx.x == 0) { sh.n_sel++; if (vn > 64) sh.n_sel_over++; }
#endif
  if (vn > 64) {
    // More costs in the step than the list holds.  The common reason (round 5): dozens of EXACTLY equal costs -- a word end's fan-out over
    // the phone-start states by weight-0 epsilon arcs -- right where min-active puts the cutoff.  If the step's costs are all one value,
    // that value is the statistic; only a step of many DIFFERENT costs needs the radix select (8 passes of 3 barriers on 64-bit keys).
    unsigned long long smin = sh.vmin[which], smax = sh.vmax[which];
    for (int j = 0; j < 64; j++) { const unsigned long long o = sh.vlist[which][j]; smin = o < smin ? o : smin; smax = o > smax ? o : smax; }
    __syncthreads();      // (everybody has read the range before the next use of the slot resets it)
    if (threadIdx.x == 0) { sh.vmin[which] = ~0ull; sh.vmax[which] = 0ull; }
    if (smin == smax) return o2d(smin);
#ifdef VBH_FASTER_PROF
    if (threadIdx.x == 0) sh.n_sel_radix++;
#endif
    return select_kth(each, lo, hi, k, sh);
  }
  const unsigned long long mykey = lane < vn ? sh.vlist[which][lane] : ~0ull;
  int rank = 0;
  for (int j = 0; j < vn; j++) { const unsigned long long o = sh.vlist[which][j]; rank += (o < mykey) || (o == mykey && j < lane); }
  const unsigned long long hit = __ballot(lane < vn && rank == vr);
  const int src = hit ? __ffsll((long long)hit) - 1 : 0;
  const unsigned klo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)mykey, src), khi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(mykey >> 32), src);
  return o2d(((unsigned long long)khi << 32) | klo);
}

// GetCutoff (faster-decoder.cc:151-213), in double, for the frame whose n costs each() enumerates: the cutoff on the tokens
// (weight_cutoff) and the beam for the arcs leaving them (adaptive_beam).  Two barriers (smallest + largest cost; the counts on
// either side of the beam + the histogram by value), two more per order statistic that decides.
template <class EachS, class Each>
__device__ void get_cutoff(EachS each_s, Each each, int n, const FCfg &cfg, Shared &sh, int &par, double &weight_cutoff, float &adaptive_beam, int &best_state) {
  unsigned long long bk = ~0ull, wk = 0ull;
  unsigned bs = 0xFFFFFFFFu;
  FPHASE(11)
  each_s([&](unsigned long long k, unsigned st) { if (k < bk || (k == bk && st < bs)) { bk = k; bs = st; } wk = k > wk ? k : wk; });
  sh.vhist[threadIdx.x] = 0;
  static_assert(NT == 1024, "one histogram step per thread");
  block_minmax_u64(bk, wk, bs, sh, par);
  best_state = (int)bs;
  FPHASE(12)
  const double best = n > 0 ? o2d(bk) : (double)INFINITY;
  if (cfg.max_active == INT_MAX && cfg.min_active == 0) { adaptive_beam = cfg.beam; weight_cutoff = best + cfg.beam; return; }
  const double beam_cutoff = best + cfg.beam;
  double mac = (double)INFINITY, mic = (double)INFINITY;
  // the order statistics are only computed when they decide the comparisons below:
  // tmp[max_active] < beam_cutoff  <=>  more than max_active costs lie below beam_cutoff;
  // tmp[min_active] > beam_cutoff  <=>  at most min_active costs are <= beam_cutoff
  const double range = n > 0 ? o2d(wk) - best : 0.0;
  const double scale = range > 0.0 && range < (double)INFINITY ? 1023.0 / range : 0.0;
  // The reference collects the costs in a std::vector<BaseFloat> (faster-decoder.h:174, faster-decoder.cc:171): both order statistics are
  // FLOAT32-ROUNDED costs, compared with the exact double beam_cutoff and handed back as the cutoff (:183, :196).  Rounding is monotone, so
  // the k-th smallest rounded cost is the rounded k-th smallest cost -- the selects below run on the exact keys and their result is rounded
  // -- and the counts that decide whether a statistic matters are taken over the rounded costs.  (Where the cutoff falls among equal costs
  // -- a word end's fan-out over the phone-start states leaves dozens, and min-active = 20 puts the cutoff there -- the rounding decides
  // whether `cost < cutoff` expands all of them or none; with exact doubles the search lost the path on most of the 512 utterances of
  // tests/golden/speech_faster512.npz: round 5.)
  unsigned long long cnt = 0;   // (#rounded cost < beam_cutoff) << 32 | (#rounded cost <= beam_cutoff)
  each([&](unsigned long long k) {
    const double c = o2d(k);
    const double cr = (double)(float)c;
    cnt += ((unsigned long long)(cr < beam_cutoff) << 32) | (unsigned long long)(cr <= beam_cutoff);
    atomicAdd(&sh.vhist[step_of(c, best, scale)], 1u);
  });
  cnt = block_sum_u64(cnt, sh, par);
  FPHASE(13)
  const int cnt_lt = (int)(cnt >> 32), cnt_le = (int)(unsigned)cnt;
  if (n > cfg.max_active && cnt_lt > cfg.max_active) mac = (double)(float)select_by_value(each, best, scale, bk, wk, cfg.max_active, 0, sh);
  if (mac < beam_cutoff) { adaptive_beam = (float)(mac - best + cfg.beam_delta); weight_cutoff = mac; return; }
  if (n > cfg.min_active) {
    if (cfg.min_active == 0) mic = best;
    else if (cnt_le <= cfg.min_active) mic = (double)(float)select_by_value(each, best, scale, bk, wk, cfg.min_active, 1, sh);
    else mic = beam_cutoff;     // some value <= beam_cutoff: the comparison below is false
  }
  if (mic > beam_cutoff) { adaptive_beam = (float)(mic - best + cfg.beam_delta); weight_cutoff = mic; }
  else { adaptive_beam = cfg.beam; weight_cutoff = beam_cutoff; }
}

// ---- state map of the frame being built: state -> (best cost, index of its token in the frame).
// First level: open addressing in LDS (tag claimed with atomicCAS, cost lowered with a 64-bit
// atomicMin on the order-preserving key).  A state whose HPROBE probe slots are all taken goes to the
// second level, the same table in HBM (g_cap slots, cleared only after frames that used it); slots
// are never freed within a frame, so a state always resolves to the same level.  A slot handle is
// >= 0 for LDS, <= -2 for HBM slot -(handle + 2), -1 for "no room".
struct Map2 { gp<unsigned> gs; gp<unsigned long long> gc; gp<int> gt; gp<int> gl; int gmask, gl_cap; };
__device__ __forceinline__ Map2 map2_of(const FSlot &s) { Map2 m; m.gs = uptr(s.g_hs); m.gc = uptr(s.g_hc); m.gt = uptr(s.g_ht); m.gl = uptr(s.g_list); m.gmask = s.g_cap - 1; m.gl_cap = s.frame_cap; return m; }
__device__ __forceinline__ unsigned map_hash(int st) { return ((unsigned)st * 2654435761u) >> (32 - HBITS); }
__device__ __forceinline__ unsigned map_hash2(int st, int gmask) { return (((unsigned)st * 2246822519u) >> 7) & (unsigned)gmask; }
__device__ __forceinline__ int map_find(const Shared &sh, const Map2 &m, int st) {
  unsigned h = map_hash(st);
#pragma unroll 1
  for (int p = 0; p < HPROBE; p++, h = (h + 1) & (HCAP - 1)) {
    const unsigned cur = sh.hs[h];
    if (cur == (unsigned)st) return (int)h;
    if (cur == HEMPTY) return -1;
  }
  if (!sh.g_used) return -1;
  h = map_hash2(st, m.gmask);
#pragma unroll 1
  for (int p = 0; p <= m.gmask; p++, h = (h + 1) & (unsigned)m.gmask) {
    const unsigned cur = __hip_atomic_load(&m.gs[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == (unsigned)st) return -(int)h - 2;
    if (cur == HEMPTY) return -1;
  }
  return -1;
}
__device__ __forceinline__ unsigned long long map_key(const Shared &sh, const Map2 &m, int slot) {
  return slot >= 0 ? sh.hc[slot] : __hip_atomic_load(&m.gc[-slot - 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int map_idx(const Shared &sh, const Map2 &m, int slot) {
  return slot >= 0 ? sh.ht[slot] : __hip_atomic_load(&m.gt[-slot - 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void map_set_idx(Shared &sh, const Map2 &m, int slot, int li) {
  if (slot >= 0) sh.ht[slot] = li; else __hip_atomic_store(&m.gt[-slot - 2], li, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// lower the cost of state ds; slot == -1: no room.  create: first to reach ds; improved: cost went down
__device__ __forceinline__ void map_relax(Shared &sh, const Map2 &m, int ds, double cost, int &slot, bool &create, bool &improved) {
  slot = -1; create = false; improved = false;
  // (a cost equal to the state's present one: the frame's winners may hang on the arcs' order -- see FSlot::wrec)
  const unsigned long long ck = d2o(cost);
  unsigned h = map_hash(ds);
#pragma unroll 1
  for (int p = 0; p < HPROBE; p++, h = (h + 1) & (HCAP - 1)) {
    unsigned cur = sh.hs[h];
    if (cur == HEMPTY) {
      cur = atomicCAS(&sh.hs[h], HEMPTY, (unsigned)ds);
      if (cur == HEMPTY) { create = true; slot = (int)h; break; }
    }
    if (cur == (unsigned)ds) { slot = (int)h; break; }
  }
  if (slot >= 0) {
    const unsigned long long oc = atomicMin(&sh.hc[slot], ck);
    improved = ck < oc;
    if (ck == oc) sh.tie = 1;
    return;
  }
  sh.g_used = 1;
  h = map_hash2(ds, m.gmask);
#pragma unroll 1
  for (int p = 0; p <= m.gmask; p++, h = (h + 1) & (unsigned)m.gmask) {
    unsigned cur = __hip_atomic_load(&m.gs[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == HEMPTY) {
      unsigned expect = HEMPTY;
      if (__hip_atomic_compare_exchange_strong(&m.gs[h], &expect, (unsigned)ds, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
        create = true; slot = -(int)h - 2;
        const int q = atomicAdd(&sh.n_g, 1);
        if (q < m.gl_cap) m.gl[q] = (int)h; else sh.error = FERR_FRAME;
        break;
      }
      cur = expect;
    }
    if (cur == (unsigned)ds) { slot = -(int)h - 2; break; }
  }
  if (slot == -1) return;
  const unsigned long long oc = __hip_atomic_fetch_min(&m.gc[-slot - 2], ck, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  improved = ck < oc;
  if (ck == oc) sh.tie = 1;
}
// where a state's winner key / record lives: its map entry (LDS slots first, then the second level's)
__device__ __forceinline__ int widx(int slot) { return slot >= 0 ? slot : HCAP + (-slot - 2); }
// the record of a lowering (FSlot::wrec)
__device__ __forceinline__ void put_wrec(gp<i32x4> wrec, int slot, double cost, int arc, int src) {
  const unsigned long long ck = d2o(cost);
  i32x4 r; r.x = (int)(unsigned)ck; r.y = (int)(unsigned)(ck >> 32); r.z = arc; r.w = src;
  wrec[slot >= 0 ? slot : HCAP + (-slot - 2)] = r;
}
// empty map for the next frame (all threads; followed by a barrier at the caller, who then sets g_used = n_g = 0).  lds_level = false:
// the second level only -- inside the frame loop make_tokens, the map's last reader, which walks every LDS slot anyway, has left the
// occupied ones empty (a frame of this search fills 570 of the 8192 slots: writing all of them again was 4 % of a frame)
__device__ __forceinline__ void map_clear(Shared &sh, const FSlot &s, bool lds_level = true) {
  if (lds_level) for (int i = threadIdx.x; i < HCAP; i += NT) { sh.hs[i] = HEMPTY; sh.hc[i] = ~0ull; }
  if (sh.g_used) {
    gp<unsigned> gs = uptr(s.g_hs); gp<unsigned long long> gc = uptr(s.g_hc);
    const int ng = sh.n_g;
    if (ng < 0 || ng > s.frame_cap) { for (int i = threadIdx.x; i < s.g_cap; i += NT) { gs[i] = HEMPTY; gc[i] = ~0ull; } }
    else {
      gp<const int> gl = uptr((const int *)s.g_list);
      for (int i = threadIdx.x; i < ng; i += NT) { const int h = gl[i]; gs[h] = HEMPTY; gc[h] = ~0ull; }
    }
  }
}


// The frame's tokens are numbered when all its states are known (after the epsilon closure): a state's token index within the frame
// is the number of occupied map slots before its own, in slot order -- so that make_tokens, which walks the slots in that order,
// writes runs of consecutive tokens (one 16-byte and one 8-byte record per lane, neighbouring lanes neighbouring tokens) -- then
// the second level's states in the order they are met.  Sets sh.n_new.  All threads; ends with a barrier.
__device__ void number_tokens(const FSlot &s, Shared &sh) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int PER = HCAP / NT, NG = PER * NW;
  unsigned long long occ[PER];
  gp<const i32x4> wrec = uptr((const i32x4 *)s.wrec);
  // (FSlot::wrec: does every state's record carry the state's final key?  The loads are issued first and checked behind the counting)
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  i32x2 wr[PER]; bool mine[PER];      // (the key half of the records)
#pragma unroll
  for (int u = 0; u < PER; u++) { mine[u] = sh.hs[u * NT + tid] != HEMPTY; wr[u].x = 0; wr[u].y = 0; if (mine[u]) wr[u] = *reinterpret_cast<gp<const i32x2>>(wrec + (u * NT + tid)); }
#pragma unroll
  for (int u = 0; u < PER; u++) {
    occ[u] = __ballot(mine[u]);
    if (lane == 0) sh.grp[u * NW + wave] = __popcll(occ[u]);
  }
  {
    bool bad = false;
#pragma unroll
    for (int u = 0; u < PER; u++) {
      if (!mine[u]) continue;
      const unsigned long long k = sh.hc[u * NT + tid];
      bad |= (unsigned)wr[u].x != (unsigned)k || (unsigned)wr[u].y != (unsigned)(k >> 32);
    }
    if (sh.g_used) {
      const int ng0 = sh.n_g < s.frame_cap ? sh.n_g : s.frame_cap;
      gp<const int> gl0 = uptr((const int *)s.g_list); gp<const unsigned long long> gc0 = uptr((const unsigned long long *)s.g_hc);
      for (int i = tid; i < ng0; i += NT) {
        const int h = gl0[i];
        const i32x4 w = wrec[HCAP + h];
        const unsigned long long k = gc0[h];
        bad |= (unsigned)w.x != (unsigned)k || (unsigned)w.y != (unsigned)(k >> 32);
      }
    }
    if (bad) sh.bad = 1;
  }
  __syncthreads();
  if (wave == 0) {
    static_assert(NG == 128, "two groups per lane");
    const int a = sh.grp[2 * lane], b = sh.grp[2 * lane + 1];
    const int incl = wave_incl_scan(a + b);
    sh.grp[2 * lane] = incl - a - b; sh.grp[2 * lane + 1] = incl - b;
    if (lane == 63) sh.grp[NG] = incl;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < PER; u++)
    if ((occ[u] >> lane) & 1ull) sh.ht[u * NT + tid] = sh.grp[u * NW + wave] + __popcll(occ[u] & ((1ull << lane) - 1ull));
  const int n_lds = sh.grp[NG];
  const int ng = sh.g_used ? (sh.n_g < s.frame_cap ? sh.n_g : s.frame_cap) : 0;
  if (ng > 0) {       // the second level's states: in the order of the frame's list of them
    gp<const int> gl = uptr((const int *)s.g_list); gp<int> gt = uptr(s.g_ht);
    for (int i = tid; i < ng; i += NT) gt[gl[i]] = n_lds + i;
  }
  if (tid == 0) sh.n_new = n_lds + ng;
  __syncthreads();
}

// ProcessNonemitting (faster-decoder.cc:306-346) to the fixed point: work lists hold states; a
// state whose cost exceeds the cutoff is not expanded, an arc that would is dropped.
__device__ void epsilon_closure(const FGraph &g, const FSlot &s, double cutoff, Shared &sh) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  gp<int> elist = uptr(s.elist);
  gp<i32x4> wrec = uptr(s.wrec);
  const Map2 m2 = map2_of(s);
  const int frame_cap = s.frame_cap, front_cap = s.front_cap;
  int cur = 0;
  for (int iter = 0;; iter++) {
    const int ncur = sh.fcnt[cur] < front_cap ? sh.fcnt[cur] : front_cap;
    if (ncur == 0) break;
    if (iter >= MAX_EPS_ITERS) { if (tid == 0) sh.error = FERR_EPS; break; }
    const int nxt = cur == 2 ? 0 : cur + 1, spare = nxt == 2 ? 0 : nxt + 1;
    if (tid == 0) sh.fcnt[spare] = 0;
    gp<const int> fcur = uptr(s.front[cur]);
    gp<int> fnext = uptr(s.front[nxt]);
    const int CH = wave_chunk(ncur);
    for (int base = wave * CH; base < ncur; base += NW * CH) {
      const int i = base + lane;
      const bool have = lane < CH && i < ncur;
      const int st = have ? fcur[i] : 0;
      const i32x4 sr = uptr(g.st)[st];
      const int slot = have ? map_find(sh, m2, st) : -1;
      const double c = slot != -1 ? o2d(map_key(sh, m2, slot)) : 0.0;
      const bool exp = have && slot != -1 && !(c > cutoff);
      const int a0 = st + 1, cnt = exp ? sr.y : 0;      // (a state's arcs follow its record)
      const int src_code = -2 - widx(slot);               // (FSlot::wrec: the source of an epsilon arc is a state of this frame, named by its map entry)
      wave_expand(sh.wx[threadIdx.x >> 6], cnt, [&](bool valid, int owner, int j) {
        const int a = __shfl(a0, owner, 64) + j;
        const double cc = __shfl(c, owner, 64);
        const int sc = __shfl(src_code, owner, 64);
        bool create = false, push = false, eps = false; int ds = 0, nslot = -1;   // (-1: none / no room)
        if (valid) {
          const i32x4 arc = uptr(g.arcs)[a];
          const double nc = cc + (double)asf(arc.y);
          if (!(nc > cutoff)) {
            ds = arc.x & STMASK;
            bool improved;
            map_relax(sh, m2, ds, nc, nslot, create, improved);
            if (nslot == -1) sh.error = FERR_MAP;
            else if (improved && map_key(sh, m2, nslot) == d2o(nc)) put_wrec(wrec, nslot, nc, a, sc);      // (still the state's cost: see pass 2)
            push = nslot != -1 && (create || improved) && arc.x < 0;
            eps = create && arc.x < 0;
          }
        }
        const int qe = wave_alloc(&sh.n_eps, eps);
        if (eps) { if (qe < frame_cap) elist[qe] = ds; else sh.error = FERR_FRAME; }
        const int q = wave_alloc(&sh.fcnt[nxt], push);
        if (push) { if (q < front_cap) fnext[q] = ds; else sh.error = FERR_FRONT; }
      });
    }
    __syncthreads();
    if (sh.error != FERR_NONE) break;
    cur = nxt;
  }
  __syncthreads();
}

// entry e of the frame's map: LDS slots first, then (if used) the HBM slots
__device__ __forceinline__ bool map_entry(const Shared &sh, const Map2 &m, int e, int &state, unsigned long long &key, int &li) {
  if (e < HCAP) {
    const unsigned tag = sh.hs[e];
    if (tag == HEMPTY) return false;
    state = (int)tag; key = sh.hc[e]; li = sh.ht[e];
    return true;
  }
  const unsigned tag = m.gs[e - HCAP];
  if (tag == HEMPTY) return false;
  state = (int)tag; key = m.gc[e - HCAP]; li = m.gt[e - HCAP];
  return true;
}

// the winning epsilon arc of every state: the frame's states that have epsilon arcs (FSlot::elist)
// re-walk them with the final costs; an arc that reproduces its destination's cost competes with its index
__device__ void epsilon_winners(const FGraph &g, const FSlot &s, double cutoff, int tok_base, Shared &sh) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  gp<unsigned long long> win = uptr(s.win);
  gp<const int> elist = uptr(s.elist);
  const Map2 m2 = map2_of(s);
  const int n = sh.n_eps < s.frame_cap ? sh.n_eps : s.frame_cap;
  const int CH = wave_chunk(n);
  for (int base = wave * CH; base < n; base += NW * CH) {
    const int i = base + lane;
    const bool have = lane < CH && i < n;
    const int st = have ? elist[i] : 0;
    const i32x4 sr = uptr(g.st)[st];
    const int slot = have ? map_find(sh, m2, st) : -1;
    const double c = slot != -1 ? o2d(map_key(sh, m2, slot)) : 0.0;
    const bool exp = have && slot != -1 && !(c > cutoff);
    const int a0 = st + 1, cnt = exp ? sr.y : 0;
    const int my_tok = exp ? tok_base + map_idx(sh, m2, slot) : 0;
    wave_expand(sh.wx[threadIdx.x >> 6], cnt, [&](bool valid, int owner, int j) {
      const int a = __shfl(a0, owner, 64) + j;
      const double cc = __shfl(c, owner, 64);
      const int src_tok = __shfl(my_tok, owner, 64);
      if (valid) {
        const i32x4 arc = uptr(g.arcs)[a];
        const double nc = cc + (double)asf(arc.y);
        if (!(nc > cutoff)) {
          const int ds_slot = map_find(sh, m2, arc.x & STMASK);
          if (ds_slot != -1 && d2o(nc) == map_key(sh, m2, ds_slot))
            (void)__hip_atomic_fetch_min(&win[widx(ds_slot)], ((unsigned long long)(unsigned)a << 32) | (unsigned)src_tok,
                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    });
  }
}

// one token per map entry, appended to the utterance's history at tok_base (numbered by number_tokens: a wave writes runs of
// consecutive tokens); then GetCutoff for the frame's turn as the current frame, from the cost keys while they are in registers.  The LDS level is walked HCAP / NT entries per thread with every winner fetched before the first is used;
// the winner entries that were read are left empty for the next frame.
// slow: the winners are in FSlot::win (winners / epsilon_winners ran); otherwise in FSlot::wrec (checked by number_tokens)
template <bool slow>
__device__ void make_tokens(const FSlot &s, int tok_base, gp<double> cost_out, gp<const double> cost_cur, int cur_b, const FCfg &cfg, Shared &sh, int &par,
                            double &weight_cutoff, float &adaptive_beam, int &best_state) {
  gp<i32x4> tok = uptr(s.tok);
  gp<unsigned long long> win = uptr(s.win);
  gp<const i32x4> wrec = uptr((const i32x4 *)s.wrec);
  const Map2 m2 = map2_of(s);
  constexpr int PER = HCAP / NT;
  int st[PER], li[PER]; unsigned long long key[PER], w[PER]; bool have[PER];
  // (a record's source: a token of the previous frame as it is, a state of this frame through its map entry's token index)
  auto src_of = [&](int src) -> int {
    if (src >= -1) return src;
    const int e = -2 - src;
    return tok_base + (e < HCAP ? sh.ht[e] : m2.gt[e - HCAP]);
  };
#pragma unroll
  for (int u = 0; u < PER; u++) {
    const int e = u * NT + threadIdx.x;
    const unsigned tag = sh.hs[e];
    have[u] = tag != HEMPTY;
    st[u] = (int)tag; key[u] = sh.hc[e]; li[u] = have[u] ? sh.ht[e] : 0;
    if (have[u]) { sh.hs[e] = HEMPTY; sh.hc[e] = ~0ull; }      // (this thread was the slot's last reader: map_clear)
  }
  // ---- the frame's costs first (the half of FSlot::cost that is not the current frame's): a token's record carries the cost of the
  //      arc that made it, cost - source's cost in double rounded to float -- what GetBestPath computes from the two tokens
  //      (faster-decoder.cc:122-127) -- and the source may be a token of THIS frame (epsilon arcs).  The history keeps no costs.
  const int ng_all = sh.g_used ? (sh.n_g < s.frame_cap ? sh.n_g : s.frame_cap) : 0;
#pragma unroll
  for (int u = 0; u < PER; u++) if (have[u]) cost_out[li[u]] = o2d(key[u]);
  for (int i = threadIdx.x; i < ng_all; i += NT) {
    const int h = m2.gl[i];
    if (slow) { int st1, li1; unsigned long long key1; if (map_entry(sh, m2, HCAP + h, st1, key1, li1)) cost_out[li1] = o2d(key1); }
    else cost_out[m2.gt[h]] = o2d(m2.gc[h]);
  }
  __syncthreads();
  auto arc_cost = [&](double c, int src) -> int {       // src: absolute token index (-1: the start token, no arc)
    if (src < 0) return 0;
    const double pc = src >= tok_base ? cost_out[src - tok_base] : cost_cur[src - cur_b];
    return __float_as_int((float)(c - pc));
  };
  if (!slow) {
#pragma unroll
    for (int u = 0; u < PER; u++) {
      if (!have[u]) continue;
      const i32x4 wr = wrec[u * NT + threadIdx.x];
      i32x4 r; r.x = src_of(wr.w); r.y = wr.z; r.z = st[u]; r.w = arc_cost(o2d(key[u]), r.x);
      tok[tok_base + li[u]] = r;
    }
    for (int i = threadIdx.x; i < ng_all; i += NT) {
      const int h = m2.gl[i];
      const i32x4 wr = wrec[HCAP + h];
      i32x4 r; r.x = src_of(wr.w); r.y = wr.z; r.z = (int)m2.gs[h]; r.w = arc_cost(o2d(m2.gc[h]), r.x);
      tok[tok_base + m2.gt[h]] = r;
    }
  }
#pragma unroll
  for (int u = 0; u < PER; u++) { w[u] = ~0ull; if (slow && have[u]) w[u] = win[u * NT + threadIdx.x]; }
#ifdef VBH_FASTER_PROF
  { unsigned long long x = 0;
#pragma unroll
    for (int u = 0; u < PER; u++) x ^= w[u];
    if (x == 0x123456789ull) sh.u0 = x; }
  FPHASE(10)
#endif
#pragma unroll
  for (int u = 0; u < PER; u++) {
    if (!slow || !have[u]) continue;
    i32x4 r; r.x = (int)(unsigned)w[u]; r.y = (int)(unsigned)(w[u] >> 32); r.z = st[u]; r.w = 0;
    if (w[u] == ~0ull) { r.x = -1; r.y = -1; }      // the start token
    else win[u * NT + threadIdx.x] = ~0ull;
    r.w = arc_cost(o2d(key[u]), r.x);
    tok[tok_base + li[u]] = r;
  }
  if (slow) {
    for (int i = threadIdx.x; i < ng_all; i += NT) {
      const int e = HCAP + m2.gl[i];
      int st1, li1; unsigned long long key1;
      if (!map_entry(sh, m2, e, st1, key1, li1)) continue;
      const unsigned long long w1 = win[e];
      i32x4 r; r.x = (int)(unsigned)w1; r.y = (int)(unsigned)(w1 >> 32); r.z = st1; r.w = 0;
      if (w1 == ~0ull) { r.x = -1; r.y = -1; }
      else win[e] = ~0ull;
      r.w = arc_cost(o2d(key1), r.x);
      tok[tok_base + li1] = r;
    }
  }
  // GetCutoff of the frame just made, for its turn as the current frame: its costs are the keys at hand (the second level's, if any,
  // from its table, which stays until the next map is started)
  const int ng = sh.g_used ? (sh.n_g < s.frame_cap ? sh.n_g : s.frame_cap) : 0;
  get_cutoff([&](auto f) {
#pragma unroll
    for (int u = 0; u < PER; u++) if (have[u]) f(key[u], (unsigned)(st[u] & STMASK));
    for (int i = threadIdx.x; i < ng; i += NT) f(m2.gc[m2.gl[i]], m2.gs[m2.gl[i]] & (unsigned)STMASK);
  }, [&](auto f) {
    // (the counts and the order statistics read the frame's costs where they were just written, token by token: three coalesced loads
    // per thread and pass whatever the map looks like -- from the map, a thread's eight slots hold anything from none to eight entries,
    // and every pass gathered the second level's keys through its list again)
    const int nn = sh.n_new;
    for (int i = threadIdx.x; i < nn; i += NT) f(d2o(cost_out[i]));
  }, sh.n_new, cfg, sh, par, weight_cutoff, adaptive_beam, best_state);
}

// Tokens no current token can reach through back pointers will never be part of a best path (the
// reference frees them by reference counting, faster-decoder.h:136-160).  When the history is about
// to run out, every current token marks its ancestors down to gc_base (tokens below survived an
// earlier collection and stay), the marked tokens are packed -- through a staging buffer, order
// kept, so the current frame stays a contiguous tail -- and back pointers are rewritten.
// Returns the new position of the current frame; everything below it is then settled.
__device__ int collect_garbage(const FSlot &s, int gc_base, int cur_b, int cur_n, Shared &sh) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  gp<i32x4> tok = uptr(s.tok);
  gp<int> remap = uptr(s.remap);
  gp<i32x4> st_tok = uptr(s.stage_tok);      // (the current frame's tokens all survive, in their order: their costs stay where they are, FSlot::cost)
  const int end = cur_b + cur_n;
  for (int i = gc_base + tid; i < end; i += NT) remap[i] = 0;
  __syncthreads();
  for (int i = cur_b + tid; i < end; i += NT) {
    int t = i;
    while (t >= gc_base) {
      if (__hip_atomic_exchange(&remap[t], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) break;
      t = tok[t].x;
    }
  }
  __syncthreads();
  // survivors per thread chunk, exclusive scan over the workgroup (sh.ht is free between frames)
  const int len = end - gc_base, per = (len + NT - 1) / NT;
  const int lo = gc_base + tid * per, hi = lo + per < end ? lo + per : end;
  int c = 0;
  for (int t = lo; t < hi; t++) c += remap[t];
  const int incl = wave_incl_scan(c);
  if (lane == 63) sh.ht[wave] = incl;
  __syncthreads();
  int wbase = 0, total = 0;
  for (int w = 0; w < NW; w++) { const int x = sh.ht[w]; if (w < wave) wbase += x; total += x; }
  __syncthreads();
  if (total > s.stage_cap) { if (tid == 0) sh.error = FERR_TOK; __syncthreads(); return cur_b; }
  int idx = wbase + incl - c;
  for (int t = lo; t < hi; t++) {
    if (remap[t]) { st_tok[idx] = tok[t]; remap[t] = gc_base + idx; idx++; }
    else remap[t] = -1;
  }
  __syncthreads();
  for (int j = tid; j < total; j += NT) {
    i32x4 r = st_tok[j];
    if (r.x >= gc_base) r.x = remap[r.x];
    tok[gc_base + j] = r;
  }
  __syncthreads();
  return gc_base + total - cur_n;
}

template <int ADM>
__global__ __launch_bounds__(NT, 4) void faster_kernel(FGraph g, FCfg cfg, const FSlot *slots, const float *__restrict__ loglikes, int P,
                                                       const FUtt *__restrict__ utts, int n_work, int *work_counter, FResult *results,
                                                       i32x4 *paths, int path_cap) {   // best path per utterance, last arc first: {ilabel, olabel, weight, tot_cost}
  __shared__ Shared sh;
  __shared__ FSlot s_slot;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  {
    const int *src = reinterpret_cast<const int *>(&slots[blockIdx.x]);
    int *dst = reinterpret_cast<int *>(&s_slot);
    for (int i = tid; i < (int)(sizeof(FSlot) / sizeof(int)); i += NT) dst[i] = src[i];
  }
  __syncthreads();
  const FSlot &s = s_slot;
  int par = 0;
#ifdef VBH_FASTER_PROF
  long long *tph = sh.tph;
  if (tid == 0) { for (int i = 0; i < 16; i++) sh.tph[i] = 0; sh.t_last = (long long)__builtin_amdgcn_s_memtime(); sh.n_sel = sh.n_sel_over = sh.n_sel_radix = 0; sh.s_cand = sh.s_new = sh.s_g = sh.s_tok = 0; sh.s_gframes = 0; }
#endif
  while (true) {
    __syncthreads();
    if (tid == 0) sh.utt = atomicAdd(work_counter, 1);
    __syncthreads();
    if (sh.utt >= n_work) break;
    const int utt = sh.utt;
    const FUtt ud = utts[utt];
    const int T = ud.T;
    gp<i32x4> tok = uptr(s.tok); gp<double> cost = uptr(s.cost);
    gp<unsigned long long> win = uptr(s.win);
    gp<double> c_w = uptr(s.c_w); gp<i32x4> c_r = uptr(s.c_r);
    const Map2 m2 = map2_of(s);
    if (tid == 0) { sh.g_used = 1; sh.n_g = -1; }      // a previous utterance may have left entries in the second level
    __syncthreads();
    map_clear(sh, s);
    for (int i = tid; i < HCAP + s.g_cap; i += NT) win[i] = ~0ull;      // (an utterance that ended in an error leaves winner keys behind)
    __syncthreads();
    if (tid == 0) { sh.n_new = 0; sh.n_cand = 0; sh.n_eps = 0; sh.fcnt[0] = sh.fcnt[1] = sh.fcnt[2] = 0; sh.error = FERR_NONE; sh.g_used = 0; sh.n_g = 0; }
    __syncthreads();
    // ---- InitDecoding (:36-45): the start token, then the epsilon closure with cutoff FLT_MAX
    if (tid == 0) {
      int slot; bool create, improved;
      const int start = ud.start >= 0 ? ud.start : g.start;
      sh.tie = 0; sh.bad = 0; sh.n_slow = 0; sh.u_tie = 0; sh.u_bad = 0;
      sh.vmin[0] = sh.vmin[1] = ~0ull; sh.vmax[0] = sh.vmax[1] = 0ull;
      map_relax(sh, m2, start, 0.0, slot, create, improved);
      if (slot != -1) put_wrec(uptr(s.wrec), slot, 0.0, -1, -1);       // (the start token: no arc, no source)
      if (uptr(g.st)[start].y > 0) { uptr(s.front[0])[0] = start; sh.fcnt[0] = 1; uptr(s.elist)[0] = start; sh.n_eps = 1; }
    }
    __syncthreads();
    double weight_cutoff = (double)INFINITY; float adaptive_beam = cfg.beam;      // of the current frame (make_tokens)
    int best_state = -1;                                                          // ... and its cheapest token's state (ties: the smallest)
    if (ADM == FADM_ORDERED) { if (tid < 64) sh.bmin[tid] = ~0ull; if (tid == 0) sh.init_bound = ~0ull; }
    epsilon_closure(g, s, (double)FLT_MAX, sh);
    number_tokens(s, sh);
    if (tid == 0 && sh.n_new > s.frame_cap) sh.error = FERR_FRAME;
    int cur_b = 0, cur_n = sh.n_new < s.frame_cap ? sh.n_new : s.frame_cap;
    if (cur_n > s.tok_cap && tid == 0) sh.error = FERR_TOK;
    __syncthreads();
    if (sh.error == FERR_NONE) {
      const bool slow0 = (sh.tie | sh.bad) != 0 || VBH_FASTER_SLOW;      // (uniform: number_tokens ended with a barrier)
      if (slow0) {
        // (the slow path's start token is the state without a winner key)
        epsilon_winners(g, s, (double)FLT_MAX, 0, sh);
        __syncthreads();
        if (tid == 0) sh.n_slow++;
      }
      if (slow0) make_tokens<true>(s, 0, cost, cost, 0, cfg, sh, par, weight_cutoff, adaptive_beam, best_state);
      else make_tokens<false>(s, 0, cost, cost, 0, cfg, sh, par, weight_cutoff, adaptive_beam, best_state);
    }
    __syncthreads();
    int frames_done = 0;
    int gc_base = 0;
    int chalf = 0;          // the half of FSlot::cost that holds the current frame's costs
    FPHASE(9)
    for (int frame = 0; frame < T && sh.error == FERR_NONE; frame++) {
      if (cur_b + cur_n + s.frame_cap > s.tok_cap) {      // make room for whatever this frame may create
        cur_b = collect_garbage(s, gc_base, cur_b, cur_n, sh);
        gc_base = cur_b;
        if (sh.error != FERR_NONE) break;
      }
      FPHASE(0)
      const float *llrow = loglikes + (ud.row0 + frame) * (long long)P;
      gp<const double> ccost = cost + (size_t)chalf * s.frame_cap;
      gp<const i32x4> ctok = tok + cur_b;
      const int n = cur_n;
      // (GetCutoff, :151-213: weight_cutoff and adaptive_beam of this frame were computed when its tokens were made)
      FPHASE(1)
      // the map of the frame about to be built
      map_clear(sh, s, /*lds_level=*/false);
      __syncthreads();
      if (tid == 0) { sh.n_new = 0; sh.n_cand = 0; sh.n_eps = 0; sh.fcnt[0] = sh.fcnt[1] = sh.fcnt[2] = 0; sh.g_used = 0; sh.n_g = 0; sh.cand_bound = ~0ull; sh.tie = 0; sh.bad = 0; }
      __syncthreads();
      // ---- pass 1 (:224-303, arc-parallel): candidates + min of new_weight + adaptive_beam
      FPHASE(2)
      unsigned long long mymin = ~0ull;
      {
        const int cand_cap = s.cand_cap;
        // Rounds of <= 64 tokens per wave, software-pipelined as decode_kernel's pass 1 (round 5): while the arcs of round r are expanded, the
        // state records of round r + 1 and the token records of round r + 2 are on their way (a round is otherwise a chain of four
        // dependent loads: token, state, arc, score)
        const int CH = wave_chunk(n), RS = NW * CH;
        struct TokIn { int z; double c; };
        auto ld_tok = [&](int b) -> TokIn {       // {state, cost} of this lane's token of the round starting at b (no token: a cost no cutoff admits)
          const int i_ = b + lane;
          TokIn t_; t_.z = 0; t_.c = (double)INFINITY;
          if (b < n) { const bool have_ = lane < CH && i_ < n; const int z_ = ctok[have_ ? i_ : 0].z; const double c_ = ccost[have_ ? i_ : 0]; if (have_) { t_.z = z_; t_.c = c_; } }
          return t_;
        };
        auto ld_st = [&](const TokIn &t_) -> i32x4 { return uptr(g.st)[(t_.c < weight_cutoff) ? (t_.z & STMASK) : 0]; };
        const int b0 = wave * CH;
        TokIn recC = ld_tok(b0), recB = ld_tok(b0 + RS);
        i32x4 srC = ld_st(recC);
        for (int base = b0; base < n; base += RS) {
          const i32x4 srB = ld_st(recB);
          const TokIn recA = ld_tok(base + 2 * RS);
          i32x4 rec; rec.x = 0; rec.y = 0; rec.z = recC.z; rec.w = 0;
          const double c = recC.c;
          const i32x4 sr = srC;
          recC = recB; srC = srB; recB = recA;
          const bool exp = c < weight_cutoff;       // (false for lanes without a token)
          // (the sign bit of the first-arc index marks the frame's best token: its arcs give the initial bound, faster-decoder.cc:241-255)
          const int a0 = ((rec.z & STMASK) + 1 + sr.y) | ((ADM == FADM_ORDERED && exp && (rec.z & STMASK) == best_state) ? EPSBIT : 0), cnt = exp ? sr.z - sr.y : 0;
          // ordered admission: the token's bucket, and its bucket's bound as it stands at the start of the round (lane b of the scan
          // holds bucket b's: min(init_bound, bmin of the buckets below b)) -- a filter here, the final bounds decide in pass 2
          int bkt = 0; unsigned long long bnd_t = ~0ull;
          if (ADM == FADM_ORDERED) {
            bkt = adm_bucket(sr.x & STMASK);
            const unsigned long long pre = wave_min_scan_u64(lane == 0 ? sh.init_bound : sh.bmin[lane - 1]);
            bnd_t = shfl_u64(pre, bkt);
          }
          wave_expand_n<UA>(sh.wx[threadIdx.x >> 6], cnt, [&](const bool (&valid)[UA], const int (&owner)[UA], const int (&j)[UA]) {
            int ao[UA], a[UA]; double cc[UA]; i32x4 arc[UA]; float sc[UA];
#pragma unroll
            for (int u = 0; u < UA; u++) {
              ao[u] = __shfl(a0, owner[u], 64);
              a[u] = (ao[u] & ~EPSBIT) + j[u];
              cc[u] = __shfl(c, owner[u], 64);
            }
#pragma unroll
            for (int u = 0; u < UA; u++) arc[u] = uptr(g.arcs)[valid[u] ? a[u] : 0];
#pragma unroll
            for (int u = 0; u < UA; u++) sc[u] = llrow[valid[u] ? arc[u].z : 0];
            double nw[UA]; unsigned long long kk[UA]; bool keep[UA]; int bko[UA];
#pragma unroll
            for (int u = 0; u < UA; u++) {
              nw[u] = 0.0; kk[u] = ~0ull;
              if (valid[u]) {
                const float ac_cost = -(cfg.acwt * sc[u]);
                nw[u] = ((double)asf(arc[u].y) + cc[u]) + (double)ac_cost;    // arc.weight + tok->cost + ac_cost
                kk[u] = d2o(nw[u] + (double)adaptive_beam);
                mymin = kk[u] < mymin ? kk[u] : mymin;
              }
            }
            if (ADM == FADM_ORDERED) {
              // Every arc may lower its bucket's minimum, the best token's arcs the initial bound -- but only an arc that gets below the
              // bound of its OWN bucket can change the bound of a later one (those are minima over this bucket's bound and more), so all
              // but the frame's few best arcs spare the atomic.  An arc at or above its bucket's present bound is not written (:266: the
              // final bound is no larger).
#pragma unroll
              for (int u = 0; u < UA; u++) {
                bko[u] = __shfl(bkt, owner[u], 64);
                const unsigned long long bnd = shfl_u64(bnd_t, owner[u]);
                keep[u] = valid[u] && d2o(nw[u]) < bnd;
                if (valid[u]) {
                  if (kk[u] < bnd) atomicMin(&sh.bmin[bko[u]], kk[u]);
                  if (ao[u] < 0) atomicMin(&sh.init_bound, kk[u]);
                }
              }
            } else {
              // An arc whose new_weight has already reached a bound on next_cutoff -- the smallest new_weight + adaptive_beam anybody in
              // the workgroup has seen so far, kept in LDS -- cannot be admitted (the final next_cutoff is the minimum over all arcs, so
              // it is no larger): it is not written to the candidate list.  The admitted set is still {new_weight < final next_cutoff}.
              unsigned long long rk = ~0ull;
#pragma unroll
              for (int u = 0; u < UA; u++) { bko[u] = 0; rk = kk[u] < rk ? kk[u] : rk; }
              const unsigned long long wk = wave_min_u64(rk);
              if (lane == 0 && wk != ~0ull) atomicMin(&sh.cand_bound, wk);
              const double bound = o2d(sh.cand_bound);
#pragma unroll
              for (int u = 0; u < UA; u++) keep[u] = valid[u] && nw[u] < bound;
            }
            int q[UA];
            wave_alloc_n<UA>(&sh.n_cand, keep, q);
#pragma unroll
            for (int u = 0; u < UA; u++) {
              if (keep[u]) {
                if (q[u] < cand_cap) {
                  c_w[q[u]] = nw[u];
                  i32x4 r; r.x = a[u]; r.y = cur_b + base + owner[u]; r.z = arc[u].x; r.w = bko[u];
                  c_r[q[u]] = r;
                } else sh.error = FERR_CAND;
              }
            }
          });
        }
      }
      mymin = block_min_u64(mymin, sh, par);
      const double next_cutoff = mymin == ~0ull ? (double)INFINITY : o2d(mymin);
      const int ncand = sh.n_cand < s.cand_cap ? sh.n_cand : s.cand_cap;
      // ordered admission: the buckets' FINAL bounds (every arc's atomic is in: the reduction's barrier), lane b holds bucket b's key; a
      // candidate is admitted iff its new_weight lies strictly below its bucket's (d2o is monotone: the comparison of the keys)
      const unsigned long long adm_pre = ADM == FADM_ORDERED ? wave_min_scan_u64(lane == 0 ? sh.init_bound : sh.bmin[lane - 1]) : 0ull;
      auto admitted = [&](double nw_, int bucket_) -> bool {
        if (ADM == FADM_ORDERED) return d2o(nw_) < shfl_u64(adm_pre, bucket_);
        return nw_ < next_cutoff;
      };
      FPHASE(3)
      // ---- pass 2: admit new_weight < next_cutoff; cheapest per state
      {
        gp<int> front0 = uptr(s.front[0]); gp<int> elist0 = uptr(s.elist);
        gp<i32x4> wrec = uptr(s.wrec);
        const int frame_cap = s.frame_cap, front_cap = s.front_cap;
#ifndef VBH_FASTER_UC
#define VBH_FASTER_UC 2
#endif
        constexpr int UC = VBH_FASTER_UC;        // candidates a thread has in flight
        for (int base = 0; base < ncand; base += NT * UC) {
          double nw[UC]; i32x4 r[UC]; bool ok[UC];
#pragma unroll
          for (int u = 0; u < UC; u++) {
            const int q = base + u * NT + tid;
            ok[u] = q < ncand;
            nw[u] = c_w[ok[u] ? q : 0];
            r[u] = c_r[ok[u] ? q : 0];
          }
#pragma unroll
          for (int u = 0; u < UC; u++) {
            const bool adm = admitted(nw[u], r[u].w & 63) && ok[u];      // (all 64 lanes take part in the bound's shuffle)
            int slot = -1; bool create = false, improved;
            if (adm) {
              map_relax(sh, m2, r[u].z & STMASK, nw[u], slot, create, improved);
              if (slot == -1) sh.error = FERR_MAP;
              // FSlot::wrec: the way into the state, recorded by whoever lowers its cost -- if that cost still stands when the atomic has
              // returned: several lanes of a wave often lower the SAME state in one instruction (a state's candidates are neighbours), all of
              // them "improved", and their stores to the one address would land in no particular order (65 % of the frames then failed the
              // check in number_tokens); behind the wave's atomics only the lowest cost reads itself back.  Between waves the window stays open
              // (a later, lower cost whose store overtakes): that is what the check is for
              else if (improved && map_key(sh, m2, slot) == d2o(nw[u])) put_wrec(wrec, slot, nw[u], r[u].x, r[u].y);
            }
            const bool pushf = create && r[u].z < 0;
            const int f = wave_alloc(&sh.fcnt[0], pushf);
            if (pushf) {      // (a created state with epsilon arcs: also the head of the frame's epsilon-state list)
              if (f < front_cap) front0[f] = r[u].z & STMASK; else sh.error = FERR_FRONT;
              if (f < frame_cap) elist0[f] = r[u].z & STMASK; else sh.error = FERR_FRAME;
            }
          }
        }
      }
      __syncthreads();
      if (tid == 0) sh.n_eps = sh.fcnt[0];     // the closure appends the states it creates (it starts with a barrier-free read of fcnt only)
      __syncthreads();
#ifdef VBH_FASTER_DEBUG
      if (tid == 0 && utt == 0 && frame < 12) printf("frame %d after pass2: n=%d ncand=%d n_new=%d g_used=%d err=%d fcnt0=%d\n", frame, n, ncand, sh.n_new, sh.g_used, sh.error, sh.fcnt[0]);
#endif
      if (sh.error != FERR_NONE) break;
      FPHASE(4)
      // ---- ProcessNonemitting(next_cutoff) to the fixed point
      epsilon_closure(g, s, next_cutoff, sh);
      FPHASE(5)
      if (sh.error != FERR_NONE) break;
      number_tokens(s, sh);
      if (sh.n_new > s.frame_cap) { if (tid == 0) sh.error = FERR_FRAME; __syncthreads(); break; }
      const int nb = cur_b + cur_n;
      const int n_new = sh.n_new;
#ifdef VBH_FASTER_PROF
      if (tid == 0) { sh.s_cand += ncand; sh.s_new += n_new; sh.s_tok += n; if (sh.g_used) { sh.s_g += sh.n_g; sh.s_gframes++; } }
#endif
#ifdef VBH_FASTER_DEBUG
      if (tid == 0 && utt == 0 && frame < 12) printf("frame %d: n=%d ncand=%d n_new=%d g_used=%d next_cutoff=%f\n", frame, n, ncand, sh.n_new, sh.g_used, next_cutoff);
#endif
      if (nb + n_new > s.tok_cap) { if (tid == 0) sh.error = FERR_TOK; __syncthreads(); break; }
      // ---- winners.  Fast path (nearly every frame): every state's record of its last lowering carries its final key and no equal cost
      //      was seen -- the records ARE the winners (FSlot::wrec).  Slow path: the emitting candidates and the epsilon arcs that reproduce
      //      a state's final cost compete with their arc index (rounds 1-4)
      const bool slow = (sh.tie | sh.bad) != 0 || VBH_FASTER_SLOW;      // (uniform: number_tokens ended with a barrier; nobody writes the flags before the next frame's resets)
      if (slow)
      for (int base = 0; base < ncand; base += NT * 2) {
        double nw[2]; i32x4 r[2]; bool ok[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const int q = base + u * NT + tid;
          ok[u] = q < ncand;
          nw[u] = c_w[ok[u] ? q : 0];
          r[u] = c_r[ok[u] ? q : 0];
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const bool adm = admitted(nw[u], r[u].w & 63) && ok[u];
          if (adm) {
            const int slot = map_find(sh, m2, r[u].z & STMASK);
            if (slot != -1 && d2o(nw[u]) == map_key(sh, m2, slot))
              (void)__hip_atomic_fetch_min(&win[widx(slot)], ((unsigned long long)(unsigned)r[u].x << 32) | (unsigned)r[u].y,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
      FPHASE(6)
      if (ADM == FADM_ORDERED) {      // (the bounds' last readers took them into registers before pass 2; two barriers lie between here and pass 1 of the next frame)
        if (tid >= NT - 64) sh.bmin[tid - (NT - 64)] = ~0ull;
        if (tid == 0) sh.init_bound = ~0ull;
      }
      if (slow) {
        epsilon_winners(g, s, next_cutoff, nb, sh);
        __syncthreads();
        if (tid == 0) { sh.n_slow++; sh.u_tie += sh.tie; sh.u_bad += sh.bad; }
      }
      FPHASE(7)
      if (slow) make_tokens<true>(s, nb, cost + (size_t)(chalf ^ 1) * s.frame_cap, ccost, cur_b, cfg, sh, par, weight_cutoff, adaptive_beam, best_state);
      else make_tokens<false>(s, nb, cost + (size_t)(chalf ^ 1) * s.frame_cap, ccost, cur_b, cfg, sh, par, weight_cutoff, adaptive_beam, best_state);
      FPHASE(14)
      __syncthreads();
      FPHASE(8)
      cur_b = nb; cur_n = n_new; chalf ^= 1;
      frames_done = frame + 1;
    }
    __syncthreads();
    // ---- ReachedFinal / GetBestPath (:78-147)
    int r_ok = 0, r_final = 0, r_len = 0, r_state = -1; float r_fc = 0.0f;
    if (sh.error == FERR_NONE && frames_done == T) {
      gp<const double> ccost = cost + (size_t)chalf * s.frame_cap;
      gp<const i32x4> ctok = tok + cur_b;
      unsigned long long anyf = 0;
      for (int i = tid; i < cur_n; i += NT) {
        const float fc = asf(uptr(g.st)[ctok[i].z & STMASK].w);
        anyf += (ccost[i] != (double)INFINITY && fc != INFINITY);
      }
      anyf = block_sum_u64(anyf, sh, par);
      const bool is_final = anyf > 0;
      unsigned long long bkey = ~0ull;
      for (int i = tid; i < cur_n; i += NT) {
        const double c = is_final ? ccost[i] + (double)asf(uptr(g.st)[ctok[i].z & STMASK].w) : ccost[i];
        if (is_final && c == (double)INFINITY) continue;
        const unsigned long long k = d2o(c);
        bkey = k < bkey ? k : bkey;
      }
      bkey = block_min_u64(bkey, sh, par);
      unsigned long long bsel = ~0ull;      // among the tokens at the best cost: smallest state (<< 32 | index)
      for (int i = tid; i < cur_n; i += NT) {
        const int st = ctok[i].z & STMASK;
        const double c = is_final ? ccost[i] + (double)asf(uptr(g.st)[st].w) : ccost[i];
        if (is_final && c == (double)INFINITY) continue;
        if (d2o(c) == bkey) { const unsigned long long k = ((unsigned long long)(unsigned)st << 32) | (unsigned)i; bsel = k < bsel ? k : bsel; }
      }
      bsel = block_min_u64(bsel, sh, par);
      if (bsel != ~0ull && cur_n > 0) {
        r_ok = 1; r_final = is_final; r_state = (int)(bsel >> 32);
        r_fc = asf(uptr(g.st)[r_state].w);
        if (tid == 0) {     // back trace, last arc first
          gp<i32x4> path = (gp<i32x4>)(paths + (size_t)utt * path_cap);
          int t = cur_b + (int)(unsigned)bsel, len = 0, err = 0;
          while (true) {
            const i32x4 r = tok[t];
            if (r.x < 0) break;                     // the start token
            const i32x4 arc = uptr(g.arcs)[r.y];
            if (len < path_cap) {
              i32x4 o; o.x = uptr(g.arc_il)[r.y]; o.y = arc.w; o.z = arc.y; o.w = r.w;      // (the arc's cost: made with the token, make_tokens)
              path[len] = o;
            } else err = 1;
            len++;
            t = r.x;
          }
          sh.u0 = ((unsigned long long)(unsigned)err << 32) | (unsigned)len;
        }
        __syncthreads();
        r_len = (int)(unsigned)sh.u0;
        if (sh.u0 >> 32) { if (tid == 0) sh.error = FERR_PATH; }
        __syncthreads();
      }
    }
    FPHASE(9)
#ifdef VBH_FASTER_PROF
    if (tid == 0 && utt < 4) {
      printf("faster utt %d (%d frames, %lld tokens, %d frames on the slow winner path [tie %d bad %d]; order statistics %d, over 64 in the step %d, radix select %d; per frame: tokens %lld, candidates %lld, new states %lld; second map level in %d frames, %lld states each): cycles/frame gc %lld cutoff %lld clear %lld pass1 %lld pass2 %lld closure %lld winners %lld eps-winners %lld make: loads %lld stores %lld min %lld count %lld selects %lld barrier %lld; outside the frame loop %lld total\n",
             utt, T, (long long)(cur_b + cur_n), sh.n_slow, sh.u_tie, sh.u_bad, sh.n_sel, sh.n_sel_over, sh.n_sel_radix, sh.s_tok / T, sh.s_cand / T, sh.s_new / T, sh.s_gframes, sh.s_g / (sh.s_gframes ? sh.s_gframes : 1), tph[0] / T, tph[1] / T, tph[2] / T, tph[3] / T, tph[4] / T, tph[5] / T, tph[6] / T, tph[7] / T, tph[10] / T, tph[11] / T, tph[12] / T, tph[13] / T, tph[14] / T, tph[8] / T, tph[9]);
    }
    if (tid == 0) { for (int i = 0; i < 16; i++) tph[i] = 0; sh.n_sel = sh.n_sel_over = sh.n_sel_radix = 0; sh.s_cand = sh.s_new = sh.s_g = sh.s_tok = 0; sh.s_gframes = 0; }
#endif
    if (tid == 0) {
      FResult *r = &results[utt];
      r->ok = r_ok; r->error = sh.error; r->reached_final = r_final; r->num_frames = frames_done; r->path_len = r_len;
      r->final_state = r_state >= 0 ? (uptr(g.st)[r_state].x & STMASK) : r_state;      // (the graph's own state id)
      r->final_cost = r_fc; r->pad = sh.n_slow; r->n_tokens = cur_b + cur_n;      // (pad: frames that took the slow winner path)
    }
    __syncthreads();
  }
}

}  // namespace

// ============================================================================ host side
struct vbh_faster_decoder {
  vbh_ctx *ctx; const vbh_graph *graph; vbh_faster_cfg cfg;
  int max_frames, n_slots;
  FSlot proto{};
  std::vector<FSlot> h_slots;
  DevBuf<FSlot> d_slots;
  std::vector<void *> allocs;
  struct Out { std::vector<int32_t> ali, words, arc_il, arc_ol; std::vector<float> arc_gr, arc_ac; vbh_best_path view; };
  std::vector<Out> results;
};

static const char *ferr_name(int e) {
  switch (e) {
    case FERR_TOK: return "token history capacity exceeded even after garbage collection";
    case FERR_FRAME: return "more new tokens in one frame than the slot was sized for";
    case FERR_CAND: return "candidate list capacity exceeded";
    case FERR_FRONT: return "epsilon work-list capacity exceeded";
    case FERR_MAP: return "state map full (more new states in one frame than the slot was sized for)";
    case FERR_EPS: return "epsilon closure did not converge (epsilon cycle in the graph?)";
    case FERR_PATH: return "best path longer than the trace buffer";
    default: return "unknown";
  }
}

extern "C" {

void vbh_faster_cfg_default(vbh_faster_cfg *c) {
  // faster-decoder.h:38-43 (FasterDecoderOptions), acoustic scale as in VB/scr/kaldi_scr.h:152-157
  c->beam = 16.0f; c->max_active = INT_MAX; c->min_active = 20; c->beam_delta = 0.5f; c->acoustic_scale = 0.083333f;
  c->admission = VBH_ADMISSION_ORDERED;
}

int vbh_faster_decoder_create(vbh_ctx *ctx, const vbh_graph *graph, const vbh_faster_cfg *cfg, int32_t max_batch,
                              int32_t max_frames, vbh_faster_decoder **out) {
  return vbh_faster_decoder_create_ex(ctx, graph, cfg, max_batch, max_frames, 0, out);
}

int vbh_faster_decoder_create_ex(vbh_ctx *ctx, const vbh_graph *graph, const vbh_faster_cfg *cfg, int32_t max_batch,
                                 int32_t max_frames, int32_t max_states_per_utt, vbh_faster_decoder **out) {
  VBH_CHECK(ctx && graph && cfg && out, "vbh_faster_decoder_create: NULL argument");
  VBH_CHECK(cfg->beam > 0.0f && cfg->max_active > 1 && cfg->min_active >= 0 && cfg->min_active < cfg->max_active && cfg->beam_delta > 0.0f,
            "vbh_faster_decoder_create: invalid config (FasterDecoderOptions::Check)");
  VBH_CHECK(max_batch >= 1 && max_frames >= 0, "vbh_faster_decoder_create: bad batch/frames");
  if (vbh_use_device(ctx)) return -1;
  vbh_graph_view gv; vbh_graph_view_get(graph, &gv);
  vbh_faster_decoder *d = new vbh_faster_decoder();
  d->ctx = ctx; d->graph = graph; d->cfg = *cfg; d->max_frames = max_frames;
  d->n_slots = std::min<int>(max_batch, ctx->num_cus);
  FSlot p{};
  // new tokens per frame: bounded by the states; beyond the on-chip map the HBM level takes over
  const long long ma = std::min<long long>(cfg->max_active, gv.S);
  // (a frame early in an utterance can admit every arc leaving its tokens: all costs are still close)
  p.frame_cap = (int)std::min<long long>(std::max<long long>(gv.S, 1), std::max<long long>(16 * HCAP, std::min<long long>(16 * ma, 1 << 21)));
  // (alignment: the graph is the union of the utterances' own graphs and a search never leaves its utterance's part)
  if (max_states_per_utt > 0) p.frame_cap = std::min(p.frame_cap, std::max(max_states_per_utt, 64));
  if (const char *e = getenv("VBH_FASTER_FRAME_CAP")) { const int x = atoi(e); if (x > 0) p.frame_cap = std::min(p.frame_cap, x); }   // tests
  p.g_cap = 1;
  while (p.g_cap < 2 * p.frame_cap) p.g_cap <<= 1;
  // history: what garbage collection leaves (a few tokens per frame) + room for ~100 frames of new tokens
  // + the frame being built; collected whenever less than one frame's worth is free
  // (small graphs: every state every frame is the worst case and cheap to provide -- nothing to collect then)
  const long long worst = (long long)p.frame_cap * ((long long)max_frames + 2);
  const long long kept = 256LL * ((long long)max_frames + 2);
  // Round 5: with the reference's cutoffs (float32-rounded order statistics) a 30 s utterance of the bench's workload makes 7-9 M
  // tokens (2300-3000 per frame), and a collection costs ~50 M cycles (every current token walks its ancestors -- the trunk of the
  // history one dependent load at a time -- then the records move through the staging buffer): 9 % of an utterance's time at one or two
  // collections (-DVBH_FASTER_PROF).  The history is therefore sized so that an ordinary utterance needs none -- 10 M tokens at least;
  // it holds {previous token, arc, state} records and the collection's remap word only (the costs of the two frames that are ever read
  // live in FSlot::cost): 20 bytes per token, 200 MB per slot, 51 GB for 256 slots of a device that has 288 -- and collection stays
  // the answer to the extraordinary one (and to VBH_FASTER_HISTORY, the tests' way of forcing it).
  p.tok_cap = (int)std::min<long long>(std::min<long long>(worst, std::max<long long>(std::max<long long>(kept, 500LL * ((long long)max_frames + 2)) + 4LL * p.frame_cap, 10LL << 20)), 1LL << 28);
  if (const char *e = getenv("VBH_FASTER_HISTORY")) {     // tests: a small history forces garbage collections
    const long long x = atoll(e);
    if (x > 0) p.tok_cap = (int)std::min<long long>(p.tok_cap, std::max<long long>(x, 2LL * p.frame_cap));
  }
  p.stage_cap = (int)std::min<long long>(p.tok_cap, std::max<long long>(kept + 2LL * p.frame_cap, 1LL << 20));
  p.cand_cap = (int)std::min<long long>(std::max<long long>(gv.A, 1), std::min<long long>((ma + 256) * std::max(gv.max_degree, 1), 1LL << 22));
  p.cand_cap = std::max(p.cand_cap, std::min<int>((int)std::max<long long>(gv.A, 1), 2 * p.frame_cap));
  p.front_cap = 2 * p.frame_cap;
  d->proto = p;
  d->h_slots.assign(d->n_slots, p);
  const size_t ns = (size_t)d->n_slots;
#define A_(field, type, count)                                                                      \
  {                                                                                                 \
    type *base = nullptr;                                                                           \
    hipError_t e_ = hipMalloc(reinterpret_cast<void **>(&base), sizeof(type) * (size_t)(count) * ns); \
    if (e_ != hipSuccess) { vbh_set_error("vbh_faster_decoder_create: hipMalloc failed: %s", hipGetErrorString(e_)); vbh_faster_decoder_destroy(d); return -1; } \
    d->allocs.push_back(base);                                                                      \
    for (size_t i = 0; i < ns; i++) d->h_slots[i].field = base + i * (size_t)(count);               \
  }
  A_(tok, i32x4, p.tok_cap) A_(cost, double, 2 * (size_t)p.frame_cap) A_(win, unsigned long long, HCAP + p.g_cap) A_(wrec, i32x4, HCAP + p.g_cap)
  A_(c_w, double, p.cand_cap) A_(c_r, i32x4, p.cand_cap)
  A_(front[0], int, p.front_cap) A_(front[1], int, p.front_cap) A_(front[2], int, p.front_cap)
  A_(elist, int, p.frame_cap)
  A_(g_hs, unsigned, p.g_cap) A_(g_hc, unsigned long long, p.g_cap) A_(g_ht, int, p.g_cap) A_(g_list, int, p.frame_cap)
  A_(remap, int, p.tok_cap) A_(stage_tok, i32x4, p.stage_cap)
#undef A_
  if (d->d_slots.upload(d->h_slots.data(), d->h_slots.size(), ctx->stream)) { vbh_faster_decoder_destroy(d); return -1; }
  VBH_HIP(hipStreamSynchronize(ctx->stream));
  *out = d;
  return 0;
}

void vbh_faster_decoder_destroy(vbh_faster_decoder *d) {
  if (!d) return;
  (void)hipSetDevice(d->ctx->device);
  (void)hipStreamSynchronize(d->ctx->stream);
  for (void *p : d->allocs) (void)hipFree(p);
  delete d;
}

static int faster_decode_batch_impl(vbh_faster_decoder *d, const float *d_loglikes, int32_t num_pdfs, const int64_t *h_frame_offsets,
                                    int32_t n_utt, const int32_t *h_tid2pdf, int32_t num_tids, const int32_t *h_start_states, float beam, bool *holds_pdf) {
  VBH_CHECK(d && h_frame_offsets && h_tid2pdf, "vbh_faster_decode_batch: NULL argument");
  VBH_CHECK(n_utt >= 0 && num_pdfs > 0 && num_tids >= 0, "vbh_faster_decode_batch: bad sizes");
  if (vbh_use_device(d->ctx)) return -1;
  d->results.clear();
  if (n_utt == 0) return 0;
  vbh_graph_view gv; vbh_graph_view_get(d->graph, &gv);
  VBH_CHECK(gv.max_ilabel <= num_tids, "vbh_faster_decode_batch: graph uses transition-id %d but the model has %d (graph/model mismatch)", gv.max_ilabel, num_tids);
  for (int t = 1; t <= num_tids; t++)
    VBH_CHECK(h_tid2pdf[t] >= 0 && h_tid2pdf[t] < num_pdfs, "vbh_faster_decode_batch: tid2pdf[%d] = %d out of range", t, h_tid2pdf[t]);
  std::vector<FUtt> utts(n_utt);
  long long total = 0;
  for (int u = 0; u < n_utt; u++) {
    const long long T = h_frame_offsets[u + 1] - h_frame_offsets[u];
    VBH_CHECK(T >= 0 && T <= d->max_frames, "vbh_faster_decode_batch: utterance %d has %lld frames (decoder built for %d)", u, T, d->max_frames);
    if (h_start_states) VBH_CHECK(h_start_states[u] >= -1 && h_start_states[u] < gv.S, "vbh_faster_decode_batch: start state %d of utterance %d is not in the graph", h_start_states[u], u);
    utts[u] = FUtt{h_frame_offsets[u], (int)T, (h_start_states && h_start_states[u] >= 0) ? gv.h_pos[h_start_states[u]] : -1};
    total += T;
  }
  VBH_CHECK(total == 0 || d_loglikes, "vbh_faster_decode_batch: NULL loglikes");
  hipStream_t st = d->ctx->stream;
  if (vbh_graph_fill_pdf(d->graph, st, h_tid2pdf, num_tids)) return -1;
  *holds_pdf = true;
  FGraph fg{(int)gv.PKN, gv.start_pk, static_cast<const i32x4 *>(gv.pk), static_cast<const i32x4 *>(gv.pk), gv.pk_il};
  FCfg fc{beam > 0.0f ? beam : d->cfg.beam, d->cfg.beam_delta, d->cfg.acoustic_scale, d->cfg.max_active, d->cfg.min_active};
  std::vector<FResult> res(n_utt);
  d->results.resize(n_utt);
  const int path_cap = 4 * (d->max_frames + 2) + 64;
  DevBuf<FUtt> d_utts; DevBuf<int> d_counter; DevBuf<FResult> d_res; DevBuf<i32x4> d_paths;
  if (d_utts.upload(utts.data(), utts.size(), st) || d_counter.alloc(1) || d_res.alloc(n_utt) || d_paths.alloc((size_t)n_utt * path_cap)) return -1;
  VBH_HIP(hipMemsetAsync(d_counter.p, 0, sizeof(int), st));
  int adm = d->cfg.admission == VBH_ADMISSION_EXACT ? FADM_EXACT : FADM_ORDERED;
  if (const char *e = getenv("VBH_FASTER_ADMISSION")) adm = !strcmp(e, "exact") ? FADM_EXACT : !strcmp(e, "ordered") ? FADM_ORDERED : adm;      // (tests, A/B)
  if (adm == FADM_ORDERED)
    hipLaunchKernelGGL(faster_kernel<FADM_ORDERED>, dim3(std::min(d->n_slots, n_utt)), dim3(NT), 0, st, fg, fc, d->d_slots.p, d_loglikes, (int)num_pdfs,
                       d_utts.p, n_utt, d_counter.p, d_res.p, d_paths.p, path_cap);
  else
    hipLaunchKernelGGL(faster_kernel<FADM_EXACT>, dim3(std::min(d->n_slots, n_utt)), dim3(NT), 0, st, fg, fc, d->d_slots.p, d_loglikes, (int)num_pdfs,
                       d_utts.p, n_utt, d_counter.p, d_res.p, d_paths.p, path_cap);
  VBH_HIP(hipGetLastError());
  VBH_HIP(hipMemcpyAsync(res.data(), d_res.p, sizeof(FResult) * n_utt, hipMemcpyDeviceToHost, st));
  VBH_HIP(hipStreamSynchronize(st));
  // an utterance that outgrows the search workspace fails alone (ok = 0), as the reference fails one utterance at a time
  // (K/gmmbin/gmm-decode-faster.cc:150-178: "Did not successfully decode utterance"); the rest of the batch is delivered
  for (int u = 0; u < n_utt; u++) {
    if (res[u].error == FERR_NONE) continue;
    fprintf(stderr, "WARNING (vbh_faster_decode_batch) utterance %d failed after %d frames (%lld tokens so far; slot: %d tokens, %d per frame, %d candidates): %s; "
                    "it is reported with ok = 0\n", u, res[u].num_frames, res[u].n_tokens, d->proto.tok_cap, d->proto.frame_cap, d->proto.cand_cap,
            ferr_name(res[u].error));
    res[u].ok = 0; res[u].reached_final = 0; res[u].path_len = 0;
  }
  std::vector<std::vector<i32x4>> paths(n_utt);
  for (int u = 0; u < n_utt; u++) {
    paths[u].resize(res[u].path_len);
    if (res[u].path_len)
      VBH_HIP(hipMemcpyAsync(paths[u].data(), d_paths.p + (size_t)u * path_cap, sizeof(i32x4) * res[u].path_len, hipMemcpyDeviceToHost, st));
  }
  VBH_HIP(hipStreamSynchronize(st));
  for (int u = 0; u < n_utt; u++) {
    vbh_faster_decoder::Out &o = d->results[u];
    // GetBestPath (:87-147): arcs from the start, (graph, acoustic) = (weight, tot - weight) summed in path order
    float t1 = 0.0f, t2 = 0.0f;
    o.ali.clear(); o.words.clear();        // (a decoder is used for batch after batch)
    o.arc_il.clear(); o.arc_ol.clear(); o.arc_gr.clear(); o.arc_ac.clear();
    for (int i = res[u].path_len - 1; i >= 0; i--) {
      const i32x4 a = paths[u][i];
      const int gb = a.z, tb = a.w;
      float graph_cost, tot_cost;
      memcpy(&graph_cost, &gb, 4); memcpy(&tot_cost, &tb, 4);
      const float ac_cost = tot_cost - graph_cost;
      t1 = graph_cost + t1; t2 = ac_cost + t2;
      if (a.x != 0) o.ali.push_back(a.x);
      if (a.y != 0) o.words.push_back(a.y);
      o.arc_il.push_back(a.x); o.arc_ol.push_back(a.y); o.arc_gr.push_back(graph_cost); o.arc_ac.push_back(ac_cost);
    }
    if (res[u].ok && res[u].reached_final) t1 = res[u].final_cost + t1;
    vbh_best_path &v = o.view;
    memset(&v, 0, sizeof(v));
    v.ok = res[u].ok; v.reached_final = res[u].reached_final; v.num_frames = res[u].num_frames;
    v.num_ali = (int)o.ali.size(); v.ali = o.ali.data(); v.num_words = (int)o.words.size(); v.words = o.words.data();
    v.best_graph = res[u].ok ? t1 : 0.0f; v.best_acoustic = res[u].ok ? t2 : 0.0f;
    v.n_tokens = res[u].n_tokens;
    v.num_arcs = (int)o.arc_il.size(); v.arc_ilabel = o.arc_il.data(); v.arc_olabel = o.arc_ol.data();
    v.arc_graph = o.arc_gr.data(); v.arc_acoustic = o.arc_ac.data();
    v.final_graph = (res[u].ok && res[u].reached_final) ? res[u].final_cost : 0.0f;
  }
  return 0;
}

int vbh_faster_decode_batch(vbh_faster_decoder *d, const float *d_loglikes, int32_t num_pdfs, const int64_t *h_frame_offsets,
                            int32_t n_utt, const int32_t *h_tid2pdf, int32_t num_tids) {
  return vbh_faster_decode_batch_ex(d, d_loglikes, num_pdfs, h_frame_offsets, n_utt, h_tid2pdf, num_tids, nullptr, 0.0f);
}

int vbh_faster_decode_batch_ex(vbh_faster_decoder *d, const float *d_loglikes, int32_t num_pdfs, const int64_t *h_frame_offsets,
                               int32_t n_utt, const int32_t *h_tid2pdf, int32_t num_tids, const int32_t *h_start_states, float beam) {
  bool holds_pdf = false;
  const int rc = faster_decode_batch_impl(d, d_loglikes, num_pdfs, h_frame_offsets, n_utt, h_tid2pdf, num_tids, h_start_states, beam, &holds_pdf);
  if (holds_pdf) vbh_graph_release_pdf(d->graph);
  return rc;
}

int vbh_faster_decoder_result(vbh_faster_decoder *d, int32_t utt, vbh_best_path *out) {
  VBH_CHECK(d && out, "vbh_faster_decoder_result: NULL argument");
  VBH_CHECK(utt >= 0 && utt < (int)d->results.size(), "vbh_faster_decoder_result: utterance %d out of range", utt);
  *out = d->results[utt].view;
  return 0;
}

}  // extern "C"
