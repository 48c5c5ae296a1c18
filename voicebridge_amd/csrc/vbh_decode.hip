// vbh_decode.hip -- frame-synchronous token-passing lattice search on gfx950.
//
// Replaces (paths relative to /root/reference/kaldi-master/src/):
//   decoder/lattice-faster-decoder.cc:56-96    InitDecoding / Decode (frame loop)
//   :618-685  GetCutoff            :687-777  ProcessEmitting     :796-858  ProcessNonemitting
//   :259-295  FindOrAddToken (+ util/hash-list-inl.h)            :300-371  PruneForwardLinks
//   :376-458  PruneForwardLinksFinal  :477-530 PruneTokensForFrame / PruneActiveTokens
//   :532-572  ComputeFinalCosts    :600-615  FinalizeDecoding    :110-192  GetRawLattice (content)
//   gmm/decodable-am-diag-gmm.h:142-145        acoustic_scale * loglike(frame, tid2pdf[tid])
//
// Execution model.  One 1024-thread workgroup owns one utterance for its whole life and
// walks its frames; a batch runs as a persistent grid that pulls utterances from a counter,
// so there is no inter-workgroup communication at all.  Per workgroup ("slot") HBM holds
//   * three state-indexed arrays (key, claim, tokid) that replace the reference's per-frame
//     hash: key[s] = (epoch | order-preserving bits of the best cost reaching s this frame)
//     updated with 64-bit atomicMin, claim[s] = (epoch | phase) taken with one atomicCAS by the
//     thread that creates the token, tokid[s] = token index.  The epoch (frame counter) in the
//     high word makes entries of older frames lose every comparison, so nothing is cleared;
//   * a raw window (W) of tokens and forward links created since the last pruning, and a
//     history (H) holding what survived earlier prunings, with stable indices.
// Every `prune_interval` frames the workgroup recomputes extra_costs backwards (link-parallel
// min with atomics, iterated per frame to the fixed point when epsilon links exist), stops
// at the first already-pruned frame whose values did not move, and compacts the survivors
// of W into H.  The reference's final result does not depend on when intermediate pruning
// happens (its final pass recomputes every frame with delta = 0), so this schedule yields
// the same lattice.
//
// Semantics.  Costs use exactly the reference's float32 expressions and association order
// (this file is compiled with -ffp-contract=off).  The one deliberate difference is the
// admission rule of ProcessEmitting: the reference tightens next_cutoff while it walks its
// hash list, so which over-the-cutoff arcs it admits depends on hash order (and on the
// utterances decoded before by the same object); here next_cutoff takes its final value
// (frame minimum + adaptive beam) before anything is admitted.  That is the
// `exact_cutoff` mode of oracle/vb_oracle_decode.c, against which this kernel is bit-exact;
// tests/test_oracle_decode.py pins how that mode relates to the reference (same best path
// and weights; lattice = reference lattice minus the order-dependent extra arcs).
//
// Per-frame algorithmic traffic: N_tok(20+16) + N_arc(16+4) + N_new 16 +
// N_link 24 bytes of random access; the counters are reported per utterance.
#include "vbh_internal.h"

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>

namespace {

#ifndef VBH_DECODE_NT
#define VBH_DECODE_NT 1024   // measured on MI355X (profiles/README.md): 256/512/1024 threads per utterance all
                             // saturate at the same ~0.65 M frames/s once >= 256 utterances are in flight
#endif
constexpr int NT = VBH_DECODE_NT;    // threads per workgroup = per utterance; 1024/NT workgroups share a CU
constexpr int NW = NT / 64;
constexpr int WBIT = 0x40000000;            // token id lives in the raw window
constexpr unsigned INF_BITS = 0x7F800000u;  // +inf as float bits
constexpr int MAX_EPS_ITERS = 4096;
constexpr int MAX_WIN_FRAMES = 512;         // frames between two compactions (prune_interval + 1 <= this)

struct DevGraph {
  int S, start;
  const float *final_cost;   // [S], +inf = not final
  const int *arc_off;        // [S+1]
  const int *nieps;          // [S] number of input-epsilon arcs
  const vbh_arc *arcs;       // [A]
  const int *arc_pdf;        // [A] tid2pdf[ilabel], -1 for input-epsilon arcs (filled per batch)
};

struct DecCfg {
  float beam, lattice_beam, beam_delta, acwt;
  int max_active, min_active, prune_interval;
};

// per-slot workspace (device pointers)
struct Slot {
  unsigned long long *key;                 // [S] (epoch | order-preserving cost bits)
  int *tokid;                              // [S] window index of the state's token in the current epoch
  // raw window
  int *w_state; float *w_cost; unsigned *w_extra, *w_tmp; int *w_remap;
  int *wl_src, *wl_dst, *wl_il, *wl_ol; float *wl_gr, *wl_ac;
  // history
  int *h_state; float *h_cost; unsigned *h_extra, *h_tmp; int *h_remap;
  int *hl_src, *hl_dst, *hl_il, *hl_ol; float *hl_gr, *hl_ac;
  // per frame_plus_one ranges: tokens, emitting links from the frame, epsilon links within it
  int *tok_b, *tok_e, *el_b, *el_e, *pl_b, *pl_e;
  unsigned char *tok_w, *el_w, *pl_w;      // 1 = range refers to the window
  float *cost_offset;                      // [max_frames]
  // candidates of the current frame (pass 1 -> passes 2, 3)
  float *c_tot, *c_ac; int *c_arc, *c_src, *c_dst;
  int *front[3];                           // epsilon work lists (state ids), rotated
  int *state;                              // persistent: [0] epoch base
  int w_tok_cap, w_link_cap, h_tok_cap, h_link_cap, front_cap, cand_cap;
};

struct UttDesc { long long row0; int T; int pad; };

struct UttResult {   // written by the kernel, one per utterance
  int ok, error, reached_final, num_frames;
  float final_relative_cost, final_best_cost;
  int final_costs_empty;
  int n_tok, n_link;                 // alive entries written to the pool
  long long tok_off, link_off;       // offsets into the pool
  long long n_tokens_expanded, n_arcs_expanded, n_tokens_created, n_links_created;
  int n_max_bound, n_min_bound;
  long long cost_off;                // offset of this utterance's cost_offsets in the pool
  long long t_phase[8];              // shader-clock cycles per phase (thread 0): cutoff, pass1, pass2+3, eps, prune, compact, final, other
};

struct Pool {   // dense per-batch output
  int *tok_frame, *tok_state; float *tok_cost, *tok_extra, *tok_final;
  int *link_src, *link_dst, *link_il, *link_ol; float *link_gr, *link_ac;
  float *cost_offsets;
  unsigned long long *counters;   // [0] tokens used, [1] links used, [2] cost offsets used
  long long tok_cap, link_cap, cost_cap;
};

enum { ERR_NONE = 0, ERR_W_TOK = 1, ERR_W_LINK = 2, ERR_H_TOK = 3, ERR_H_LINK = 4, ERR_FRONT = 5,
       ERR_EPS_LOOP = 6, ERR_POOL = 7, ERR_CAND = 8 };

// order-preserving map float -> uint32
__device__ __forceinline__ unsigned f2o(float f) {
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float o2f(unsigned o) {
  unsigned u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
  return __uint_as_float(u);
}
__device__ __forceinline__ unsigned long long ld_u64(const unsigned long long *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int ld_i32(const int *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_i32(int *p, int v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// make an LDS-loaded (per-lane) pointer wave-uniform again so that it lives in SGPRs
template <class T>
__device__ __forceinline__ T *uptr(T *p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<T *>(((unsigned long long)hi << 32) | lo);
}
// One LDS atomic per wave instead of one per lane: the active lanes that want a slot elect a
// leader, which reserves popcount slots; every lane gets base + its rank.  (Same-address LDS
// atomics serialise per lane; with thousands of allocations per frame they dominated.)
__device__ __forceinline__ int wave_alloc(int *counter, bool want) {
  const unsigned long long m = __ballot(want);
  if (m == 0) return -1;
  const int lane = threadIdx.x & 63;
  const int leader = __ffsll((long long)m) - 1;
  int base = 0;
  if (lane == leader) base = atomicAdd(counter, __popcll(m));
  base = __shfl(base, leader, 64);
  return want ? base + __popcll(m & ((1ull << lane) - 1ull)) : -1;
}
__device__ __forceinline__ long long clk() { return (long long)__builtin_amdgcn_s_memtime(); }

struct Shared {
  unsigned long long red[2][NW];   // reduction scratch, alternated so that one barrier per reduction suffices
  unsigned hist[256];
  int w_ntok, w_nlink, h_ntok, h_nlink;
  int n_cand;
  int fcnt[3];                     // epsilon work-list fill counts, rotated with Slot::front
  int changed2[2];                 // "something changed" flags, alternated per use (see prune_frame)
  int error, flag;
  unsigned sel_prefix, sel_mask; int sel_k;
  float f0;
  unsigned long long u0;
  int n_max_bound, n_min_bound, n_visited;
  int utt;
  // the block appended to H by the previous compaction (frames (tail_f0, tail_f1]); it is
  // re-compacted in place at the next one, once those frames have been pruned with look-ahead
  int tail_tok_b, tail_link_b, tail_f0, tail_f1;
  long long tph[8];                // phase cycle counters (thread 0)
  int fr_cnt[3 * MAX_WIN_FRAMES];  // compaction: survivors per frame (tokens, emitting links, epsilon links)
  int fr_base[3 * MAX_WIN_FRAMES];
};

// block reductions with ONE barrier each: partials go to red[par], par flips on every call
__device__ __forceinline__ unsigned long long block_min_u64(unsigned long long v, Shared &sh, int &par) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    unsigned long long w = __shfl_xor(v, o, 64);
    v = w < v ? w : v;
  }
  if ((threadIdx.x & 63) == 0) sh.red[par][threadIdx.x >> 6] = v;
  __syncthreads();
  unsigned long long r = sh.red[par][0];
#pragma unroll
  for (int i = 1; i < NW; i++) r = sh.red[par][i] < r ? sh.red[par][i] : r;
  par ^= 1;
  return r;
}
__device__ __forceinline__ unsigned long long block_sum_u64(unsigned long long v, Shared &sh, int &par) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) sh.red[par][threadIdx.x >> 6] = v;
  __syncthreads();
  unsigned long long r = 0;
#pragma unroll
  for (int i = 0; i < NW; i++) r += sh.red[par][i];
  par ^= 1;
  return r;
}
__device__ __forceinline__ float block_min_f(float v, Shared &sh, int &par) {
  return o2f((unsigned)block_min_u64((unsigned long long)f2o(v), sh, par));
}

// k-th smallest (0-based) of cost[0..n): exact radix select on the order-preserving keys
// (the std::nth_element value in GetCutoff, lattice-faster-decoder.cc:654-676)
__device__ float select_kth(const float *cost, int n, int k, Shared &sh) {
  const int tid = threadIdx.x;
  __syncthreads();
  if (tid == 0) { sh.sel_prefix = 0; sh.sel_mask = 0; sh.sel_k = k; }
  for (int shift = 24; shift >= 0; shift -= 8) {
    for (int i = tid; i < 256; i += NT) sh.hist[i] = 0;
    __syncthreads();
    const unsigned prefix = sh.sel_prefix, mask = sh.sel_mask;
    for (int i = tid; i < n; i += NT) {
      const unsigned key = f2o(cost[i]);
      if ((key & mask) == prefix) atomicAdd(&sh.hist[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      int kk = sh.sel_k, b = 0;
      for (; b < 255; b++) {
        if (kk < (int)sh.hist[b]) break;
        kk -= (int)sh.hist[b];
      }
      sh.sel_k = kk;
      sh.sel_prefix = prefix | ((unsigned)b << shift);
      sh.sel_mask = mask | (255u << shift);
    }
    __syncthreads();
  }
  return o2f(sh.sel_prefix);
}

__device__ __forceinline__ float tok_cost(const Slot &s, int id) { return (id & WBIT) ? s.w_cost[id ^ WBIT] : s.h_cost[id]; }
__device__ __forceinline__ unsigned tok_extra(const Slot &s, int id) { return (id & WBIT) ? s.w_extra[id ^ WBIT] : s.h_extra[id]; }

// evaluate the forward links [lb, le) of one frame: excise those over the lattice beam,
// atomicMin the rest into tmp[src] (lattice-faster-decoder.cc:331-356)
__device__ __forceinline__ void eval_links(const Slot &s, bool lw, int lb, int le_, const float *tcost, unsigned *tmp,
                                           float lattice_beam) {
  if (le_ <= lb) return;
  int *lsrc = uptr(lw ? s.wl_src : s.hl_src);
  const int *ldst = uptr(lw ? s.wl_dst : s.hl_dst);
  const float *lgr = uptr(lw ? s.wl_gr : s.hl_gr), *lac = uptr(lw ? s.wl_ac : s.hl_ac);
  const float *wc = uptr(s.w_cost), *hc = uptr(s.h_cost);
  const unsigned *we = uptr(s.w_extra), *he = uptr(s.h_extra);
  tcost = uptr(tcost); tmp = uptr(tmp);
  for (int l = lb + threadIdx.x; l < le_; l += NT) {
    const int src = lsrc[l];
    if (src < 0) continue;   // already excised
    const int dst = ldst[l];
    const float nx = __uint_as_float((dst & WBIT) ? we[dst ^ WBIT] : he[dst]);
    const float dc = (dst & WBIT) ? wc[dst ^ WBIT] : hc[dst];
    // next_tok->extra_cost + ((tok->tot_cost + link->acoustic_cost + link->graph_cost) - next_tok->tot_cost)
    float le = nx + (((tcost[src & ~WBIT] + lac[l]) + lgr[l]) - dc);
    if (le > lattice_beam) { lsrc[l] = -1; continue; }
    if (le < 0.0f) le = 0.0f;
    atomicMin(&tmp[src & ~WBIT], __float_as_uint(le));
  }
}

// PruneForwardLinks for one frame (lattice-faster-decoder.cc:300-371), evaluated to its fixed
// point.  tmp[] of the frame's tokens must hold +inf on entry and holds +inf again on exit.
// Returns (uniformly) whether any token's extra_cost changed.
__device__ bool prune_frame(const Slot &s, int f, float lattice_beam, float delta, Shared &sh, int &pp) {
  const int tid = threadIdx.x;
  const int tb = s.tok_b[f], te = s.tok_e[f];
  const bool tw = s.tok_w[f];
  unsigned *extra = tw ? s.w_extra : s.h_extra, *tmp = tw ? s.w_tmp : s.h_tmp;
  const float *tcost = tw ? s.w_cost : s.h_cost;
  const int eb = s.el_b[f], ee = s.el_e[f], pb = s.pl_b[f], pe = s.pl_e[f];
  const bool ew = s.el_w[f], pw = s.pl_w[f];
  bool frame_changed = false;
  for (int iter = 0; iter < MAX_EPS_ITERS; iter++) {
    eval_links(s, ew, eb, ee, tcost, tmp, lattice_beam);
    eval_links(s, pw, pb, pe, tcost, tmp, lattice_beam);
    __syncthreads();
    // flag[pp] is written below and read after the next barrier; flag[pp^1] (read by every
    // thread before it arrived at the barrier above) is cleared for the following use
    if (tid == 0) sh.changed2[pp ^ 1] = 0;
    int ch = 0;
    {
      unsigned *tmp_u = uptr(tmp), *extra_u = uptr(extra);
      for (int t = tb + tid; t < te; t += NT) {
        const unsigned v = tmp_u[t], o = extra_u[t];
        tmp_u[t] = INF_BITS;
        if (v != o) {
          extra_u[t] = v;
          // "changed" as the reference counts it (:358): |new - old| > delta; inf vs inf is no change
          if (fabsf(__uint_as_float(v) - __uint_as_float(o)) > delta) ch = 1;
        }
      }
    }
    if (ch) sh.changed2[pp] = 1;
    __syncthreads();
    const bool any = sh.changed2[pp] != 0;
    pp ^= 1;
    frame_changed |= any;
    if (!any || pe == pb) break;
  }
  return frame_changed;
}

// backward pass from frame `from` down; frames >= first_unpruned are always processed, older
// ones until a frame's extra_costs move by no more than delta (PruneActiveTokens :503-530).
// extra_costs only grow as the frontier advances, so values left stale by the early stop are
// lower bounds and pruning with them never removes anything the final pass (delta = 0, every
// frame revisited) would keep.
__device__ void prune_backward(const Slot &s, int from, int first_unpruned, int stop_at, float lattice_beam, float delta,
                               Shared &sh, int &pp) {
  int visited = 0;
  for (int f = from; f >= 0 && f >= stop_at; f--) {
    const bool ch = prune_frame(s, f, lattice_beam, delta, sh, pp);
    visited++;
    if (!ch && f < first_unpruned) break;
  }
  if (threadIdx.x == 0) sh.n_visited += visited;   // diagnostic
}

// Re-compact the block that the previous compaction appended to H (frames (tail_f0, tail_f1]
// and their links).  When it was appended its youngest frames had no look-ahead, so most of
// their tokens and links survived; by now they have been pruned again and nearly all of them are
// dead.  Survivors are staged in the free space behind the block and copied back to its start;
// h_remap[] maps old to new token indices (-1 = dead) for this block.  Everything that refers to
// these tokens is either inside the block or still in the raw window.  Returns false on overflow.
__device__ bool recompact_tail(const Slot &s, Shared &sh) {
  const int tid = threadIdx.x;
  const int tf0 = sh.tail_f0, tf1 = sh.tail_f1;
  if (tf1 < 0) return true;                       // nothing appended yet
  const int fa = tf0 < 0 ? 0 : tf0;               // link frames fa..tf1 (emitting: fa..tf1-1), token frames tf0+1..tf1
  const int nf = tf1 - fa + 1;
  const int tail_tb = sh.tail_tok_b, tail_lb = sh.tail_link_b;
  const int old_nt = sh.h_ntok, old_nl = sh.h_nlink;
  for (int i = tid; i < 3 * nf; i += NT) sh.fr_cnt[i] = 0;
  __syncthreads();
  const unsigned *he = uptr(s.h_extra);
  const int *hs = uptr(s.hl_src), *hd = uptr(s.hl_dst);
  for (int k = 0; k < nf; k++) {
    const int f = fa + k;
    if (f > tf0) {
      int c = 0;
      for (int t = s.tok_b[f] + tid; t < s.tok_e[f]; t += NT) c += he[t] != INF_BITS;
      if (c) atomicAdd(&sh.fr_cnt[k], c);
    }
    for (int pass = 0; pass < 2; pass++) {
      if (pass == 0 && f == tf1) continue;        // emitting links of the frontier frame are still in W
      if (pass == 1 && f <= tf0) continue;
      const int lb = pass ? s.pl_b[f] : s.el_b[f], le_ = pass ? s.pl_e[f] : s.el_e[f];
      int c = 0;
      for (int l = lb + tid; l < le_; l += NT) {
        const int src = hs[l];
        c += src >= 0 && he[src] != INF_BITS && he[hd[l]] != INF_BITS;
      }
      if (c) atomicAdd(&sh.fr_cnt[(1 + pass) * nf + k], c);
    }
  }
  __syncthreads();
  if (tid == 0) {
    int tb = tail_tb, lb = tail_lb;
    for (int k = 0; k < nf; k++) { sh.fr_base[k] = tb; tb += sh.fr_cnt[k]; }
    for (int k = 0; k < nf; k++) {
      sh.fr_base[nf + k] = lb; lb += sh.fr_cnt[nf + k];
      sh.fr_base[2 * nf + k] = lb; lb += sh.fr_cnt[2 * nf + k];
    }
    // staging area = [old_nt, old_nt + alive) and [old_nl, old_nl + alive)
    if (old_nt + (tb - tail_tb) > s.h_tok_cap) sh.error = ERR_H_TOK;
    if (old_nl + (lb - tail_lb) > s.h_link_cap) sh.error = ERR_H_LINK;
    sh.h_ntok = tb; sh.h_nlink = lb;             // new ends of the history
  }
  __syncthreads();
  if (sh.error != ERR_NONE) return false;
  const int new_nt = sh.h_ntok, new_nl = sh.h_nlink;
  for (int i = tid; i < 3 * nf; i += NT) sh.fr_cnt[i] = 0;
  __syncthreads();
  const int stage_t = old_nt - tail_tb, stage_l = old_nl - tail_lb;   // staging offset relative to the final position
  // ---- tokens -> staging, remap table
  for (int k = 0; k < nf; k++) {
    const int f = fa + k;
    if (f <= tf0) continue;
    for (int t = s.tok_b[f] + tid; t < s.tok_e[f]; t += NT) {
      int nid = -1;
      if (s.h_extra[t] != INF_BITS) {
        nid = sh.fr_base[k] + atomicAdd(&sh.fr_cnt[k], 1);
        const int q = nid + stage_t;
        s.h_state[q] = s.h_state[t]; s.h_cost[q] = s.h_cost[t]; s.h_extra[q] = s.h_extra[t];
      }
      s.h_remap[t] = nid;
    }
  }
  __syncthreads();
  // ---- links -> staging (end points remapped when they lie in the block)
  for (int k = 0; k < nf; k++) {
    const int f = fa + k;
    for (int pass = 0; pass < 2; pass++) {
      if (pass == 0 && f == tf1) continue;
      if (pass == 1 && f <= tf0) continue;
      const int lb = pass ? s.pl_b[f] : s.el_b[f], le_ = pass ? s.pl_e[f] : s.el_e[f];
      const int slot = (1 + pass) * nf + k;
      for (int l = lb + tid; l < le_; l += NT) {
        int src = s.hl_src[l];
        if (src < 0) continue;
        int dst = s.hl_dst[l];
        if (src >= tail_tb) src = s.h_remap[src];
        if (dst >= tail_tb) dst = s.h_remap[dst];
        if (src < 0 || dst < 0) continue;
        if (s.h_extra[s.hl_src[l]] == INF_BITS) continue;   // source outside the block but dead
        const int q = sh.fr_base[slot] + atomicAdd(&sh.fr_cnt[slot], 1) + stage_l;
        s.hl_src[q] = src; s.hl_dst[q] = dst; s.hl_il[q] = s.hl_il[l]; s.hl_ol[q] = s.hl_ol[l];
        s.hl_gr[q] = s.hl_gr[l]; s.hl_ac[q] = s.hl_ac[l];
      }
    }
  }
  __syncthreads();
  // ---- staging -> final position (the two regions never overlap: staging starts at the old end)
  for (int i = tail_tb + tid; i < new_nt; i += NT) {
    const int q = i + stage_t;
    s.h_state[i] = s.h_state[q]; s.h_cost[i] = s.h_cost[q]; s.h_extra[i] = s.h_extra[q]; s.h_tmp[i] = INF_BITS;
  }
  for (int i = tail_lb + tid; i < new_nl; i += NT) {
    const int q = i + stage_l;
    s.hl_src[i] = s.hl_src[q]; s.hl_dst[i] = s.hl_dst[q]; s.hl_il[i] = s.hl_il[q]; s.hl_ol[i] = s.hl_ol[q];
    s.hl_gr[i] = s.hl_gr[q]; s.hl_ac[i] = s.hl_ac[q];
  }
  if (tid == 0) {
    for (int k = 0; k < nf; k++) {
      const int f = fa + k;
      if (f > tf0) { s.tok_b[f] = sh.fr_base[k]; s.tok_e[f] = sh.fr_base[k] + sh.fr_cnt[k]; }
      if (f != tf1) { s.el_b[f] = sh.fr_base[nf + k]; s.el_e[f] = sh.fr_base[nf + k] + sh.fr_cnt[nf + k]; }
      if (f > tf0) { s.pl_b[f] = sh.fr_base[2 * nf + k]; s.pl_e[f] = sh.fr_base[2 * nf + k] + sh.fr_cnt[2 * nf + k]; }
    }
  }
  __threadfence_block();
  __syncthreads();
  return true;
}

// Move the survivors of the raw window into the history: tokens of frames (prev_front, front],
// emitting links from frames [prev_front, front-1], epsilon links within (prev_front, front].
// Survivors are counted per frame first so that every frame stays contiguous in H with a
// handful of barriers in total.  Returns false (uniformly) on overflow.
__device__ bool compact_window(const Slot &s, int prev_front, int front, Shared &sh) {
  const int tid = threadIdx.x;
  const int old_tail_tb = sh.tail_tok_b;
  const bool had_tail = sh.tail_f1 >= 0;
  if (!recompact_tail(s, sh)) return false;
  const int f0 = prev_front < 0 ? 0 : prev_front;
  const int nf = front - f0 + 1;     // frames f0..front  (<= MAX_WIN_FRAMES, checked on the host)
  for (int i = tid; i < 3 * nf; i += NT) sh.fr_cnt[i] = 0;
  __syncthreads();
  // ---- count
  for (int k = 0; k < nf; k++) {
    const int f = f0 + k;
    if (s.tok_w[f]) {
      int c = 0;
      for (int t = s.tok_b[f] + tid; t < s.tok_e[f]; t += NT) c += s.w_extra[t] != INF_BITS;
      if (c) atomicAdd(&sh.fr_cnt[k], c);
    }
    for (int pass = 0; pass < 2; pass++) {
      if (!(pass ? s.pl_w[f] : s.el_w[f])) continue;
      const int lb = pass ? s.pl_b[f] : s.el_b[f], le_ = pass ? s.pl_e[f] : s.el_e[f];
      int c = 0;
      for (int l = lb + tid; l < le_; l += NT) {
        const int src = s.wl_src[l];
        if (src < 0) continue;
        const int dst = s.wl_dst[l];
        const bool sa = (src & WBIT) ? s.w_extra[src ^ WBIT] != INF_BITS
                                     : (had_tail && src >= old_tail_tb ? s.h_remap[src] >= 0 : true);
        const bool da = (dst & WBIT) ? s.w_extra[dst ^ WBIT] != INF_BITS : true;
        c += sa && da;
      }
      if (c) atomicAdd(&sh.fr_cnt[(1 + pass) * nf + k], c);
    }
  }
  __syncthreads();
  if (tid == 0) {
    int tb = sh.h_ntok, lb = sh.h_nlink;
    for (int k = 0; k < nf; k++) { sh.fr_base[k] = tb; tb += sh.fr_cnt[k]; }
    for (int k = 0; k < nf; k++) {
      sh.fr_base[nf + k] = lb; lb += sh.fr_cnt[nf + k];
      sh.fr_base[2 * nf + k] = lb; lb += sh.fr_cnt[2 * nf + k];
    }
    if (tb > s.h_tok_cap) sh.error = ERR_H_TOK;
    if (lb > s.h_link_cap) sh.error = ERR_H_LINK;
    sh.h_ntok = tb; sh.h_nlink = lb;
  }
  __syncthreads();
  if (sh.error != ERR_NONE) return false;
  for (int i = tid; i < 3 * nf; i += NT) sh.fr_cnt[i] = 0;   // reused as fill counters
  __syncthreads();
  // ---- move tokens
  for (int k = 0; k < nf; k++) {
    const int f = f0 + k;
    if (!s.tok_w[f]) continue;
    for (int t = s.tok_b[f] + tid; t < s.tok_e[f]; t += NT) {
      int nid = -1;
      if (s.w_extra[t] != INF_BITS) {
        nid = sh.fr_base[k] + atomicAdd(&sh.fr_cnt[k], 1);
        s.h_state[nid] = s.w_state[t]; s.h_cost[nid] = s.w_cost[t]; s.h_extra[nid] = s.w_extra[t];
        s.h_tmp[nid] = INF_BITS;
      }
      s.w_remap[t] = nid;
    }
  }
  __syncthreads();
  // ---- move links
  for (int k = 0; k < nf; k++) {
    const int f = f0 + k;
    for (int pass = 0; pass < 2; pass++) {
      if (!(pass ? s.pl_w[f] : s.el_w[f])) continue;
      const int lb = pass ? s.pl_b[f] : s.el_b[f], le_ = pass ? s.pl_e[f] : s.el_e[f];
      const int slot = (1 + pass) * nf + k;
      for (int l = lb + tid; l < le_; l += NT) {
        int src = s.wl_src[l];
        if (src < 0) continue;
        int dst = s.wl_dst[l];
        if (src & WBIT) src = s.w_remap[src ^ WBIT];
        else if (had_tail && src >= old_tail_tb) src = s.h_remap[src];   // a token of the re-compacted block
        if (dst & WBIT) dst = s.w_remap[dst ^ WBIT];
        if (src < 0 || dst < 0) continue;
        const int nid = sh.fr_base[slot] + atomicAdd(&sh.fr_cnt[slot], 1);
        s.hl_src[nid] = src; s.hl_dst[nid] = dst; s.hl_il[nid] = s.wl_il[l]; s.hl_ol[nid] = s.wl_ol[l];
        s.hl_gr[nid] = s.wl_gr[l]; s.hl_ac[nid] = s.wl_ac[l];
      }
    }
  }
  __syncthreads();
  if (tid == 0) {
    for (int k = 0; k < nf; k++) {
      const int f = f0 + k;
      if (s.tok_w[f]) { s.tok_b[f] = sh.fr_base[k]; s.tok_e[f] = sh.fr_base[k] + sh.fr_cnt[k]; s.tok_w[f] = 0; }
      if (s.el_w[f]) { s.el_b[f] = sh.fr_base[nf + k]; s.el_e[f] = sh.fr_base[nf + k] + sh.fr_cnt[nf + k]; s.el_w[f] = 0; }
      if (s.pl_w[f]) { s.pl_b[f] = sh.fr_base[2 * nf + k]; s.pl_e[f] = sh.fr_base[2 * nf + k] + sh.fr_cnt[2 * nf + k]; s.pl_w[f] = 0; }
    }
    sh.w_ntok = 0; sh.w_nlink = 0;
    sh.tail_tok_b = sh.fr_base[0]; sh.tail_link_b = sh.fr_base[nf];
    sh.tail_f0 = prev_front; sh.tail_f1 = front;
  }
  __threadfence_block();
  __syncthreads();
  return true;
}

// ---------------------------------------------------------------------------------------
// ProcessNonemitting (lattice-faster-decoder.cc:796-858) for the tokens of frame fpo, which
// occupy window indices [nb, sh.w_ntok).  Work lists hold STATE ids: list `cur` (count
// sh.fcnt[cur]) was filled by the caller with the new states that have input-epsilon arcs.
// Iterates until no cost changes, then stores the final costs and writes the epsilon links.
// Returns the first link index of the frame's epsilon links; on return a barrier has made all
// writes visible.
__device__ int process_nonemitting(const DevGraph &g, const Slot &s, int nb, float cutoff, unsigned ep,
                                   int cur, Shared &sh) {
  const int tid = threadIdx.x;
  const unsigned long long EPK = (unsigned long long)ep << 32;
  for (int iter = 0;; iter++) {
    const int ncur = sh.fcnt[cur] < s.front_cap ? sh.fcnt[cur] : s.front_cap;
    if (ncur == 0) break;
    if (iter >= MAX_EPS_ITERS) { if (tid == 0) sh.error = ERR_EPS_LOOP; break; }
    const int nxt = cur == 2 ? 0 : cur + 1, spare = nxt == 2 ? 0 : nxt + 1;
    if (tid == 0) sh.fcnt[spare] = 0;      // nobody touches the spare list during this iteration
    const int *fcur = uptr(s.front[cur]);
    int *fnext = uptr(s.front[nxt]);
    unsigned long long *key = uptr(s.key);
    int *tokid = uptr(s.tokid), *w_state = uptr(s.w_state);
    unsigned *w_extra = uptr(s.w_extra), *w_tmp = uptr(s.w_tmp);
    const int w_tok_cap = s.w_tok_cap, front_cap = s.front_cap;
    for (int i = tid; i < ncur; i += NT) {
      const int st = fcur[i];
      const float c = o2f((unsigned)ld_u64(&key[st]));
      if (c > cutoff) continue;
      for (int a = g.arc_off[st]; a < g.arc_off[st + 1]; a++) {
        if (g.arc_pdf[a] >= 0) continue;     // emitting arc
        const vbh_arc arc = g.arcs[a];
        const float tot = c + arc.weight;
        if (!(tot < cutoff)) continue;
        const int ds = arc.nextstate;
        const unsigned long long nk = EPK | f2o(tot);
        const unsigned long long old = atomicMin(&key[ds], nk);
        const bool create = (unsigned)(old >> 32) != ep;   // first touch of this state in this frame
        const int nid = wave_alloc(&sh.w_ntok, create);
        if (create) {
          if (nid < w_tok_cap) {
            w_state[nid] = ds; w_extra[nid] = 0u; w_tmp[nid] = INF_BITS;
            st_i32(&tokid[ds], nid);
          } else sh.error = ERR_W_TOK;
        }
        const bool push = (create || nk < old) && g.nieps[ds] > 0;   // new or improved: (re-)expand
        const int q = wave_alloc(&sh.fcnt[nxt], push);
        if (push) { if (q < front_cap) fnext[q] = ds; else sh.error = ERR_FRONT; }
      }
    }
    __syncthreads();
    if (sh.error != ERR_NONE) break;
    cur = nxt;
  }
  __syncthreads();
  // final costs of the frame's tokens + epsilon links regenerated from them (:824-845)
  const int ne = sh.w_ntok < s.w_tok_cap ? sh.w_ntok : s.w_tok_cap;
  const int lb = sh.w_nlink;
  __syncthreads();
  {
    const unsigned long long *key = uptr(s.key);
    const int *tokid = uptr(s.tokid), *w_state = uptr(s.w_state);
    float *w_cost = uptr(s.w_cost);
    int *wl_src = uptr(s.wl_src), *wl_dst = uptr(s.wl_dst), *wl_il = uptr(s.wl_il), *wl_ol = uptr(s.wl_ol);
    float *wl_gr = uptr(s.wl_gr), *wl_ac = uptr(s.wl_ac);
    const int w_link_cap = s.w_link_cap;
    for (int t = nb + tid; t < ne; t += NT) {
      const int st = w_state[t];
      const float c = o2f((unsigned)ld_u64(&key[st]));
      w_cost[t] = c;
      if (g.nieps[st] == 0 || c > cutoff) continue;
      for (int a = g.arc_off[st]; a < g.arc_off[st + 1]; a++) {
        if (g.arc_pdf[a] >= 0) continue;
        const vbh_arc arc = g.arcs[a];
        const float tot = c + arc.weight;
        if (!(tot < cutoff)) continue;
        const int l = wave_alloc(&sh.w_nlink, true);
        if (l < w_link_cap) {
          wl_src[l] = t | WBIT; wl_dst[l] = ld_i32(&tokid[arc.nextstate]) | WBIT;
          wl_il[l] = 0; wl_ol[l] = arc.olabel; wl_gr[l] = arc.weight; wl_ac[l] = 0.0f;
        } else sh.error = ERR_W_LINK;
      }
    }
  }
  __syncthreads();
  return lb;
}

__global__ __launch_bounds__(NT, (1024 / NT) * (NT / 256)) void decode_kernel(DevGraph g, DecCfg cfg, const Slot *slots,
                                                     const float *__restrict__ loglikes, int P,
                                                     const UttDesc *__restrict__ utts, const int *__restrict__ work_list,
                                                     int n_work, int *work_counter, UttResult *results, const Pool *pool_p) {
  __shared__ Shared sh;
  // The slot's ~45 pointers live in LDS: by value they overflow the SGPR file, and read through
  // the global pointer the compiler re-fetches them with vector loads inside every loop.
  __shared__ Slot s_slot;
  __shared__ Pool s_pool;
  const int tid = threadIdx.x;
  {
    const int *src = reinterpret_cast<const int *>(&slots[blockIdx.x]);
    int *dst = reinterpret_cast<int *>(&s_slot);
    for (int i = tid; i < (int)(sizeof(Slot) / sizeof(int)); i += NT) dst[i] = src[i];
    const int *psrc = reinterpret_cast<const int *>(pool_p);
    int *pdst = reinterpret_cast<int *>(&s_pool);
    for (int i = tid; i < (int)(sizeof(Pool) / sizeof(int)); i += NT) pdst[i] = psrc[i];
  }
  __syncthreads();
  const Slot &s = s_slot;
  const Pool &pool = s_pool;
  const float INF = INFINITY;
  int par = 0, pp = 0;

  while (true) {
    __syncthreads();
    if (tid == 0) sh.utt = atomicAdd(work_counter, 1);
    __syncthreads();
    if (sh.utt >= n_work) break;    // every wave of the grid reaches this exit
    const int utt = work_list ? work_list[sh.utt] : sh.utt;
    const UttDesc ud = utts[utt];
    const int T = ud.T;
    // epochs start at 1: the high word 0xFFFFFFFF is the cleared (memset 0xFF) state
    const int epoch_base = s.state[0] + 1;
    long long c_tok_exp = 0, c_arc_exp = 0;        // per-thread counters, reduced at the end
    long long t_last = clk();
#define PHASE(i) { if (tid == 0) { const long long now_ = clk(); sh.tph[i] += now_ - t_last; t_last = now_; } }
    if (tid == 0) {
      for (int i = 0; i < 8; i++) sh.tph[i] = 0;
      sh.w_ntok = 0; sh.w_nlink = 0; sh.h_ntok = 0; sh.h_nlink = 0; sh.error = ERR_NONE;
      sh.n_max_bound = 0; sh.n_min_bound = 0; sh.n_cand = 0; sh.n_visited = 0;
      sh.fcnt[0] = 0; sh.fcnt[1] = 0; sh.fcnt[2] = 0;
      sh.changed2[0] = 0; sh.changed2[1] = 0;
      sh.tail_tok_b = 0; sh.tail_link_b = 0; sh.tail_f0 = -1; sh.tail_f1 = -2;
    }
    pp = 0;
    __syncthreads();

    // ---- InitDecoding (:56-73): start token, then the epsilon closure with cutoff = beam
    int cur_b, cur_e;        // token range of the current frame (window or history indices)
    bool cur_w = true;
    {
      const unsigned ep = (unsigned)(0xFFFFFFFFu - (unsigned)(epoch_base + 0));
      if (tid == 0) {
        s.w_state[0] = g.start; s.w_cost[0] = 0.0f; s.w_extra[0] = 0u; s.w_tmp[0] = INF_BITS;
        s.key[g.start] = ((unsigned long long)ep << 32) | f2o(0.0f);
        s.tokid[g.start] = 0;
        sh.w_ntok = 1;
        if (g.nieps[g.start] > 0) { s.front[0][0] = g.start; sh.fcnt[0] = 1; }
        s.el_b[0] = 0; s.el_e[0] = 0; s.el_w[0] = 1;
      }
      __threadfence_block();
      __syncthreads();
      const int plb = process_nonemitting(g, s, 0, cfg.beam, ep, 0, sh);
      cur_b = 0; cur_e = sh.w_ntok < s.w_tok_cap ? sh.w_ntok : s.w_tok_cap;
      if (tid == 0) {
        s.tok_b[0] = 0; s.tok_e[0] = cur_e; s.tok_w[0] = 1;
        s.pl_b[0] = plb; s.pl_e[0] = sh.w_nlink < s.w_link_cap ? sh.w_nlink : s.w_link_cap; s.pl_w[0] = 1;
        sh.fcnt[0] = 0; sh.fcnt[1] = 0; sh.fcnt[2] = 0;
      }
      __syncthreads();
    }
    long long c_tok_new = cur_e, c_link_new = sh.w_nlink;

    int prev_front = -1;      // frontier frame at the previous pruning (-1: none yet)
    int frames_done = 0;
    PHASE(6)
    for (int frame = 0; frame < T && sh.error == ERR_NONE; frame++) {
      // ---- periodic pruning (:85-87); frame 0 has nothing behind it
      if (frame > 0 && frame % cfg.prune_interval == 0) {
        __syncthreads();
        // intermediate passes walk back at most 4 intervals beyond the unpruned frames: a bound on
        // the work (long-lived parallel paths would otherwise drag every pass back to frame 0);
        // whatever is left stale is a lower bound and is settled by the final pass
        prune_backward(s, frame - 1, prev_front < 0 ? 0 : prev_front, (prev_front < 0 ? 0 : prev_front) - 4 * cfg.prune_interval,
                       cfg.lattice_beam, cfg.lattice_beam * 0.1f, sh, pp);
        PHASE(4)
        if (!compact_window(s, prev_front, frame, sh)) break;
        prev_front = frame;
        cur_b = s.tok_b[frame]; cur_e = s.tok_e[frame]; cur_w = false;
        PHASE(5)
      }
      const int tb = cur_b, n = cur_e - cur_b;
      const int *tstate = uptr(cur_w ? s.w_state : s.h_state) + tb;
      const float *tcost = uptr(cur_w ? s.w_cost : s.h_cost) + tb;
      float *c_tot = uptr(s.c_tot), *c_ac = uptr(s.c_ac);
      int *c_arc = uptr(s.c_arc), *c_src = uptr(s.c_src), *c_dst = uptr(s.c_dst);
      const int cand_cap = s.cand_cap;
      const int enc = cur_w ? WBIT : 0;
      const float *llrow = loglikes + (ud.row0 + frame) * (long long)P;
      const unsigned ep = (unsigned)(0xFFFFFFFFu - (unsigned)(epoch_base + frame + 1));
      const unsigned long long EPK = (unsigned long long)ep << 32;

      // ---- GetCutoff (:618-685)
      unsigned long long bk = ~0ull;
      for (int i = tid; i < n; i += NT) {
        const unsigned long long k = ((unsigned long long)f2o(tcost[i]) << 32) | (unsigned)tstate[i];
        bk = k < bk ? k : bk;
      }
      bk = block_min_u64(bk, sh, par);
      const float best = n > 0 ? o2f((unsigned)(bk >> 32)) : INF;
      const int best_state = (int)(unsigned)bk;
      float cur_cutoff, adaptive_beam;
      if (cfg.max_active == INT_MAX && cfg.min_active == 0) {
        adaptive_beam = cfg.beam; cur_cutoff = best + cfg.beam;
      } else {
        const float beam_cutoff = best + cfg.beam;
        float mac = INF, mic = INF;
        unsigned long long cnt = 0;   // (#cost < beam_cutoff) << 32 | (#cost <= beam_cutoff)
        for (int i = tid; i < n; i += NT) cnt += ((unsigned long long)(tcost[i] < beam_cutoff) << 32) | (unsigned long long)(tcost[i] <= beam_cutoff);
        cnt = block_sum_u64(cnt, sh, par);
        const int cnt_lt = (int)(cnt >> 32), cnt_le = (int)(unsigned)cnt;
        // the order statistic is only needed when it decides the comparison
        if (n > cfg.max_active && cnt_lt > cfg.max_active) mac = select_kth(tcost, n, cfg.max_active, sh);
        if (mac < beam_cutoff) {
          adaptive_beam = mac - best + cfg.beam_delta; cur_cutoff = mac;
          if (tid == 0) sh.n_max_bound++;
        } else {
          if (n > cfg.min_active) {
            if (cfg.min_active == 0) mic = best;
            else if (cnt_le <= cfg.min_active) mic = select_kth(tcost, n, cfg.min_active, sh);
            else mic = beam_cutoff;   // some value <= beam_cutoff: the comparison below is false
          }
          if (mic > beam_cutoff) {
            adaptive_beam = mic - best + cfg.beam_delta; cur_cutoff = mic;
            if (tid == 0) sh.n_min_bound++;
          } else { adaptive_beam = cfg.beam; cur_cutoff = beam_cutoff; }
        }
      }
      // ---- cost_offset and the initial next_cutoff from the best token (:706-724)
      const float cost_offset = n > 0 ? -best : 0.0f;
      if (tid == 0) {
        float nc = INF;
        if (n > 0) {
          for (int a = g.arc_off[best_state]; a < g.arc_off[best_state + 1]; a++) {
            const int pdf = g.arc_pdf[a];
            if (pdf >= 0) {
              const float ll = cfg.acwt * llrow[pdf];
              const float nw = ((g.arcs[a].weight + cost_offset) - ll) + best;
              if (nw + adaptive_beam < nc) nc = nw + adaptive_beam;
            }
          }
        }
        sh.f0 = nc;
        s.cost_offset[frame] = cost_offset;
      }
      PHASE(0)
      // ---- pass 1 (token-parallel): every candidate arc -> scratch list; min of tot + adaptive_beam
      float mymin = INF;
      for (int i = tid; i < n; i += NT) {
        const float c = tcost[i];
        if (!(c <= cur_cutoff)) continue;
        c_tok_exp++;
        const int st = tstate[i];
        for (int a = g.arc_off[st]; a < g.arc_off[st + 1]; a++) {
          const int pdf = g.arc_pdf[a];
          if (pdf < 0) continue;
          c_arc_exp++;
          const vbh_arc arc = g.arcs[a];
          const float ac = cost_offset - cfg.acwt * llrow[pdf];
          const float tot = (c + ac) + arc.weight;
          mymin = fminf(mymin, tot + adaptive_beam);
          const int q = wave_alloc(&sh.n_cand, true);
          if (q < cand_cap) { c_tot[q] = tot; c_ac[q] = ac; c_arc[q] = a; c_src[q] = i; c_dst[q] = arc.nextstate; }
          else sh.error = ERR_CAND;
        }
      }
      mymin = block_min_f(mymin, sh, par);
      const float next_cutoff = fminf(sh.f0, mymin);
      const int ncand = sh.n_cand < cand_cap ? sh.n_cand : cand_cap;
      const int nb = sh.w_ntok;     // first token of frame+1
      PHASE(1)
      // ---- pass 2 (candidate-parallel): admit tot <= next_cutoff; min-cost per state via one
      //      64-bit atomicMin; the thread that meets a stale epoch creates the token
      {
        unsigned long long *key = uptr(s.key);
        int *tokid = uptr(s.tokid), *w_state = uptr(s.w_state), *front0 = uptr(s.front[0]);
        unsigned *w_extra = uptr(s.w_extra), *w_tmp = uptr(s.w_tmp);
        const int w_tok_cap = s.w_tok_cap, front_cap = s.front_cap;
        for (int q = tid; q < ncand; q += NT) {
          const float tot = c_tot[q];
          if (tot > next_cutoff) continue;
          const int ds = c_dst[q];
          const unsigned long long old = atomicMin(&key[ds], EPK | f2o(tot));
          const bool create = (unsigned)(old >> 32) != ep;
          const int nid = wave_alloc(&sh.w_ntok, create);
          bool ok_tok = false;
          if (create) {
            if (nid < w_tok_cap) {
              w_state[nid] = ds; w_extra[nid] = 0u; w_tmp[nid] = INF_BITS;
              st_i32(&tokid[ds], nid);
              ok_tok = true;
            } else sh.error = ERR_W_TOK;
          }
          const bool pushf = ok_tok && g.nieps[ds] > 0;
          const int f = wave_alloc(&sh.fcnt[0], pushf);
          if (pushf) { if (f < front_cap) front0[f] = ds; else sh.error = ERR_FRONT; }
        }
      }
      __syncthreads();
      // ---- pass 3 (candidate-parallel): forward links (:765-767)
      const int lb = sh.w_nlink;
      if (sh.error == ERR_NONE) {
        const int *tokid = uptr(s.tokid);
        int *wl_src = uptr(s.wl_src), *wl_dst = uptr(s.wl_dst), *wl_il = uptr(s.wl_il), *wl_ol = uptr(s.wl_ol);
        float *wl_gr = uptr(s.wl_gr), *wl_ac = uptr(s.wl_ac);
        const int w_link_cap = s.w_link_cap;
        for (int q = tid; q < ncand; q += NT) {
          const float tot = c_tot[q];
          if (tot > next_cutoff) continue;
          const vbh_arc arc = g.arcs[c_arc[q]];
          const int l = wave_alloc(&sh.w_nlink, true);
          if (l < w_link_cap) {
            wl_src[l] = (tb + c_src[q]) | enc; wl_dst[l] = ld_i32(&tokid[arc.nextstate]) | WBIT;
            wl_il[l] = arc.ilabel; wl_ol[l] = arc.olabel; wl_gr[l] = arc.weight; wl_ac[l] = c_ac[q];
          } else sh.error = ERR_W_LINK;
        }
      }
      __syncthreads();
      const int le_emit = sh.w_nlink < s.w_link_cap ? sh.w_nlink : s.w_link_cap;
      if (sh.error != ERR_NONE) break;
      PHASE(2)
      // ---- ProcessNonemitting(next_cutoff)
      const int plb = process_nonemitting(g, s, nb, next_cutoff, ep, 0, sh);
      cur_b = nb; cur_e = sh.w_ntok < s.w_tok_cap ? sh.w_ntok : s.w_tok_cap; cur_w = true;
      const int pl_end = sh.w_nlink < s.w_link_cap ? sh.w_nlink : s.w_link_cap;
      if (tid == 0) {
        s.el_b[frame] = lb; s.el_e[frame] = le_emit; s.el_w[frame] = 1;
        s.tok_b[frame + 1] = cur_b; s.tok_e[frame + 1] = cur_e; s.tok_w[frame + 1] = 1;
        s.pl_b[frame + 1] = plb; s.pl_e[frame + 1] = pl_end; s.pl_w[frame + 1] = 1;
        sh.n_cand = 0; sh.fcnt[0] = 0; sh.fcnt[1] = 0; sh.fcnt[2] = 0;
      }
      c_tok_new += cur_e - cur_b; c_link_new += pl_end - lb;
      frames_done = frame + 1;
      __syncthreads();
      PHASE(3)
    }
    __syncthreads();
    const bool search_ok = sh.error == ERR_NONE && frames_done == T;

    // ---- FinalizeDecoding (:600-615)
    float final_relative = INF, final_best = INF;
    int final_empty = 1;
    if (search_ok) {
      const int F = T;
      if (tid == 0) {   // ranges of the last frame as the pruning code reads them
        s.el_b[F] = 0; s.el_e[F] = 0; s.el_w[F] = 0;
      }
      __threadfence_block();
      __syncthreads();
      const int tb = s.tok_b[F], te = s.tok_e[F];
      const bool tw = s.tok_w[F];
      const int *tstate = tw ? s.w_state : s.h_state;
      const float *tcost = tw ? s.w_cost : s.h_cost;
      unsigned *extra = tw ? s.w_extra : s.h_extra, *tmp = tw ? s.w_tmp : s.h_tmp;
      // ComputeFinalCosts (:532-572)
      float bc = INF, bwf = INF; unsigned long long anyfinal = 0;
      for (int t = tb + tid; t < te; t += NT) {
        const float fc = g.final_cost[tstate[t]], c = tcost[t];
        bc = fminf(bc, c); bwf = fminf(bwf, c + fc);
        anyfinal += fc != INF;
      }
      bc = block_min_f(bc, sh, par); bwf = block_min_f(bwf, sh, par);
      anyfinal = block_sum_u64(anyfinal, sh, par);
      final_empty = anyfinal == 0;
      final_relative = (bc == INF && bwf == INF) ? INF : bwf - bc;
      final_best = (bwf != INF) ? bwf : bc;
      // PruneForwardLinksFinal (:376-458), to the fixed point
      const int pb = s.pl_b[F], pe = s.pl_e[F];
      const bool pw = s.pl_w[F];
      for (int iter = 0; iter < MAX_EPS_ITERS; iter++) {
        for (int t = tb + tid; t < te; t += NT) {
          const float fc = final_empty ? 0.0f : g.final_cost[tstate[t]];
          const float x = tcost[t] + fc - final_best;    // >= 0 by construction
          tmp[t] = __float_as_uint(x);
        }
        __syncthreads();
        eval_links(s, pw, pb, pe, tcost, tmp, cfg.lattice_beam);
        __syncthreads();
        if (tid == 0) sh.changed2[pp ^ 1] = 0;
        int ch = 0;
        for (int t = tb + tid; t < te; t += NT) {
          float v = __uint_as_float(tmp[t]);
          tmp[t] = INF_BITS;
          if (v > cfg.lattice_beam) v = INF;
          const unsigned vb = __float_as_uint(v);
          if (vb != extra[t]) { extra[t] = vb; ch = 1; }
        }
        if (ch) sh.changed2[pp] = 1;
        __syncthreads();
        const bool any = sh.changed2[pp] != 0;
        pp ^= 1;
        if (!any || pe == pb) break;
      }
      prune_backward(s, F - 1, /*first_unpruned=*/0, /*stop_at=*/0, cfg.lattice_beam, 0.0f, sh, pp);   // every frame, delta 0 (:606-611)
      PHASE(4)
      compact_window(s, prev_front, F, sh);
      PHASE(5)
    }
    __syncthreads();

    // ---- write the surviving lattice to the dense pool (content of GetRawLattice :110-192)
    int r_ok = 0, r_error = sh.error, r_reached_final = 0, r_n_tok = 0, r_n_link = 0;
    long long r_tok_off = 0, r_link_off = 0, r_cost_off = 0;
    if (search_ok && sh.error == ERR_NONE) {
      const int F = T;
      unsigned long long na = 0, nl = 0;
      for (int t = tid; t < sh.h_ntok; t += NT) na += s.h_extra[t] != INF_BITS;
      na = block_sum_u64(na, sh, par);
      for (int l = tid; l < sh.h_nlink; l += NT) {
        const int src = s.hl_src[l];
        nl += src >= 0 && s.h_extra[src] != INF_BITS && s.h_extra[s.hl_dst[l]] != INF_BITS;
      }
      nl = block_sum_u64(nl, sh, par);
      if (tid == 0) {
        sh.u0 = atomicAdd(&pool.counters[0], na);
        sh.flag = 0;
      }
      __syncthreads();
      const long long toff = (long long)sh.u0;
      __syncthreads();
      if (tid == 0) sh.u0 = atomicAdd(&pool.counters[1], nl);
      __syncthreads();
      const long long loff = (long long)sh.u0;
      __syncthreads();
      if (tid == 0) sh.u0 = atomicAdd(&pool.counters[2], (unsigned long long)T);
      __syncthreads();
      const long long coff_off = (long long)sh.u0;
      const bool fits = toff + (long long)na <= pool.tok_cap && loff + (long long)nl <= pool.link_cap && coff_off + T <= pool.cost_cap;
      if (fits) {
        // dense index per alive history token, kept in h_tmp
        for (int f = 0; f <= F; f++) {
          for (int t = s.tok_b[f] + tid; t < s.tok_e[f]; t += NT) {
            if (s.h_extra[t] == INF_BITS) continue;
            const int d = atomicAdd(&sh.flag, 1);
            s.h_tmp[t] = (unsigned)d;
            const long long o = toff + d;
            pool.tok_frame[o] = f; pool.tok_state[o] = s.h_state[t]; pool.tok_cost[o] = s.h_cost[t];
            pool.tok_extra[o] = __uint_as_float(s.h_extra[t]);
            float fin = INF;
            if (f == F) fin = final_empty ? 0.0f : g.final_cost[s.h_state[t]];
            pool.tok_final[o] = fin;
          }
        }
        __syncthreads();
        if (tid == 0) sh.flag = 0;
        __syncthreads();
        for (int f = 0; f <= F; f++) {
          for (int pass = 0; pass < 2; pass++) {
            const int lb = pass ? s.pl_b[f] : s.el_b[f], le_ = pass ? s.pl_e[f] : s.el_e[f];
            const float coff = (!pass && f < T) ? s.cost_offset[f] : 0.0f;
            for (int l = lb + tid; l < le_; l += NT) {
              const int src = s.hl_src[l];
              if (src < 0) continue;
              const int dst = s.hl_dst[l];
              if (s.h_extra[src] == INF_BITS || s.h_extra[dst] == INF_BITS) continue;
              const int d = atomicAdd(&sh.flag, 1);
              const long long o = loff + d;
              pool.link_src[o] = (int)s.h_tmp[src]; pool.link_dst[o] = (int)s.h_tmp[dst];
              pool.link_il[o] = s.hl_il[l]; pool.link_ol[o] = s.hl_ol[l]; pool.link_gr[o] = s.hl_gr[l];
              pool.link_ac[o] = s.hl_il[l] != 0 ? s.hl_ac[l] - coff : s.hl_ac[l];
            }
          }
        }
        for (int f = tid; f < T; f += NT) pool.cost_offsets[coff_off + f] = s.cost_offset[f];
        r_n_tok = (int)na; r_n_link = (int)nl; r_tok_off = toff; r_link_off = loff; r_cost_off = coff_off;
        r_reached_final = final_relative != INF;
        // Decode() returns whether the last frame still has tokens (:93-95)
        unsigned long long alive_last = 0;
        for (int t = s.tok_b[F] + tid; t < s.tok_e[F]; t += NT) alive_last += s.h_extra[t] != INF_BITS;
        alive_last = block_sum_u64(alive_last, sh, par);
        r_ok = alive_last > 0;
        // restore the +inf invariant of h_tmp for the next utterance in this slot
        __syncthreads();
        for (int t = tid; t < sh.h_ntok; t += NT) s.h_tmp[t] = INF_BITS;
      } else r_error = ERR_POOL;
    }
    PHASE(6)
    {
      const unsigned long long te_ = block_sum_u64((unsigned long long)c_tok_exp, sh, par);
      const unsigned long long ae_ = block_sum_u64((unsigned long long)c_arc_exp, sh, par);
      if (tid == 0) {
        UttResult *r = &results[utt];
        r->ok = r_ok; r->error = r_error; r->reached_final = r_reached_final; r->num_frames = frames_done;
        r->final_relative_cost = final_relative; r->final_best_cost = final_best; r->final_costs_empty = final_empty;
        r->n_tok = r_n_tok; r->n_link = r_n_link; r->tok_off = r_tok_off; r->link_off = r_link_off; r->cost_off = r_cost_off;
        r->n_tokens_expanded = (long long)te_; r->n_arcs_expanded = (long long)ae_;
        r->n_tokens_created = c_tok_new; r->n_links_created = c_link_new;
        r->n_max_bound = sh.n_max_bound; r->n_min_bound = sh.n_min_bound;
        for (int i = 0; i < 7; i++) r->t_phase[i] = sh.tph[i];
        r->t_phase[7] = sh.n_visited;   // slot 7: frames visited by pruning (not a time)
        s.state[0] = epoch_base + T + 2;   // next utterance in this slot: disjoint epochs
      }
    }
    __syncthreads();
  }
#undef PHASE
}

// arc_pdf[a] = tid2pdf[ilabel] (-1 for input-epsilon arcs): one dependent load less per arc
__global__ void fill_arc_pdf_kernel(const vbh_arc *__restrict__ arcs, long long A, const int *__restrict__ tid2pdf,
                                    int *__restrict__ arc_pdf) {
  for (long long a = blockIdx.x * (long long)blockDim.x + threadIdx.x; a < A; a += (long long)gridDim.x * blockDim.x) {
    const int il = arcs[a].ilabel;
    arc_pdf[a] = il == 0 ? -1 : tid2pdf[il];
  }
}

}  // namespace

// ============================================================================ host side
struct vbh_graph {
  vbh_ctx *ctx;
  int S, start; long long A; int max_ilabel, max_degree;
  DevBuf<float> final_cost; DevBuf<int> arc_off, nieps, arc_pdf; DevBuf<vbh_arc> arcs;
  std::vector<float> h_final;
};

struct HostLattice {
  std::vector<int32_t> tok_frame, tok_state, link_src, link_dst, link_il, link_ol, ali, words;
  std::vector<float> tok_cost, tok_extra, tok_final, link_gr, link_ac;
  vbh_lattice view;
};

struct SlotSet {          // workspaces of one size class
  int n = 0;
  Slot proto{};
  std::vector<Slot> h;
  DevBuf<Slot> d;
};

struct vbh_decoder {
  vbh_ctx *ctx; const vbh_graph *graph; vbh_decode_cfg cfg;
  int max_frames;
  SlotSet small_set, big_set;       // big: generous capacities, allocated on first overflow
  std::vector<void *> allocs;
  std::vector<HostLattice> results;
  std::vector<UttResult> h_res;
  int n_retried = 0;
};

static int slot_alloc(vbh_decoder *d, size_t bytes, void **p, int memset_val = -2) {
  VBH_HIP(hipMalloc(p, bytes ? bytes : 8));
  d->allocs.push_back(*p);
  if (memset_val != -2) VBH_HIP(hipMemsetAsync(*p, memset_val, bytes, d->ctx->stream));
  return 0;
}

static const char *err_name(int e);

static int alloc_slot_set(vbh_decoder *d, SlotSet *set, int n, bool big) {
  const vbh_graph *graph = d->graph;
  const vbh_decode_cfg *cfg = &d->cfg;
  const int max_frames = d->max_frames;
  const long long S = graph->S;
  const long long ma = std::min<long long>(cfg->max_active, S);
  // new tokens per frame are bounded by the arcs leaving the expanded tokens (~max_active * out-degree)
  const long long per_frame_tok = std::min<long long>(S, big ? std::max<long long>(8192, 8 * ma) : std::max<long long>(4096, 2 * ma));
  const long long per_frame_link = std::min<long long>(std::max<long long>(graph->A, 1), 2 * per_frame_tok);
  const long long surv_tok = big ? 2048 : 160, surv_link = big ? 4096 : 320;   // lattice survivors per frame
  const int win = cfg->prune_interval + 1;
  Slot p{};
  p.w_tok_cap = (int)std::min<long long>(per_frame_tok * win, 1LL << 28);
  p.w_link_cap = (int)std::min<long long>((per_frame_link + per_frame_tok) * win, 1LL << 28);
  // history: one unpruned frontier frame per interval plus the survivors
  const long long n_int = max_frames / cfg->prune_interval + 2;
  const long long front_avg = big ? per_frame_tok : per_frame_tok / 2;
  p.h_tok_cap = (int)std::min<long long>(front_avg * n_int + surv_tok * (max_frames + 1) + per_frame_tok, 1LL << 29);
  p.h_link_cap = (int)std::min<long long>(front_avg * n_int + surv_link * (max_frames + 1) + per_frame_link, 1LL << 29);
  p.front_cap = (int)(2 * per_frame_tok);
  // candidates of one frame = arcs leaving the expanded tokens: at most ~max_active tokens are
  // expanded (GetCutoff), each with at most max_degree arcs
  p.cand_cap = (int)std::min<long long>(std::max<long long>(graph->A, 1),
                                        std::min<long long>((ma + 256) * std::max(graph->max_degree, 1), big ? (1LL << 24) : (1LL << 21)));
  set->proto = p; set->n = n;
  set->h.assign(n, p);
  const size_t F = (size_t)max_frames + 2;
  // one allocation per field, covering all slots (slot i owns [i*count, (i+1)*count))
  const size_t ns = (size_t)n;
#define A_(field, type, count, ms)                                                                   \
  {                                                                                                  \
    type *base = nullptr;                                                                            \
    if (slot_alloc(d, sizeof(type) * (size_t)(count) * ns, reinterpret_cast<void **>(&base), ms)) return -1; \
    for (size_t i = 0; i < ns; i++) set->h[i].field = base + i * (size_t)(count);                    \
  }
  A_(key, unsigned long long, S, 0xFF) A_(tokid, int, S, 0)
  A_(w_state, int, p.w_tok_cap, -2) A_(w_cost, float, p.w_tok_cap, -2) A_(w_extra, unsigned, p.w_tok_cap, -2)
  A_(w_tmp, unsigned, p.w_tok_cap, -2) A_(w_remap, int, p.w_tok_cap, -2)
  A_(wl_src, int, p.w_link_cap, -2) A_(wl_dst, int, p.w_link_cap, -2) A_(wl_il, int, p.w_link_cap, -2)
  A_(wl_ol, int, p.w_link_cap, -2) A_(wl_gr, float, p.w_link_cap, -2) A_(wl_ac, float, p.w_link_cap, -2)
  A_(h_state, int, p.h_tok_cap, -2) A_(h_cost, float, p.h_tok_cap, -2) A_(h_extra, unsigned, p.h_tok_cap, -2)
  A_(h_tmp, unsigned, p.h_tok_cap, -2) A_(h_remap, int, p.h_tok_cap, -2)
  A_(hl_src, int, p.h_link_cap, -2) A_(hl_dst, int, p.h_link_cap, -2) A_(hl_il, int, p.h_link_cap, -2)
  A_(hl_ol, int, p.h_link_cap, -2) A_(hl_gr, float, p.h_link_cap, -2) A_(hl_ac, float, p.h_link_cap, -2)
  A_(tok_b, int, F, 0) A_(tok_e, int, F, 0) A_(el_b, int, F, 0) A_(el_e, int, F, 0) A_(pl_b, int, F, 0) A_(pl_e, int, F, 0)
  A_(tok_w, unsigned char, F, 0) A_(el_w, unsigned char, F, 0) A_(pl_w, unsigned char, F, 0)
  A_(cost_offset, float, F, 0)
  A_(c_tot, float, p.cand_cap, -2) A_(c_ac, float, p.cand_cap, -2) A_(c_arc, int, p.cand_cap, -2)
  A_(c_src, int, p.cand_cap, -2) A_(c_dst, int, p.cand_cap, -2)
  A_(front[0], int, p.front_cap, -2) A_(front[1], int, p.front_cap, -2) A_(front[2], int, p.front_cap, -2)
  A_(state, int, 4, 0)
#undef A_
  if (set->d.upload(set->h.data(), set->h.size(), d->ctx->stream)) return -1;
  return 0;
}

extern "C" {

void vbh_decode_cfg_default(vbh_decode_cfg *c) {
  // VB/scr/kaldi_scr.h:152-157 (Decode defaults) over lattice-faster-decoder.h:59-67
  c->beam = 13.0f; c->max_active = 7000; c->min_active = 200; c->lattice_beam = 6.0f;
  c->prune_interval = 25; c->beam_delta = 0.5f; c->acoustic_scale = 0.083333f;
}

int vbh_graph_create(vbh_ctx *ctx, int32_t num_states, int32_t start, const float *h_final,
                     const int64_t *h_arc_offsets, const vbh_arc *h_arcs, vbh_graph **out) {
  VBH_CHECK(ctx && h_final && h_arc_offsets && out, "vbh_graph_create: NULL argument");
  VBH_CHECK(num_states > 0 && start >= 0 && start < num_states, "vbh_graph_create: bad start state %d / %d states", start, num_states);
  const long long A = h_arc_offsets[num_states];
  VBH_CHECK(h_arc_offsets[0] == 0 && A >= 0 && A < 2147483647LL, "vbh_graph_create: bad arc offsets");
  VBH_CHECK(A == 0 || h_arcs, "vbh_graph_create: arcs missing");
  if (vbh_use_device(ctx)) return -1;
  std::vector<int> off(num_states + 1), nieps(num_states, 0);
  int max_il = 0, max_deg = 0;
  for (int s = 0; s <= num_states; s++) {
    VBH_CHECK(s == 0 || h_arc_offsets[s] >= h_arc_offsets[s - 1], "vbh_graph_create: arc offsets not monotone at state %d", s);
    off[s] = (int)h_arc_offsets[s];
  }
  for (int s = 0; s < num_states; s++) {
    if (off[s + 1] - off[s] > max_deg) max_deg = off[s + 1] - off[s];
    for (int a = off[s]; a < off[s + 1]; a++) {
      VBH_CHECK(h_arcs[a].nextstate >= 0 && h_arcs[a].nextstate < num_states, "vbh_graph_create: arc %d leaves the graph", a);
      VBH_CHECK(h_arcs[a].ilabel >= 0, "vbh_graph_create: negative ilabel on arc %d", a);
      nieps[s] += h_arcs[a].ilabel == 0;
      if (h_arcs[a].ilabel > max_il) max_il = h_arcs[a].ilabel;
    }
  }
  vbh_graph *g = new vbh_graph();
  g->ctx = ctx; g->S = num_states; g->start = start; g->A = A; g->max_ilabel = max_il; g->max_degree = max_deg;
  g->h_final.assign(h_final, h_final + num_states);
  hipStream_t st = ctx->stream;
  if (g->final_cost.upload(h_final, num_states, st) || g->arc_off.upload(off.data(), off.size(), st) ||
      g->nieps.upload(nieps.data(), nieps.size(), st) || g->arcs.upload(h_arcs, (size_t)A, st) ||
      g->arc_pdf.alloc((size_t)(A > 0 ? A : 1))) { delete g; return -1; }
  VBH_HIP(hipStreamSynchronize(st));
  *out = g;
  return 0;
}

void vbh_graph_destroy(vbh_graph *g) {
  if (!g) return;
  (void)hipSetDevice(g->ctx->device);
  (void)hipStreamSynchronize(g->ctx->stream);
  delete g;
}

int vbh_decoder_create(vbh_ctx *ctx, const vbh_graph *graph, const vbh_decode_cfg *cfg, int32_t max_batch,
                       int32_t max_frames, vbh_decoder **out) {
  VBH_CHECK(ctx && graph && cfg && out, "vbh_decoder_create: NULL argument");
  VBH_CHECK(cfg->beam > 0.0f && cfg->max_active > 1 && cfg->lattice_beam > 0.0f && cfg->prune_interval > 0 && cfg->beam_delta > 0.0f,
            "vbh_decoder_create: invalid config (LatticeFasterDecoderConfig::Check)");
  VBH_CHECK(cfg->min_active >= 0, "vbh_decoder_create: min_active < 0");
  VBH_CHECK(max_batch >= 1 && max_frames >= 0, "vbh_decoder_create: bad batch/frames");
  VBH_CHECK(cfg->prune_interval <= MAX_WIN_FRAMES - 2, "vbh_decoder_create: prune_interval %d > %d", cfg->prune_interval, MAX_WIN_FRAMES - 2);
  if (vbh_use_device(ctx)) return -1;
  vbh_decoder *d = new vbh_decoder();
  d->ctx = ctx; d->graph = graph; d->cfg = *cfg; d->max_frames = max_frames;
  // Two size classes.  "small": sized for the common case (new tokens per frame <= 2*max_active,
  // a few hundred lattice survivors per frame), four workgroups per CU.  "big": the generous
  // worst-case sizes, a handful of slots, allocated only when an utterance overflows a small slot;
  // such utterances are searched again there (vbh_decode_batch), so capacity never truncates a result.
  if (alloc_slot_set(d, &d->small_set, std::min<int>(max_batch, (1024 / NT) * ctx->num_cus), /*big=*/false)) { vbh_decoder_destroy(d); return -1; }
  VBH_HIP(hipStreamSynchronize(ctx->stream));
  *out = d;
  return 0;
}

void vbh_decoder_destroy(vbh_decoder *d) {
  if (!d) return;
  (void)hipSetDevice(d->ctx->device);
  (void)hipStreamSynchronize(d->ctx->stream);
  for (void *p : d->allocs) (void)hipFree(p);
  delete d;
}

}  // extern "C"

// LatticeWeight natural order (fstext/lattice-weight.h Compare): smaller sum wins, ties on
// the first component
static inline bool lw_better(float a1, float a2, float b1, float b2) {
  const float fa = a1 + a2, fb = b1 + b2;
  if (fa < fb) return true;
  if (fa > fb) return false;
  return a1 < b1;
}

// canonical order (frame, state) for tokens, (src, dst, ilabel, olabel, graph, acoustic) for
// links; best path = ShortestPath over the raw lattice + GetLinearSymbolSequence
// (lattice-faster-decoder.cc:100-106, fstext/fstext-utils-inl.h:178-216)
static void finish_lattice(const vbh_graph *g, const UttResult &r, const Pool &hp, HostLattice *L) {
  const int n = r.n_tok, nl = r.n_link;
  std::vector<int> order(n), rank(n);
  for (int i = 0; i < n; i++) order[i] = i;
  const int *tf = hp.tok_frame + r.tok_off, *ts = hp.tok_state + r.tok_off;
  std::sort(order.begin(), order.end(), [&](int a, int b) { return tf[a] != tf[b] ? tf[a] < tf[b] : ts[a] < ts[b]; });
  L->tok_frame.resize(n); L->tok_state.resize(n); L->tok_cost.resize(n); L->tok_extra.resize(n); L->tok_final.resize(n);
  for (int i = 0; i < n; i++) {
    const int o = order[i];
    rank[o] = i;
    L->tok_frame[i] = tf[o]; L->tok_state[i] = ts[o]; L->tok_cost[i] = hp.tok_cost[r.tok_off + o];
    L->tok_extra[i] = hp.tok_extra[r.tok_off + o]; L->tok_final[i] = hp.tok_final[r.tok_off + o];
  }
  struct LR { int s, d, il, ol; float g, a; };
  std::vector<LR> recs(nl);
  for (int i = 0; i < nl; i++) {
    const long long o = r.link_off + i;
    recs[i] = LR{rank[hp.link_src[o]], rank[hp.link_dst[o]], hp.link_il[o], hp.link_ol[o], hp.link_gr[o], hp.link_ac[o]};
  }
  std::sort(recs.begin(), recs.end(), [](const LR &x, const LR &y) {
    if (x.s != y.s) return x.s < y.s;
    if (x.d != y.d) return x.d < y.d;
    if (x.il != y.il) return x.il < y.il;
    if (x.ol != y.ol) return x.ol < y.ol;
    if (x.g != y.g) return x.g < y.g;
    return x.a < y.a;
  });
  L->link_src.resize(nl); L->link_dst.resize(nl); L->link_il.resize(nl); L->link_ol.resize(nl); L->link_gr.resize(nl); L->link_ac.resize(nl);
  for (int i = 0; i < nl; i++) {
    L->link_src[i] = recs[i].s; L->link_dst[i] = recs[i].d; L->link_il[i] = recs[i].il; L->link_ol[i] = recs[i].ol;
    L->link_gr[i] = recs[i].g; L->link_ac[i] = recs[i].a;
  }
  // single shortest path: links are sorted by source token, i.e. by frame
  const float INF = INFINITY;
  std::vector<float> d1(n, INF), d2(n, INF);
  std::vector<int> par(n, -1);
  L->ali.clear(); L->words.clear();
  float bg = INF, ba = INF;
  int start = -1;
  for (int i = 0; i < n && L->tok_frame[i] == 0; i++) if (L->tok_state[i] == g->start) start = i;
  if (start >= 0) {
    d1[start] = 0.0f; d2[start] = 0.0f;
    int lo = 0;
    while (lo < nl) {
      const int f = L->tok_frame[recs[lo].s];
      int hi = lo;
      while (hi < nl && L->tok_frame[recs[hi].s] == f) hi++;
      bool again = true;
      while (again) {
        again = false;
        for (int i = lo; i < hi; i++) {
          const int s_ = recs[i].s, t = recs[i].d;
          if (d1[s_] == INF && d2[s_] == INF) continue;
          const float n1 = d1[s_] + recs[i].g, n2 = d2[s_] + recs[i].a;
          if (lw_better(n1, n2, d1[t], d2[t])) {
            d1[t] = n1; d2[t] = n2; par[t] = i;
            if (L->tok_frame[t] == f) again = true;
          }
        }
      }
      lo = hi;
    }
    int best = -1; float b1 = INF, b2 = INF;
    for (int i = 0; i < n; i++) {
      if (L->tok_frame[i] != r.num_frames || L->tok_final[i] == INF) continue;
      if (d1[i] == INF && d2[i] == INF) continue;
      const float f1 = d1[i] + L->tok_final[i], f2 = d2[i] + 0.0f;
      if (best < 0 || lw_better(f1, f2, b1, b2)) { best = i; b1 = f1; b2 = f2; }
    }
    if (best >= 0) {
      std::vector<int> path;
      for (int t = best; par[t] >= 0; t = recs[par[t]].s) path.push_back(par[t]);
      float t1 = 0.0f, t2 = 0.0f;
      for (int i = (int)path.size() - 1; i >= 0; i--) {
        const LR &x = recs[path[i]];
        t1 = x.g + t1; t2 = x.a + t2;
        if (x.il != 0) L->ali.push_back(x.il);
        if (x.ol != 0) L->words.push_back(x.ol);
      }
      bg = L->tok_final[best] + t1; ba = 0.0f + t2;
    }
  }
  vbh_lattice &v = L->view;
  memset(&v, 0, sizeof(v));
  v.ok = r.ok; v.reached_final = r.reached_final; v.num_frames = r.num_frames; v.final_relative_cost = r.final_relative_cost;
  v.num_toks = n; v.tok_frame = L->tok_frame.data(); v.tok_state = L->tok_state.data(); v.tok_cost = L->tok_cost.data();
  v.tok_extra = L->tok_extra.data(); v.tok_final = L->tok_final.data();
  v.num_links = nl; v.link_src = L->link_src.data(); v.link_dst = L->link_dst.data(); v.link_ilabel = L->link_il.data();
  v.link_olabel = L->link_ol.data(); v.link_graph = L->link_gr.data(); v.link_acoustic = L->link_ac.data();
  v.num_ali = (int)L->ali.size(); v.ali = L->ali.data(); v.num_words = (int)L->words.size(); v.words = L->words.data();
  v.best_graph = bg; v.best_acoustic = ba;
  v.n_tokens_expanded = r.n_tokens_expanded; v.n_arcs_expanded = r.n_arcs_expanded;
  v.n_tokens_created = r.n_tokens_created; v.n_links_created = r.n_links_created;
  v.n_frames_maxactive_bound = r.n_max_bound; v.n_frames_minactive_bound = r.n_min_bound;
}

static const char *err_name(int e) {
  switch (e) {
    case ERR_W_TOK: return "raw-window token capacity exceeded";
    case ERR_W_LINK: return "raw-window link capacity exceeded";
    case ERR_H_TOK: return "history token capacity exceeded";
    case ERR_H_LINK: return "history link capacity exceeded";
    case ERR_FRONT: return "epsilon work-list capacity exceeded";
    case ERR_EPS_LOOP: return "epsilon closure did not converge (epsilon cycle in the graph?)";
    case ERR_POOL: return "output pool capacity exceeded";
    case ERR_CAND: return "candidate list capacity exceeded";
    default: return "unknown";
  }
}

extern "C" {

int vbh_decode_batch(vbh_decoder *d, const float *d_loglikes, int32_t num_pdfs, const int64_t *h_frame_offsets,
                     int32_t n_utt, const int32_t *h_tid2pdf, int32_t num_tids) {
  VBH_CHECK(d && h_frame_offsets && h_tid2pdf, "vbh_decode_batch: NULL argument");
  VBH_CHECK(n_utt >= 0 && num_pdfs > 0 && num_tids >= 0, "vbh_decode_batch: bad sizes");
  if (vbh_use_device(d->ctx)) return -1;
  d->results.clear(); d->h_res.clear();
  if (n_utt == 0) return 0;
  const vbh_graph *g = d->graph;
  // the graph's ilabels must be valid transition-ids ("Likely graph/model mismatch",
  // decodable-am-diag-gmm.cc:32-33) -- checked once per batch on the tid2pdf table
  VBH_CHECK(g->max_ilabel <= num_tids, "vbh_decode_batch: graph uses transition-id %d but the model has %d (graph/model mismatch)", g->max_ilabel, num_tids);
  for (int t = 1; t <= num_tids; t++)
    VBH_CHECK(h_tid2pdf[t] >= 0 && h_tid2pdf[t] < num_pdfs, "vbh_decode_batch: tid2pdf[%d] = %d out of range", t, h_tid2pdf[t]);
  std::vector<UttDesc> utts(n_utt);
  long long total_frames = 0;
  for (int u = 0; u < n_utt; u++) {
    const long long T = h_frame_offsets[u + 1] - h_frame_offsets[u];
    VBH_CHECK(T >= 0 && T <= d->max_frames, "vbh_decode_batch: utterance %d has %lld frames (decoder built for %d)", u, T, d->max_frames);
    utts[u] = UttDesc{h_frame_offsets[u], (int)T, 0};
    total_frames += T;
  }
  VBH_CHECK(total_frames == 0 || d_loglikes, "vbh_decode_batch: NULL loglikes");
  hipStream_t st = d->ctx->stream;
  DevBuf<UttDesc> d_utts; DevBuf<int> d_t2p, d_counter; DevBuf<UttResult> d_res;
  // graph labels beyond the table would read out of bounds: verify on the host copy
  // (cheap: done in vbh_graph_create for states; here for labels)
  std::vector<int> t2p(h_tid2pdf, h_tid2pdf + num_tids + 1);
  t2p[0] = 0;
  if (d_utts.upload(utts.data(), utts.size(), st) || d_t2p.upload(t2p.data(), t2p.size(), st) || d_counter.alloc(1) || d_res.alloc(n_utt)) return -1;
  VBH_HIP(hipMemsetAsync(d_counter.p, 0, sizeof(int), st));
  // output pool: the pruned lattice is a small multiple of the frame count
  Pool pool{};
  const long long tok_cap = std::max<long long>(4096, 64 * (total_frames + n_utt)), link_cap = std::max<long long>(8192, 128 * (total_frames + n_utt));
  pool.tok_cap = tok_cap; pool.link_cap = link_cap; pool.cost_cap = total_frames + 1;
  DevBuf<int> p_tf, p_ts, p_ls, p_ld, p_li, p_lo; DevBuf<float> p_tc, p_te, p_tfin, p_lg, p_la, p_co; DevBuf<unsigned long long> p_cnt;
  if (p_tf.alloc(tok_cap) || p_ts.alloc(tok_cap) || p_tc.alloc(tok_cap) || p_te.alloc(tok_cap) || p_tfin.alloc(tok_cap) ||
      p_ls.alloc(link_cap) || p_ld.alloc(link_cap) || p_li.alloc(link_cap) || p_lo.alloc(link_cap) || p_lg.alloc(link_cap) ||
      p_la.alloc(link_cap) || p_co.alloc(pool.cost_cap) || p_cnt.alloc(4)) return -1;
  VBH_HIP(hipMemsetAsync(p_cnt.p, 0, 4 * sizeof(unsigned long long), st));
  pool.tok_frame = p_tf.p; pool.tok_state = p_ts.p; pool.tok_cost = p_tc.p; pool.tok_extra = p_te.p; pool.tok_final = p_tfin.p;
  pool.link_src = p_ls.p; pool.link_dst = p_ld.p; pool.link_il = p_li.p; pool.link_ol = p_lo.p; pool.link_gr = p_lg.p; pool.link_ac = p_la.p;
  pool.cost_offsets = p_co.p; pool.counters = p_cnt.p;
  if (g->A > 0) {
    hipLaunchKernelGGL(fill_arc_pdf_kernel, dim3(1024), dim3(256), 0, st, g->arcs.p, g->A, d_t2p.p, g->arc_pdf.p);
    VBH_HIP(hipGetLastError());
  }
  DevBuf<Pool> d_pool;
  if (d_pool.upload(&pool, 1, st)) return -1;
  DevGraph dg{g->S, g->start, g->final_cost.p, g->arc_off.p, g->nieps.p, g->arcs.p, g->arc_pdf.p};
  DecCfg dc{d->cfg.beam, d->cfg.lattice_beam, d->cfg.beam_delta, d->cfg.acoustic_scale, d->cfg.max_active, d->cfg.min_active, d->cfg.prune_interval};
  d->h_res.resize(n_utt);
  {
    const int grid = std::min(d->small_set.n, n_utt);
    hipLaunchKernelGGL(decode_kernel, dim3(grid), dim3(NT), 0, st, dg, dc, d->small_set.d.p, d_loglikes, (int)num_pdfs,
                       d_utts.p, (const int *)nullptr, n_utt, d_counter.p, d_res.p, d_pool.p);
    VBH_HIP(hipGetLastError());
    VBH_HIP(hipMemcpyAsync(d->h_res.data(), d_res.p, sizeof(UttResult) * n_utt, hipMemcpyDeviceToHost, st));
    VBH_HIP(hipStreamSynchronize(st));
  }
  // utterances that outgrew a small slot are searched again in a big one
  std::vector<int> retry;
  for (int u = 0; u < n_utt; u++) {
    const int e = d->h_res[u].error;
    if (e == ERR_W_TOK || e == ERR_W_LINK || e == ERR_H_TOK || e == ERR_H_LINK || e == ERR_FRONT || e == ERR_CAND) retry.push_back(u);
  }
  if (!retry.empty() && getenv("VBH_DEBUG")) {
    int cnt[16] = {0};
    for (int u : retry) cnt[d->h_res[u].error & 15]++;
    fprintf(stderr, "[vbh] %zu of %d utterances overflowed a small slot:", retry.size(), n_utt);
    for (int e = 1; e < 9; e++) if (cnt[e]) fprintf(stderr, " %s x%d;", err_name(e), cnt[e]);
    fprintf(stderr, " (frames done of first: %d)\n", d->h_res[retry[0]].num_frames);
  }
  if (!retry.empty()) {
    if (d->big_set.n == 0) {
      if (alloc_slot_set(d, &d->big_set, std::min<int>(8, (int)retry.size()), /*big=*/true)) return -1;
    }
    DevBuf<int> d_list;
    if (d_list.upload(retry.data(), retry.size(), st)) return -1;
    VBH_HIP(hipMemsetAsync(d_counter.p, 0, sizeof(int), st));
    const int grid = std::min<int>(d->big_set.n, (int)retry.size());
    hipLaunchKernelGGL(decode_kernel, dim3(grid), dim3(NT), 0, st, dg, dc, d->big_set.d.p, d_loglikes, (int)num_pdfs,
                       d_utts.p, d_list.p, (int)retry.size(), d_counter.p, d_res.p, d_pool.p);
    VBH_HIP(hipGetLastError());
    VBH_HIP(hipMemcpyAsync(d->h_res.data(), d_res.p, sizeof(UttResult) * n_utt, hipMemcpyDeviceToHost, st));
    VBH_HIP(hipStreamSynchronize(st));
  }
  d->n_retried = (int)retry.size();
  unsigned long long used[4];
  VBH_HIP(hipMemcpyAsync(used, p_cnt.p, sizeof(used), hipMemcpyDeviceToHost, st));
  VBH_HIP(hipStreamSynchronize(st));
  for (int u = 0; u < n_utt; u++)
    VBH_CHECK(d->h_res[u].error == ERR_NONE, "vbh_decode_batch: utterance %d failed: %s", u, err_name(d->h_res[u].error));
  // bring the used part of the pool to the host
  const long long nt = std::min<long long>((long long)used[0], tok_cap), nl = std::min<long long>((long long)used[1], link_cap);
  std::vector<int> tf(nt), ts(nt), ls(nl), ld(nl), li(nl), lo(nl); std::vector<float> tc(nt), te(nt), tfin(nt), lg(nl), la(nl);
#define D2H(vec, src, cnt) if (cnt) VBH_HIP(hipMemcpyAsync(vec.data(), src.p, sizeof(vec[0]) * (cnt), hipMemcpyDeviceToHost, st));
  D2H(tf, p_tf, nt) D2H(ts, p_ts, nt) D2H(tc, p_tc, nt) D2H(te, p_te, nt) D2H(tfin, p_tfin, nt)
  D2H(ls, p_ls, nl) D2H(ld, p_ld, nl) D2H(li, p_li, nl) D2H(lo, p_lo, nl) D2H(lg, p_lg, nl) D2H(la, p_la, nl)
#undef D2H
  VBH_HIP(hipStreamSynchronize(st));
  Pool hp{};
  hp.tok_frame = tf.data(); hp.tok_state = ts.data(); hp.tok_cost = tc.data(); hp.tok_extra = te.data(); hp.tok_final = tfin.data();
  hp.link_src = ls.data(); hp.link_dst = ld.data(); hp.link_il = li.data(); hp.link_ol = lo.data(); hp.link_gr = lg.data(); hp.link_ac = la.data();
  d->results.resize(n_utt);
  for (int u = 0; u < n_utt; u++) finish_lattice(g, d->h_res[u], hp, &d->results[u]);
  return 0;
}

int vbh_decoder_phase_cycles(vbh_decoder *d, int32_t utt, int64_t *out8) {
  VBH_CHECK(d && out8, "vbh_decoder_phase_cycles: NULL argument");
  VBH_CHECK(utt >= 0 && utt < (int)d->h_res.size(), "vbh_decoder_phase_cycles: utterance %d out of range", utt);
  for (int i = 0; i < 8; i++) out8[i] = d->h_res[utt].t_phase[i];
  return 0;
}

int vbh_decoder_result(vbh_decoder *d, int32_t utt, vbh_lattice *out) {
  VBH_CHECK(d && out, "vbh_decoder_result: NULL argument");
  VBH_CHECK(utt >= 0 && utt < (int)d->results.size(), "vbh_decoder_result: utterance %d out of range", utt);
  *out = d->results[utt].view;
  return 0;
}

}  // extern "C"
