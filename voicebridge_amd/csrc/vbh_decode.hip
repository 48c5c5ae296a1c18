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
// Execution model.  One workgroup owns one utterance for its whole life and walks its frames;
// a batch runs as a persistent grid that pulls utterances from a counter, so there is no
// inter-workgroup communication at all.  The search is LATENCY bound (a frame has a few
// thousand tokens and arcs, spread over ~25 dependent passes), so the layout and the passes
// are built to minimise dependent memory round trips and to keep several independent loads in
// flight per lane (profiles/README.md, r01_c):
//   * everything an access needs sits in ONE 16-byte record: state {first arc, #eps arcs,
//     #arcs, final cost}; arc {next state | has-eps flag, weight, pdf, olabel} (epsilon arcs
//     first within a state); token {state, cost, extra_cost, tmp}; link {src, dst, graph,
//     acoustic} (+ arc id, read only when the lattice is written out);
//   * arcs are expanded by wave_expand(): a wave takes up to 64 tokens, prefix-sums their arc
//     counts and spreads the arcs evenly over its lanes, so a state with many arcs costs a
//     few more wave iterations instead of a serial loop on one lane;
//   * streaming passes (candidates, links, tokens, compaction) are unrolled U times with all
//     loads issued before the first use.
// Per workgroup ("slot") HBM holds
//   * ent[records of the packed graph / 2] = {key, tokid} (EntMap): key = (epoch | order-preserving bits of the best cost reaching
//     the state this frame), updated with one 64-bit atomicMin whose return value tells the
//     thread whether it is the first to reach the state (then it creates the token).  The
//     epoch (frame counter) in the high word makes entries of older frames lose every
//     comparison, so nothing is ever cleared;
//   * a raw window (W) of tokens and forward links created since the last pruning, and a
//     history (H) holding what survived earlier prunings.
// Every `prune_interval` frames the workgroup recomputes extra_costs backwards (link-parallel
// min with atomics, iterated per frame to the fixed point when epsilon links exist), stops
// at the first already-pruned frame whose values did not move, and moves the survivors of W
// into H (the sweeps count survivors per frame, so the move needs no counting pass).  The
// reference's final result does not depend on when intermediate pruning happens (its final
// pass recomputes every frame with delta = 0), so this schedule yields the same lattice.
//
// Semantics.  Costs use exactly the reference's float32 expressions and association order
// (this file is compiled with -ffp-contract=off).  The one thing a parallel search cannot replay is
// the ORDER in which ProcessEmitting visits the tokens: the reference tightens next_cutoff while it
// walks its hash list (:747-760), so which arcs above the final cutoff it admits depends on hash
// order (and on the utterances decoded before by the same object).  Its rule is a prefix minimum --
// an arc is admitted iff tot <= min(initial next_cutoff, tot' + adaptive_beam of every arc visited
// earlier) -- and a prefix minimum over a FIXED order is a scan, not a sequential loop.  The default
// here (VBH_ADMIT_ORDERED, vbh_decode_cfg::admission) applies the rule in a canonical order: arcs
// leaving tokens whose state hashes to a lower one of 64 buckets count as earlier.  That is
// oracle/vb_oracle_decode.c's exact_cutoff = 3 (bit-exact with this kernel) and statistically the
// reference's own rule: the same number of tokens created (1.005 x), WER within 0.03 abs on 512
// utterances with P(improves) 0.34 (tests/test_gpu_e2e512.py).  VBH_ADMIT_EXACT is rounds
// 1-3's order-independent subset {tot <= the frame's final next_cutoff} (oracle exact_cutoff = 1):
// 3 % fewer tokens, 7 % faster, and -- measured on the same 512 -- a slightly better search than the
// reference's, which is a systematic difference all the same.
//
// Per-frame algorithmic traffic: N_tok(20+16) + N_arc(16+4) + N_new 16 +
// N_link 24 bytes of random access; the counters are reported per utterance.
#include "vbh_internal.h"

#include <algorithm>
#include <cfloat>
#include <atomic>
#include <chrono>
#include <thread>
#include <climits>
#include <cstddef>
#include <cstring>
#include <cmath>
#include <memory>
#include <mutex>

namespace {

// Threads per utterance.  One utterance per CU (the workgroup's ~150 KB of LDS leave room for no other workgroup), so the
// thread count decides how many registers a thread gets: 1024 threads = 4 waves per SIMD = 128 VGPRs, 768 = 3 waves = 168
// VGPRs, 512 = 2 waves = 256 VGPRs.  Which one wins depends on how much the frame loop spills at 128 (r02, same-box A/B, per-
// utterance search time on the bench's speech workload): while it spilled ~230 registers, 1024 / 768 / 512 threads took
// 376 / 301 / 330 ms (and the 1024-thread build moved by 20 % from one compile to the next); with the leaner passes of the end
// of r02 (16-byte links and candidates, tokens made once by the slot scan: 44 spilled registers at 128) 229 / 243 / 284 ms.
#ifndef VBH_DECODE_NT
#define VBH_DECODE_NT 1024
#endif
constexpr int NT = VBH_DECODE_NT;    // threads per workgroup = per utterance; 1024/NT workgroups share a CU
constexpr int NW = NT / 64;
#ifndef VBH_WAVES_PER_EU
#define VBH_WAVES_PER_EU (VBH_DECODE_NT / 256)    // one workgroup per CU: 4 waves per SIMD (128 VGPRs) at 1024 threads, 3 (168) at 768, 2 (256) at 512
#endif
constexpr int WG_PER_CU = ((1024 / NT) * VBH_WAVES_PER_EU) / 4 > 0 ? ((1024 / NT) * VBH_WAVES_PER_EU) / 4 : 1;
constexpr int WBIT = 0x40000000;            // token id lives in the raw window
constexpr unsigned INF_BITS = 0x7F800000u;  // +inf as float bits
constexpr int MAX_EPS_ITERS = 4096;
constexpr int MAX_WIN_FRAMES = 512;         // frames between two compactions (prune_interval + 1 <= this); the job arrays live in the arena
constexpr int EPSBIT = (int)0x80000000;     // on an arc's next state / a token's state: that state has epsilon arcs
constexpr int PLDST = 0x40000000;           // on a token's state: the token is the destination of an epsilon link of its frame
constexpr int STMASK = 0x3FFFFFFF;
#ifndef VBH_U
#define VBH_U 1                 // (round 5, same box, ms per 1024 x 30 s: 1 / 2 / 3 items per lane in the streaming passes: 765-768 / 775-778 / 812; in round 4 it was 802 / 799 / -)
#endif
#ifndef VBH_UL
#define VBH_UL 1                // (r02 same-box A/B at 1024 threads: 1 / 2 / 4 links per lane in flight in the pruning sweep: 226-229 / 229 / 236 ms per utterance)
#endif
constexpr int U = VBH_U;                    // independent items per lane in the streaming passes
#ifndef VBH_LINK_PF
#define VBH_LINK_PF 1           // (round 5, with VBH_U = 1 / VBH_UA = 3, same box: 755.2-755.5 against 760.5-760.8 ms; in round 4 it cost registers: 1216) pruning sweeps: the next round's link records are fetched before this round's are worked on
#endif
// VBH_LAZY_DST: an emitting link keeps the provisional destination pass 2 gives it -- the destination state's slot in the frame's LDS map
// -- instead of being read and patched once the frame's tokens are numbered (a pass over every link of the frame: 9 k cycles, and every
// link's line written a second time).  The slot -> token table of a frame (Shared::grp, 4 KB) goes to a ring in the slot's memory; a pruning
// sweep loads the table of the frame above and translates on the fly; a link that SURVIVES a sweep (a handful per frame) gets its
// destination written then.  Links to second-level states (a few per cent) are listed by pass 2 and resolved as before.
#ifndef VBH_LAZY_DST
#define VBH_LAZY_DST 1
#endif
#ifndef VBH_FX
#define VBH_FX 1                // emitting links per thread in flight when their destinations are filled in
#endif
#ifndef VBH_X1
#define VBH_X1 1                // 32-bit per-thread work counters (two registers fewer alive across the frame loop)
#endif
#ifndef VBH_X2
#define VBH_X2 1                // the phase timer's last stamp stays wave-uniform (scalar registers); r04: 882.7 -> 876.7 ms with both
#endif
#ifndef VBH_UC
#define VBH_UC 1                // (r04, same-box A/B of the ordered kernel: 4 / 3 / 2 / 1 candidates per lane in pass 2: 930 / 914 / 898 / 877-880 ms per 1024 x 30 s;
#endif                          //  exact kernel 858 / - / 855 / 848: what more candidates in flight overlap costs the frame loop more in spilled registers elsewhere)
#ifndef VBH_UM
#define VBH_UM 4
#endif
#ifndef VBH_UA
#define VBH_UA 3                // (round 5, with VBH_U = 1, same box: 2 / 3 / 4 strips of 64 arcs in flight per wave in pass 1: 769-773 / 759-762 / 767-768 ms; with VBH_U = 2 it was 796 / 783 / 776)
#endif
#ifndef VBH_ADM_REFRESH
#define VBH_ADM_REFRESH 1       // (round 5: with the two prefetches 751.4-751.9 against 753.6-754.2 ms) ordered admission, pass 1: 1 = the candidate filter re-reads the buckets' bounds for every 64 x UA arcs
#endif
#ifndef VBH_ADM_LATE
#define VBH_ADM_LATE 0          // ordered admission, pass 1: 1 = an arc fetches its token's {bound, bucket} word when it needs it, not with the token's other words
#endif
constexpr int UA = VBH_UA;                  // x 64 arcs a wave has in flight in pass 1
static_assert(UA >= 1 && UA <= 4, "Shared::wx holds 64 x 4 marker bytes per wave");
constexpr int UM = VBH_UM;                  // ... in the compaction moves
constexpr int UC = VBH_UC;                  // ... in the candidate passes (LDS-bound: more only costs registers)
constexpr int UL = VBH_UL;                  // ... in the pruning sweeps over links
#ifndef VBH_CAND_PF
#define VBH_CAND_PF 1          // (round 5: with VBH_LINK_PF 753.6-754.2 against 755.2-755.5 ms) pass 2: the next round's candidates in flight while this round's are admitted
#endif
#ifndef VBH_SEL_SKIP
#define VBH_SEL_SKIP 1         // GetCutoff's radix select starts below the key bytes all of a frame's costs share (select_kth)
#endif
#ifndef VBH_HBITS
#define VBH_HBITS (VBH_DECODE_NT >= 512 ? 14 : 11)
#endif
constexpr int HBITS = VBH_HBITS;
constexpr int HCAP = 1 << HBITS;            // slots of the per-frame state map in LDS
#ifndef VBH_HPROBE
#define VBH_HPROBE 16           // (round 5, same box, ms per 1024 x 30 s: 8 / 12 / 16 / 24 / 48 / 96 probes: 778-780 / 774-776 / 772 / 777-779 / 790-794 / 824: a crowded
#endif                          //  neighbourhood is given up on sooner; the HBM level takes 3 % of the admissions either way)
constexpr int HPROBE = VBH_HPROBE;          // probes before a state is sent to the table in HBM
constexpr unsigned HEMPTY = 0xFFFFFFFFu;
#ifndef VBH_SPT
#define VBH_SPT 32              // (r04: 16 -- all 1024 threads number and write tokens, half as many rounds -- 877 -> 1000 ms: the frame loop's register allocation again)
#endif
constexpr int SPT = VBH_SPT;                // slots a thread numbers in the slot scan: 32 = one group of Shared::grp (threads 0 .. HCAP / 32 - 1 take part), 16 = half a group
constexpr int COST_CAP = 2 * HCAP;          // token costs of the current frame cached in LDS (the map's memory, idle at that time)
constexpr int FCAP = NT >= 1024 ? 1024 : 512;   // entries of each epsilon work list kept in LDS
static_assert(HCAP / SPT <= NT && (SPT == 32 || SPT == 16), "slot scan: one thread per group of 32 slots, or per half group");
// VBH_LL_LDS: the frame's row of acoustic scores (P floats, fetched coalesced at the start of the frame) is kept in LDS, so
// that the per-arc score lookup of pass 1 is an LDS read instead of a third dependent global access (token -> state -> arc
// -> score); models of more than LLCAP pdfs read the row from global memory as before
#ifndef VBH_LL_LDS
#define VBH_LL_LDS 0           // measured (r02): 1 is 15 % slower at 1024 threads -- see DESIGN.md (register allocation of the frame loop)
#endif
constexpr int LLCAP = VBH_LL_LDS ? 2560 : 1;
constexpr int GRP_RING = MAX_WIN_FRAMES + 2; // frames whose slot -> token tables are kept (VBH_LAZY_DST): every frame with links in the raw window
constexpr int DIRTY_FRAMES = 1 << 13;       // frames covered by the dirty bitmap (1 KB of LDS); later frames count as dirty

// The pruning and compaction passes run once per pruning interval; as functions of their own they get their own register
// allocation instead of competing with the per-frame passes for the kernel's 128 VGPRs (VBH_OUTLINE=0: inlined, as before)
// VBH_SURVIVOR_BITS: the pruning sweeps leave one bit per window token / link (alive after the sweep), and the compaction gathers the
// records of the set bits instead of streaming the whole raw window (300 frames x 7 k records x 16 B = 34 MB per compaction, to move
// a few hundred survivors) through the CU a second time
#ifndef VBH_SURVIVOR_BITS
#define VBH_SURVIVOR_BITS 1
#endif
#ifndef VBH_DST_BITS
#define VBH_DST_BITS VBH_SURVIVOR_BITS
#endif
// x 64 tokens of a frame (the links to tokens beyond them gather their destination as before); half as many in the builds with two
// workgroups per CU, whose LDS must stay under 80 KB (the stress variant of voicebridge_amd/build.py)
constexpr int DSTBITS_WORDS = VBH_DST_BITS ? (VBH_DECODE_NT >= 1024 ? 256 : 64) : 1;
#ifndef VBH_OUTLINE
#define VBH_OUTLINE 0          // measured (r02, tools/variants.py): 1 and 2 are 15 % slower -- the call sites cost the hot loop more registers than the callees free
#endif
// Which phase functions are functions of their own is NOT left to the inliner (r04).  prune_backward (two call sites, just under the
// inliner's size threshold) must be part of the kernel: whenever an edit pushed it over the threshold and it became a call, the frame
// loop's state had to live in the callee-saved registers across it and pass 1's inner loop got the spill code -- the "register
// lottery" of rounds 2-4 (+35-60 % kernel time from edits that had nothing to do with pass 1; spill instructions by source line:
// llvm-objdump -l of a -gline-tables-only build).  compact_window and process_nonemitting (two call sites each, far over the
// threshold) are calls: inlined as well (VBH_INLINE_ALL=3) the kernel is 110 KB and 1 % slower.  Same-box A/B, ms per 1024 x 30 s:
// this 817-819; everything inlined 826-828; prune_backward and compact_window inlined 831; process_nonemitting inlined and
// prune_backward a call 1119.
#ifndef VBH_INLINE_ALL
#define VBH_INLINE_ALL 0
#endif
#if VBH_OUTLINE
#define VBH_COLD __attribute__((noinline))
#define VBH_PRUNE_ATTR __attribute__((noinline))
#elif VBH_INLINE_ALL & 1
#define VBH_COLD __attribute__((always_inline))
#define VBH_PRUNE_ATTR __attribute__((always_inline))
#else
#define VBH_COLD __attribute__((noinline))
#define VBH_PRUNE_ATTR __attribute__((always_inline))
#endif
#if VBH_OUTLINE >= 2
#define VBH_WARM __attribute__((noinline))
#define VBH_NONEMIT_ATTR __attribute__((noinline))
#elif VBH_INLINE_ALL & 2
#define VBH_WARM __attribute__((always_inline))
#define VBH_NONEMIT_ATTR __attribute__((always_inline))
#else
#define VBH_WARM
#define VBH_NONEMIT_ATTR __attribute__((noinline))
#endif
// The per-phase functions below carry the kernel's template argument (KI, otherwise unused): every instantiation of the kernel then has
// its own copies with the same callers as when there was one kernel, so the inliner decides for each as it did then -- with shared
// copies the "only caller of a static function" rule no longer applied, other functions were inlined, and the frame loop's register
// allocation (on which its speed hangs) came out 11 % slower in BOTH instantiations (r04: 880 -> 976 ms).
#define VBH_KI template <int KI>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

// Pointers to the slot's arrays come out of LDS, where the compiler cannot know what they point
// to and would emit FLAT instructions (they complete out of order, so every wait on one becomes
// a full drain and nothing overlaps); cast to the global address space once, they give
// global_load / global_store / global_atomic.
#define VBH_GLOBAL __attribute__((address_space(1)))
template <class T> using gp = VBH_GLOBAL T *;

__device__ __forceinline__ float asf(int x) { return __int_as_float(x); }
__device__ __forceinline__ int asi(float x) { return __float_as_int(x); }
// Cache policy (VBH_NT_GRAPH / VBH_NT_LINKS): a record of the graph is read once per expansion out of hundreds of megabytes, a link is written
// once and read a pruning interval later -- neither should push the frame's candidates and tokens, read back within the frame, out of L2
#ifndef VBH_NT_GRAPH
#define VBH_NT_GRAPH 0
#endif
#ifndef VBH_NT_LINKS
#define VBH_NT_LINKS 1
#endif
#ifndef VBH_NT_EPSLINKS
#define VBH_NT_EPSLINKS 1      // the epsilon links' stores too, and ...
#endif
#ifndef VBH_NT_P1TOK
#define VBH_NT_P1TOK 1         // ... pass 1's token loads (their last reader until a pruning pass); r04, same box, three alternations: 788.0 / 791.4 / 788.4 against 798.4 / 793.7 / 796.4 ms
#endif
#ifndef VBH_NT_PRUNE
#define VBH_NT_PRUNE 1         // the pruning sweep's link loads as well (r04, ms per 1024 x 30 s: 790.2-793.2 against 795.2-796.5; profiles/README.md)
#endif

__device__ __forceinline__ i32x4 ld_graph(gp<const i32x4> p) { return VBH_NT_GRAPH ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ void st_link(gp<i32x4> p, i32x4 v) { if (VBH_NT_LINKS) __builtin_nontemporal_store(v, p); else *p = v; }

// The graph as the searches read it (built by vbh_graph_create; the 1-best search, vbh_faster.hip, reads the same arrays):
// ONE array of 16-byte records in which a state's record sits directly in front of its arcs, and a state's id IS the index of its record
// -- st and arcs both point to that array.  Expanding a token then touches one random 64-byte line (record + arcs: a state of this graph
// has 2 arcs on average; blocks of up to four records never straddle a line, larger ones start one) where the two arrays cost two or three:
// the search is bound by cache lines in flight per CU, and the graph's gathers were half of its L2 reads.  The ids are
// monotone in the graph's own, so every tie broken by state id falls as before; the graph's own id rides in the record (ordered admission
// hashes it, the lattice reports it).
struct DevGraph {
  int S, start;         // records in the array, the start state's
  const i32x4 *st;      // [S] a state's record: {the graph's own state id, #epsilon arcs, #arcs, final cost bits}; its arcs follow it
  const i32x4 *arcs;    // [S] an arc: {next state's record | EPSBIT, weight bits, pdf (filled per batch), olabel}; epsilon arcs first
  const int *arc_il;    // [S] ilabel (-1: not an arc)
};

struct DecCfg {
  float beam, lattice_beam, beam_delta, acwt;
  int max_active, min_active, prune_interval;
  int use_lds_map;      // 0: every state goes to the table in HBM (diagnostic)
  int prune_back;       // frames an intermediate pruning pass may walk back beyond the unpruned window
};

struct Ent { unsigned long long key; int tokid; int pad; };   // per graph state
// The second level of the state map.  The search's state ids are record indices of the packed graph (DevGraph), in which every state
// takes at least two records (vbh_graph_create pads a state without arcs): id >> 1 is unique per state, and the map has half as many
// entries as the graph has records -- about twice the number of states, where one per record would be four times.  (Found through the
// state's record instead, which carries the graph's own id: one entry per state, and 1.4 % of the kernel for the gather's place in pass 2
// -- 807 against 794-797 ms -- although only states that found no room in the LDS map take that path.)
struct EntMap {
  gp<Ent> e;
  __device__ __forceinline__ VBH_GLOBAL Ent &operator[](int id) const { return e[id >> 1]; }
};

struct FrameRec {
  int tok_b, tok_e;     // tokens of this frame (frame_plus_one index)
  int el_b, el_e;       // emitting links leaving this frame
  int pl_b, pl_e;       // epsilon links within this frame
  int flags;            // 1: tokens in W, 2: emitting links in W, 4: epsilon links in W
  float cost_offset;
  int a_tok, a_el, a_pl;  // survivors as of the last pruning sweep over the frame (upper bounds)
  int pad;
};

// per-slot workspace (device pointers)
struct Slot {
  Ent *ent;                                // [S]
  // raw window: tokens, remap scratch, links {src, dst, tot, arc}.  A link's graph and acoustic cost are not stored: the graph cost is
  // the arc's weight and the acoustic cost is recomputed from the frame's cost offset and the score matrix, with the expression that
  // made it, for the few links that reach the output (the scores of a batch stay resident until its lattices are out)
  i32x4 *w_tok; int *w_remap; i32x4 *wl;
  // survivor bitmaps of the raw window, one bit per token / link, written by the last pruning sweep over a frame (a wave's 64 items
  // are one word: every frame's token and link ranges start on a multiple of 64 in W): the compaction walks these words instead of
  // reading the window's 16-byte records again (VBH_SURVIVOR_BITS)
  unsigned long long *w_tok_alive, *w_lk_alive;
  unsigned long long *w_grp;               // [GRP_RING][HCAP / 32] first token << 32 | occupancy of the map's 32-slot groups, per frame (VBH_LAZY_DST)
  i32x4 *h_tok; int *h_remap; i32x4 *hl;                // history
  FrameRec *fr;                            // [max_frames + 2]
  i32x4 *cand;                             // candidates of the current frame: {tot, arc, source token, next state | EPSBIT}
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
  long long t_phase[16];             // shader-clock cycles per phase (thread 0): cutoff, pass1, pass2+3, eps, prune, compact, final; [7] frames
                                     // visited by pruning; [8..] only in -DVBH_DIAG builds (see vbh_decoder_diag)
  long long t_start, t_end;          // s_memrealtime (100 MHz) when the workgroup took the utterance / finished it
  int wg, pad2;
};

struct Pool {   // dense per-batch output
  int *tok_frame, *tok_state; float *tok_cost, *tok_extra, *tok_final;
  int *link_src, *link_dst, *link_il, *link_ol; float *link_gr, *link_ac;
  float *cost_offsets;
  unsigned long long *counters;   // [0] tokens used, [1] links used, [2] cost offsets used
  long long tok_cap, link_cap, cost_cap;
};

enum { VBH_ADMIT_EXACT = 0, VBH_ADMIT_ORDERED = 1 };
enum { ERR_NONE = 0, ERR_W_TOK = 1, ERR_W_LINK = 2, ERR_H_TOK = 3, ERR_H_LINK = 4, ERR_FRONT = 5,
       ERR_EPS_LOOP = 6, ERR_POOL = 7, ERR_CAND = 8, ERR_NOT_RUN = 9, ERR_LDS = 10, ERR_INTERNAL = 11 };

// order-preserving map float -> uint32
__device__ __forceinline__ unsigned f2o(float f) {
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float o2f(unsigned o) {
  unsigned u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
  return __uint_as_float(u);
}
__device__ __forceinline__ unsigned long long ld_u64(gp<const unsigned long long> p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int ld_i32(gp<const int> p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_i32(gp<int> p, int v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Read-modify-write atomics on per-utterance data: one workgroup owns an utterance, so
// workgroup scope is all the coherence needed.
#ifndef VBH_RMW_SCOPE
#define VBH_RMW_SCOPE __HIP_MEMORY_SCOPE_WORKGROUP
#endif
__device__ __forceinline__ unsigned long long rmw_min_u64(gp<unsigned long long> p, unsigned long long v) {
  return __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, VBH_RMW_SCOPE);
}
__device__ __forceinline__ void rmw_min_u32(gp<unsigned> p, unsigned v) {
  (void)__hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, VBH_RMW_SCOPE);
}
// make an LDS-loaded (per-lane) pointer wave-uniform again so that it lives in SGPRs, and global
template <class T>
__device__ __forceinline__ gp<T> uptr(T *p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (gp<T>)(((unsigned long long)hi << 32) | lo);
}
// per-lane pointer, global
template <class T>
__device__ __forceinline__ gp<T> gptr(T *p) { return (gp<T>)p; }
// FrameRec <-> global memory as three 16-byte vectors
__device__ __forceinline__ FrameRec ld_frame(const FrameRec *p) {
  gp<const i32x4> q = (gp<const i32x4>)p;
  const i32x4 a = q[0], b = q[1], c = q[2];
  FrameRec r;
  r.tok_b = a.x; r.tok_e = a.y; r.el_b = a.z; r.el_e = a.w; r.pl_b = b.x; r.pl_e = b.y; r.flags = b.z; r.cost_offset = asf(b.w);
  r.a_tok = c.x; r.a_el = c.y; r.a_pl = c.z; r.pad = c.w;
  return r;
}
__device__ __forceinline__ void st_frame(FrameRec *p, const FrameRec &r) {
  gp<i32x4> q = (gp<i32x4>)p;
  i32x4 a, b, c;
  a.x = r.tok_b; a.y = r.tok_e; a.z = r.el_b; a.w = r.el_e; b.x = r.pl_b; b.y = r.pl_e; b.z = r.flags; b.w = asi(r.cost_offset);
  c.x = r.a_tok; c.y = r.a_el; c.z = r.a_pl; c.w = 0;
  q[0] = a; q[1] = b; q[2] = c;
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
  base = __builtin_amdgcn_readlane(base, leader);
  return want ? base + __popcll(m & ((1ull << lane) - 1ull)) : -1;
}
// the same for N requests per lane with ONE LDS atomic; all 64 lanes must call it together
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
// inclusive prefix sum over the wave on the VALU (DPP row shifts + row broadcasts; no LDS traffic)
__device__ __forceinline__ int wave_incl_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);    // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);    // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);    // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);    // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
  return v;
}
// reductions over the wave on the VALU, result in every lane (same DPP ladder as the scan; lanes a step does not reach keep the identity)
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
  const int id = -1;      // 0xFFFFFFFF
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
__device__ __forceinline__ long long clk() { return (long long)__builtin_amdgcn_s_memtime(); }

// Load-balanced expansion inside a wave.  Every lane brings one item with `cnt` sub-items
// (0 = none); body(valid, owner, j) is then called wave-uniformly with the sub-items dealt
// out 64 at a time: lane handles sub-item j of the item that lane `owner` brought.  All 64
// lanes must call this together.
// Who owns sub-item k: the last lane whose item starts at or before k.  Found without a search: every lane whose item starts
// inside the round's window of 64 x UA sub-items leaves its number at that position in a per-wave strip of LDS bytes (`mk`,
// 64 x UA of them), every lane reads the positions it will handle, and a running maximum over the window (DPP scan, plus what
// is carried in from before the window) is the owner; the same scan over the marked positions gives the item's first
// sub-item.  (The binary search over the prefix sums it replaces was seven dependent LDS shuffles per sub-item: 12 k cycles
// of a frame's 43 k in pass 1, r02.)
__device__ __forceinline__ int wave_max_scan(int v) {      // inclusive running maximum over the lanes, values >= 0
#define VBH_DPP_MAX(ctrl, rows) { const int o_ = __builtin_amdgcn_update_dpp(0, v, ctrl, rows, 0xf, false); v = o_ > v ? o_ : v; }
  VBH_DPP_MAX(0x111, 0xf) VBH_DPP_MAX(0x112, 0xf) VBH_DPP_MAX(0x114, 0xf) VBH_DPP_MAX(0x118, 0xf) VBH_DPP_MAX(0x142, 0xa) VBH_DPP_MAX(0x143, 0xc)
#undef VBH_DPP_MAX
  return v;
}
template <int UA>
__device__ __forceinline__ void wave_owners(unsigned char *mk, int cnt, int excl, int k0, int total, bool (&valid)[UA], int (&owner)[UA], int (&j)[UA]) {
  const int lane = threadIdx.x & 63;
  if (lane < 16 * UA) reinterpret_cast<unsigned *>(mk)[lane] = 0u;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const bool has = cnt > 0;
  if (has && excl >= k0 && excl < k0 + 64 * UA) mk[excl - k0] = (unsigned char)(lane + 1);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // carried in: the last item that starts before the window (owner + 1, first sub-item + 1; 0 = none)
  const unsigned long long before = __ballot(has && excl < k0);
  int c_own = 0, c_start = 0;
  if (before) {
    const int l = __builtin_amdgcn_readfirstlane(63 - __clzll((long long)before));
    c_own = l + 1;
    c_start = __builtin_amdgcn_readlane(excl, l) + 1;
  }
  int m[UA];
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
}
template <class F>
__device__ __forceinline__ void wave_expand(unsigned char *mk, int cnt, F body) {
  const int incl = wave_incl_scan(cnt);
  const int total = __builtin_amdgcn_readlane(incl, 63);
  const int excl = incl - cnt;
  for (int k0 = 0; k0 < total; k0 += 64) {
    bool valid[1]; int owner[1], j[1];
    wave_owners<1>(mk, cnt, excl, k0, total, valid, owner, j);
    body(valid[0], owner[0], j[0]);
  }
}
// the same, UA x 64 sub-items per round: body(valid[UA], owner[UA], j[UA]) can issue the loads of all of them before it uses any
template <int UA, class F>
__device__ __forceinline__ void wave_expand_n(unsigned char *mk, int cnt, F body) {
  const int incl = wave_incl_scan(cnt);
  const int total = __builtin_amdgcn_readlane(incl, 63);
  const int excl = incl - cnt;
  for (int k0 = 0; k0 < total; k0 += 64 * UA) {
    bool valid[UA]; int owner[UA], j[UA];
    wave_owners<UA>(mk, cnt, excl, k0, total, valid, owner, j);
    body(valid, owner, j);
  }
}
// tokens/items per wave and round: spreads few items over all waves
__device__ __forceinline__ int wave_chunk(int n) {
  const int c = (n + NW - 1) / NW;
  return c < 1 ? 1 : (c > 64 ? 64 : c);
}

struct Shared {
#ifdef VBH_LDS_CANARY
  unsigned canary_lo[4];
#endif
  unsigned long long red[2][NW];   // reduction scratch, alternated so that one barrier per reduction suffices
  unsigned hist[256];
  int w_ntok, w_nlink, h_ntok, h_nlink;
  int n_cand;
  unsigned cand_bound;             // order-preserving key of the smallest tot + adaptive_beam seen so far in pass 1 (>= the final next_cutoff)
  int fcnt[3];                     // epsilon work-list fill counts, rotated with Slot::front
  int changed2[2];                 // "something changed" flags, alternated per use (see prune_frame)
  int cnt_tok[2], cnt_el[2], cnt_pl[2];   // survivors seen by a pruning sweep, alternated the same way
  int error, flag;
  unsigned sel_prefix, sel_mask; int sel_k;
  float f0;
  unsigned long long u0;
  int n_max_bound, n_min_bound, n_visited;
  int utt;
  // the block appended to H by the previous compaction (frames (tail_f0, tail_f1]); it is
  // re-compacted in place at the next one, once those frames have been pruned with look-ahead
  int tail_tok_b, tail_link_b, tail_f0, tail_f1;
  long long tph[12];               // phase cycle counters (thread 0)
#ifdef VBH_DIAG_EPS
  long long ept[12], ep_last;      // process_nonemitting by part (thread 0's clock): closure, scan + token records, second-level tokens, link fix-up, epsilon links
#endif
#ifdef VBH_DIAG
  int diag[8];
  long long fcl[8];                // cycles / frames by class of frame (tokens at its start: < 512, < 2048, < 6144, more)
  long long hph[12];               // phase cycles of the frames of the last class alone
#endif
  unsigned dirty[DIRTY_FRAMES / 32];   // see frame_dirty()
  // VBH_DST_BITS: the survivor bitmap of the tokens of the frame swept last (what its final sweep wrote to Slot::w_tok_alive), for the sweep
  // over the frame below it: an emitting link whose destination the bitmap says is dead is excised without gathering the destination's
  // record -- all but a per cent of a frame's links (dst_frame: whose tokens these are, -1: nobody's; dst_tb: their first window index)
  uint2 grp2[VBH_LAZY_DST ? HCAP / 32 : 1];  // VBH_LAZY_DST: the slot -> token table of frame grp2_frame (tokens from grp2_tb on), for the sweep over the frame below it
  int grp2_frame, grp2_tb;
  int n_l2;                                  // ... emitting links of the frame being built whose destination is a second-level state (listed in hist)
  unsigned long long dstbits[2][DSTBITS_WORDS];      // [frame & 1]: the sweeps over frame f read f + 1's bits while they write f's
  int dst_frame, dst_tb, dst_n;
  int use_lds_map;
  int scan_part[NW], scan_base;    // slot scan: occupied slots per wave, first token index it hands out
  alignas(4) unsigned char wx[NW][64 * 4];   // wave_owners: a strip of 64 x UA marker bytes per wave (UA <= 4)
  int chk[4];                      // first failed consistency check: {where, frame, value, bound}
  float ll[LLCAP];                 // the frame's acoustic scores (VBH_LL_LDS)
  int j_ntf, j_nseg, j_tf0, j_tdst, j_ldst;   // compaction job (its arrays: struct Job, in the arena)
  int front_lds[3 * FCAP];         // the first FCAP entries of the three rotating epsilon work lists
  // token index of a map slot: {occupancy of the group's 32 slots, index of the group's first token relative to the frame's first}
  uint2 grp[HCAP / 32];
  // Ordered admission (VBH_ADMIT_ORDERED): bmin[b] = key of the smallest tot + adaptive_beam over the frame's arcs that leave a token whose
  // state hashes to bucket b (adm_bucket), init_bound = the same over the best token's arcs in the reference's own association order
  // (its initial next_cutoff, lattice-faster-decoder.cc:712-724).  An arc of bucket b is admitted against min(init_bound, bmin[b'] for
  // b' < b): the reference's prefix minimum (:747-760) over a visiting order that does not depend on where tokens are stored.
  unsigned bmin[64];
  unsigned init_bound;
  // The arena.  While a frame is built (pass 2 to the end of the frame): the state map, tags [0, HCAP) and cost keys
  // [HCAP, 2 HCAP) (see map_relax).  In the cutoff phase: the frame's token costs (lds_cost).  During pruning: the link
  // minima per token (prune_frame).  During compaction: FrameRec staging and the job arrays (struct Job).
  alignas(16) unsigned arena[2 * HCAP];
#ifdef VBH_LDS_CANARY
  unsigned canary_hi[4];
#endif
};

// compaction job: token frames and link segments in source order (lives in the arena behind the FrameRec staging area)
struct Job {
  int tbeg[MAX_WIN_FRAMES + 1], tbase[MAX_WIN_FRAMES], tcnt[MAX_WIN_FRAMES], tup[MAX_WIN_FRAMES];
  int sbeg[2 * MAX_WIN_FRAMES + 1], sbase[2 * MAX_WIN_FRAMES], scnt[2 * MAX_WIN_FRAMES], sup[2 * MAX_WIN_FRAMES];
  int sid[2 * MAX_WIN_FRAMES];     // frame << 1 | (1 = epsilon links)
};
static_assert(offsetof(Shared, arena) % 16 == 0, "the arena is cleared and read with 16-byte LDS accesses");
constexpr int JOB_OFF_WORDS = ((MAX_WIN_FRAMES + 2) * 12 + 63) & ~63;
static_assert(JOB_OFF_WORDS * 4 + sizeof(Job) <= sizeof(unsigned) * 2 * HCAP, "FrameRec staging + job arrays must fit in the arena");
__device__ __forceinline__ Job &job_of(Shared &sh) { return *reinterpret_cast<Job *>(sh.arena + JOB_OFF_WORDS); }
__device__ __forceinline__ float *lds_cost(Shared &sh) { return reinterpret_cast<float *>(sh.arena); }
__device__ __forceinline__ const float *lds_cost(const Shared &sh) { return reinterpret_cast<const float *>(sh.arena); }
__device__ __forceinline__ unsigned *map_tag(Shared &sh) { return sh.arena; }
__device__ __forceinline__ const unsigned *map_tag(const Shared &sh) { return sh.arena; }
__device__ __forceinline__ unsigned *map_key(Shared &sh) { return sh.arena + HCAP; }
__device__ __forceinline__ const unsigned *map_key(const Shared &sh) { return sh.arena + HCAP; }

// ordered admission: bucket of a token's state (one bucket per lane of a wave: the prefix minimum over buckets is a DPP scan and a
// bucket's bound one cross-lane read); oracle/vb_oracle_decode.c VBO_ADM_BUCKET with adm_bucket_bits = 6
__device__ __forceinline__ int adm_bucket(int st) { return (int)(((unsigned)st * 2654435761u) >> 26); }
// inclusive running minimum over the lanes (the ladder of wave_min_u32 without its final broadcast)
__device__ __forceinline__ unsigned wave_min_scan_u32(unsigned v) {
  const int id = -1;      // 0xFFFFFFFF
#define VBH_DPP_MIN(ctrl, rows) { const unsigned o_ = (unsigned)__builtin_amdgcn_update_dpp(id, (int)v, ctrl, rows, 0xf, false); v = o_ < v ? o_ : v; }
  VBH_DPP_MIN(0x111, 0xf) VBH_DPP_MIN(0x112, 0xf) VBH_DPP_MIN(0x114, 0xf) VBH_DPP_MIN(0x118, 0xf) VBH_DPP_MIN(0x142, 0xa) VBH_DPP_MIN(0x143, 0xc)
#undef VBH_DPP_MIN
  return v;
}
// the bound of every bucket as it stands: lane b gets min(init_bound, bmin[b'] for b' < b) (a key; all 64 lanes call it together)
__device__ __forceinline__ unsigned adm_bounds(const unsigned *bmin, const unsigned *init_bound) {
  const int lane = threadIdx.x & 63;
  return wave_min_scan_u32(lane == 0 ? *init_bound : bmin[lane - 1]);
}

// block reductions with ONE barrier each: partials go to red[par], par flips on every call
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
__device__ __forceinline__ unsigned long long block_sum_u64(unsigned long long v, Shared &sh, int &par) {
  v = ((unsigned long long)wave_sum_u32((unsigned)(v >> 32)) << 32) | wave_sum_u32((unsigned)v);      // (both halves stay below 2^32: counts of a frame's tokens)
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

// cost of token i of the current frame: LDS copy, or the record itself beyond the cache
__device__ __forceinline__ float frame_cost(const Shared &sh, gp<const i32x4> tok, int i) {
  return i < COST_CAP ? lds_cost(sh)[i] : asf(tok[i].y);
}

// k-th smallest (0-based) of the current frame's costs: exact radix select on the
// order-preserving keys (the std::nth_element value in GetCutoff, lattice-faster-decoder.cc:654-676)
// lo, hi: the smallest and the largest key of the frame (VBH_SEL_SKIP).  A frame's costs lie within a few beams of each other far from zero,
// so their keys agree in the leading byte or two (sign, exponent, the top of the mantissa): the digits all keys share are the
// statistic's as well, and the select starts below them -- two of its four passes (three barriers and a sweep over the costs each) in a
// typical frame.  Exact: only passes whose histogram would have had a single occupied bin are left out.
VBH_KI __device__ VBH_WARM float select_kth(gp<const i32x4> tok, int n, int k, Shared &sh, unsigned lo, unsigned hi) {
  const int tid = threadIdx.x;
  int shift0 = 24;
  if (VBH_SEL_SKIP) while (shift0 > 0 && (lo >> shift0) == (hi >> shift0)) shift0 -= 8;
  __syncthreads();
  if (tid == 0) {
    const unsigned m0 = shift0 == 24 ? 0u : ~0u << (shift0 + 8);
    sh.sel_prefix = lo & m0; sh.sel_mask = m0; sh.sel_k = k;
  }
  for (int shift = shift0; shift >= 0; shift -= 8) {
    for (int i = tid; i < 256; i += NT) sh.hist[i] = 0;
    __syncthreads();
    const unsigned prefix = sh.sel_prefix, mask = sh.sel_mask;
    // four costs per thread and LDS read (the cost list is contiguous in the arena), their histogram adds in flight together
    {
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      const int nl = n < COST_CAP ? n : COST_CAP, n4 = nl & ~3;
      const f32x4 *c4 = reinterpret_cast<const f32x4 *>(lds_cost(sh));
      for (int i = tid * 4; i < n4; i += NT * 4) {
        const f32x4 c = c4[i >> 2];
        const unsigned k0 = f2o(c.x), k1 = f2o(c.y), k2 = f2o(c.z), k3 = f2o(c.w);
        if ((k0 & mask) == prefix) atomicAdd(&sh.hist[(k0 >> shift) & 255u], 1u);
        if ((k1 & mask) == prefix) atomicAdd(&sh.hist[(k1 >> shift) & 255u], 1u);
        if ((k2 & mask) == prefix) atomicAdd(&sh.hist[(k2 >> shift) & 255u], 1u);
        if ((k3 & mask) == prefix) atomicAdd(&sh.hist[(k3 >> shift) & 255u], 1u);
      }
      for (int i = n4 + tid; i < n; i += NT) {
        const unsigned key = f2o(frame_cost(sh, tok, i));
        if ((key & mask) == prefix) atomicAdd(&sh.hist[(key >> shift) & 255u], 1u);
      }
    }
    __syncthreads();
    if (tid < 64) {     // wave 0: 4 bins per lane, prefix over lanes, then the bin inside the lane
      const int kk = sh.sel_k;
      const int h0 = (int)sh.hist[4 * tid], h1 = (int)sh.hist[4 * tid + 1], h2 = (int)sh.hist[4 * tid + 2], h3 = (int)sh.hist[4 * tid + 3];
      const int mine = h0 + h1 + h2 + h3;
      const int incl = wave_incl_scan(mine);
      const unsigned long long m = __ballot(incl > kk);
      const int owner = m ? __ffsll((long long)m) - 1 : 63;   // k < n always, so some lane qualifies
      if (tid == owner) {
        int r = kk - (incl - mine), b = 0;
        if (r >= h0) { r -= h0; b = 1; if (r >= h1) { r -= h1; b = 2; if (r >= h2) { r -= h2; b = 3; } } }
        sh.sel_k = r;
        sh.sel_prefix = prefix | ((unsigned)(4 * tid + b) << shift);
        sh.sel_mask = mask | (255u << shift);
      }
    }
    __syncthreads();
  }
  return o2f(sh.sel_prefix);
}

__device__ __forceinline__ gp<i32x4> tokp(gp<i32x4> wt, gp<i32x4> ht, int id) { return (id & WBIT) ? wt + (id ^ WBIT) : ht + id; }

// Consistency checks on indices that are about to address memory: a violation is recorded (first one wins), the utterance
// fails with ERR_INTERNAL instead of the device faulting.
#ifndef VBH_EPS_RELAX
#define VBH_EPS_RELAX 2        // the epsilon links of a frame relaxed on their own instead of sweeping the frame once per level (prune_frame): 1 = any number of them,
                               // 2 = up to 64 (one wave, no barriers; more: the sweeps).  r04 same-box A/B, ms per 1024 x 30 s: 0: 850.2, 1: 1172 (the frame loop's
                               // register allocation: 125 spills) / 838.5 as a function call, 2: 814.2 / 820.6 as a function call
#endif
#ifndef VBH_SWEEP_BITS
#define VBH_SWEEP_BITS VBH_SURVIVOR_BITS      // the pruning sweeps record excised window links in the survivor bitmap only (eval_links)
#endif
// VBH_SPARSE_TOK (r05): the token pass of a pruning sweep over a frame of the raw window touches the records of the SURVIVORS only.  A window
// frame is swept exactly once while it is in the window (the pass that ends with its compaction visits every window frame once, top down),
// so every token's old extra_cost is the 0 it was created with and need not be read; and a token that dies -- all but a handful per frame
// -- is recorded in the survivor bitmap alone: its readers are the sweep over the frame below (sh.dstbits) and the compaction
// (Slot::w_tok_alive), both of which consult the bitmap before the record.  Before: 16 bytes read and a 4-byte store into every token
// record of the frame, i.e. every line of the frame's tokens fetched and written back once more, 300 frames after the search had left them
// (a sixth of the kernel's memory traffic).  Frames the bitmap in LDS does not cover entirely, or whose epsilon links are swept the old
// way, keep the old pass (their records stay exact).
#ifndef VBH_SPARSE_TOK
#define VBH_SPARSE_TOK 1
#endif
#ifndef VBH_CHECKS
#define VBH_CHECKS 0           // (diagnostic builds: -DVBH_CHECKS=1; measured r02: 20 % slower, the compares cost registers the frame loop does not have)
#endif
__device__ __forceinline__ bool chk_fail(Shared &sh, int where, int value, int bound) {
  if (atomicCAS(&sh.chk[0], 0, where) == 0) { sh.chk[2] = value; sh.chk[3] = bound; }
  sh.error = ERR_INTERNAL;
  return false;
}
#if VBH_CHECKS
#define VBH_CHK(sh, where, value, bound) (((unsigned)(value) < (unsigned)(bound)) ? true : chk_fail(sh, where, value, bound))
#else
#define VBH_CHK(sh, where, value, bound) true
#endif
__device__ __forceinline__ bool chk_tok_id(Shared &sh, const Slot &s, int where, int id) {
  return (id & WBIT) ? VBH_CHK(sh, where, id ^ WBIT, s.w_tok_cap) : VBH_CHK(sh, where + 1, id, s.h_tok_cap);
}

// evaluate the forward links [lb, le) of one frame: excise those over the lattice beam,
// atomicMin the rest into the source token's tmp (lattice-faster-decoder.cc:331-356).  A link
// carries tot = (source cost + acoustic) + graph, so only the destination token is gathered.
// Links at positions < split are the frame's epsilon links, the others its emitting links.
// Returns (valid on lane 0 of each wave) the links this wave kept: epsilon << 32 | emitting.
template <bool LDS_TMP>
__device__ __forceinline__ unsigned long long eval_links(gp<i32x4> lk, int lb, int le_, int split, gp<i32x4> wt, gp<i32x4> ht,
                                                         float lattice_beam, unsigned *lds_tmp, int tb, Shared &sh, const Slot &s,
                                                         gp<unsigned long long> alive_bits = nullptr, bool bits_valid = false,
                                                         int lds_off = 0, int lds_cap = 128, const unsigned long long *dst_bits = nullptr, int dst_tb = 0, int dst_n = 0,
                                                         const uint2 *grp2 = nullptr, int grp_tb = 0, bool eps_skip = false) {
  // With a survivor bitmap (links of the raw window; VBH_SWEEP_BITS): the bitmap IS the record of which links are excised -- nothing is
  // written into the link records (a scattered 4-byte store per excised link: whole 64-byte half-lines written back for it), and from
  // the second sweep of a visit on (bits_valid) the links the bitmap says are gone are not read again.  The words of the frame being
  // swept are kept in LDS (sh.hist, 128 words = 8192 links; beyond that from the bitmap in HBM) between the sweeps of a visit.
  const bool use_bits = VBH_SWEEP_BITS && alive_bits != nullptr;
  unsigned long long *lbits = reinterpret_cast<unsigned long long *>(sh.hist) + lds_off;      // (a frame with its links in two ranges: half of the words each)
  unsigned long long kept = 0;
  // the records of one round: is the link there (and, from a visit's second sweep on, alive by the bitmap), its record
  auto fetch = [&](int base, i32x4 (&Lx)[UL], bool (&okx)[UL]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < UL; u++) {
      const int l = base + u * NT + (int)threadIdx.x;
      okx[u] = l < le_ && !(eps_skip && l < split);      // (eps_skip: the epsilon links are left to relax_eps_links; their bitmap bits are written as 0 here)
      if (use_bits && bits_valid) {
        const int wfirst = base + u * NT + ((int)threadIdx.x & ~63), wi = (wfirst - lb) >> 6;
        const unsigned long long word = wfirst < le_ ? (wi < lds_cap ? lbits[wi] : alive_bits[wfirst >> 6]) : 0ull;
        okx[u] = okx[u] && ((word >> ((int)threadIdx.x & 63)) & 1ull);
      }
      Lx[u] = VBH_NT_PRUNE ? __builtin_nontemporal_load(lk + (okx[u] ? l : lb)) : lk[okx[u] ? l : lb];
    }
  };
#if VBH_LINK_PF
  i32x4 Lp[UL]; bool okp[UL];       // the next round's records: in flight while this round's are worked on
  if (lb < le_) fetch(lb, Lp, okp);
#endif
  for (int base = lb; base < le_; base += NT * UL) {
    i32x4 L[UL]; bool ok[UL];
#if VBH_LINK_PF
#pragma unroll
    for (int u = 0; u < UL; u++) { L[u] = Lp[u]; ok[u] = okp[u]; }
    if (base + NT * UL < le_) fetch(base + NT * UL, Lp, okp);
#else
    fetch(base, L, ok);
#endif
    bool prov[UL];       // VBH_LAZY_DST: the link still carries its destination's map slot (grp2: the slot -> token table of the frame above)
#pragma unroll
    for (int u = 0; u < UL; u++) {
      ok[u] = ok[u] && L[u].x >= 0;   // negative source: already excised
      prov[u] = false;
      if (VBH_LAZY_DST && grp2 != nullptr) {
        const int l = base + u * NT + (int)threadIdx.x, y = L[u].y;
        prov[u] = ok[u] && l >= split && y >= 0 && !(y & WBIT);
        if (prov[u]) { const uint2 gq = grp2[y >> 5]; L[u].y = (grp_tb + (int)gq.y + __popc(gq.x & ((1u << (y & 31)) - 1u))) | WBIT; }
      }
      if (ok[u]) {
        ok[u] = chk_tok_id(sh, s, 10, L[u].y) && chk_tok_id(sh, s, 12, L[u].x);
        if (LDS_TMP && ok[u]) ok[u] = VBH_CHK(sh, 14, (L[u].x & ~WBIT) - tb, COST_CAP);
      }
    }
    // (dst_bits: the emitting links' destinations are the window tokens [dst_tb, dst_tb + dst_n) of the frame above, whose survivor bits are
    // in sh.dstbits: a link to a dead token has link_extra_cost = +inf (:338) and goes without its destination's record being read)
    if (VBH_DST_BITS && dst_bits) {
#pragma unroll
      for (int u = 0; u < UL; u++) {
        const int l = base + u * NT + (int)threadIdx.x;
        const unsigned di = (unsigned)((L[u].y ^ WBIT) - dst_tb);
        if (ok[u] && l >= split && di < (unsigned)dst_n) ok[u] = (dst_bits[di >> 6] >> (di & 63)) & 1ull;
      }
    }
    i32x4 D[UL];
#pragma unroll
    for (int u = 0; u < UL; u++) {
      if (VBH_DST_BITS) { D[u].x = 0; D[u].y = 0; D[u].z = 0; D[u].w = 0; if (ok[u]) D[u] = *tokp(wt, ht, L[u].y); }
      else { gp<const i32x4> dp = ok[u] ? tokp(wt, ht, L[u].y) : ht; D[u] = *dp; }
    }
#pragma unroll
    for (int u = 0; u < UL; u++) {
      const int l = base + u * NT + (int)threadIdx.x;
      // next_tok->extra_cost + ((tok->tot_cost + link->acoustic_cost + link->graph_cost) - next_tok->tot_cost)
      float le = asf(D[u].z) + (asf(L[u].z) - asf(D[u].y));
      const bool keep = ok[u] && !(le > lattice_beam);
      if (ok[u] && !keep && !use_bits) ((gp<int>)&lk[l])[0] = -1;
      if (VBH_LAZY_DST && keep && prov[u]) ((gp<int>)&lk[l])[1] = L[u].y;       // a survivor: from now on the link names its destination token
      if (keep) {
        if (le < 0.0f) le = 0.0f;
        if (LDS_TMP) atomicMin(&lds_tmp[(L[u].x & ~WBIT) - tb], __float_as_uint(le));
        else rmw_min_u32((gp<unsigned>)tokp(wt, ht, L[u].x) + 3, __float_as_uint(le));
      }
      const unsigned long long m = __ballot(keep);
      // a wave's 64 links are consecutive, so the split falls on a lane boundary
      const int first = base + u * NT + ((int)threadIdx.x & ~63);
#if VBH_SURVIVOR_BITS
      if (alive_bits && ((int)threadIdx.x & 63) == 0 && first < le_) {      // (lb is a multiple of 64 in W)
        alive_bits[first >> 6] = m;
        if (use_bits && ((first - lb) >> 6) < lds_cap) lbits[(first - lb) >> 6] = m;
      }
#endif
      const int ne = split - first;   // lanes [0, ne) hold epsilon links
      const unsigned long long em = ne <= 0 ? 0ull : (ne >= 64 ? ~0ull : ((1ull << ne) - 1ull));
      kept += ((unsigned long long)__popcll(m & em) << 32) | (unsigned long long)__popcll(m & ~em);
    }
  }
  return kept;
}

// The epsilon links [pb, pe) of a frame whose tokens' emitting-link minima are in lds_tmp (all of them, behind a barrier): relax them to
// the fixed point (see prune_frame), then record which of them survive -- in the bitmap (window links; eval_links left their bits 0) or
// in the records (history links) -- and add their number to *cnt.  Source and destination both belong to the frame [tb, te).
#ifndef VBH_RELAX_ATTR
#define VBH_RELAX_ATTR __forceinline__
#endif
__device__ VBH_RELAX_ATTR void relax_eps_links(gp<i32x4> plk, int pb, int pe, gp<i32x4> wt, gp<i32x4> ht, float lattice_beam, unsigned *lds_tmp,
                                                int tb, int te, Shared &sh, const Slot &s, gp<unsigned long long> pbits, int *cnt) {
  const int tid = threadIdx.x, ne = pe - pb;
  // "did anybody's link move a token" across the workgroup with one barrier per round: three flags in rotation (round k sets and reads
  // flag k % 3 around its barrier and clears flag (k + 2) % 3, which nobody touches before the barrier of round k + 1); all three are 0 on
  // entry (prune_frame) and on exit.  (select_kth's words, idle during pruning: __syncthreads_or would take LDS of its own, which the
  // two-workgroups-per-CU build does not have)
  int *flag3 = reinterpret_cast<int *>(&sh.sel_prefix);
  int round = 0;
  auto block_or = [&](bool ch) -> bool {
    if (ch) flag3[round] = 1;
    __syncthreads();
    const bool any = flag3[round] != 0;
    if (tid == 0) flag3[round >= 1 ? round - 1 : 2] = 0;
    round = round == 2 ? 0 : round + 1;
    return any;
  };
  // one link: its record, the cost of its destination; a = (tok->tot_cost + acoustic + graph) - next_tok->tot_cost (:337)
  auto load = [&](int l, bool mine, int &si, int &di, float &a) -> bool {
    const i32x4 L = plk[mine ? l : pb];
    bool ok = mine && L.x >= 0;
    if (ok) ok = chk_tok_id(sh, s, 16, L.y) && chk_tok_id(sh, s, 18, L.x);
    si = (L.x & ~WBIT) - tb; di = (L.y & ~WBIT) - tb;
    if (ok) ok = VBH_CHK(sh, 22, si, te - tb) && VBH_CHK(sh, 23, di, te - tb);
    gp<const i32x4> dp = ok ? tokp(wt, ht, L.y) : ht;
    a = asf(L.z) - asf(((gp<const int>)dp)[1]);
    return ok;
  };
  auto step = [&](bool ok, int si, int di, float a, bool &keep) -> bool {
    float le = asf((int)lds_tmp[ok ? di : 0]) + a;       // next_tok->extra_cost as it stands + a
    keep = ok && !(le > lattice_beam);
    if (!keep) return false;
    if (le < 0.0f) le = 0.0f;
    const unsigned v = __float_as_uint(le);
    return atomicMin(&lds_tmp[si], v) > v;
  };
  auto record = [&](int l, bool ok, bool keep) {
    if (pbits) { if (keep) (void)__hip_atomic_fetch_or(pbits + (l >> 6), 1ull << (l & 63), __ATOMIC_RELAXED, VBH_RMW_SCOPE); }
    else if (ok && !keep) ((gp<int>)&plk[l])[0] = -1;
    const unsigned long long m = __ballot(keep);
    if ((tid & 63) == 0 && m) atomicAdd(cnt, __popcll(m));
  };
#if VBH_EPS_RELAX == 2
  if (tid < 64) {       // (ne <= 64: prune_frame)
    int si, di; float a; bool keep;
    const bool ok = load(pb + tid, tid < ne, si, di, a);
    while (__ballot(step(ok, si, di, a, keep)) != 0ull) {}
    record(pb + tid, ok, keep);
  }
  return;
#endif
  if (ne <= NT) {
    // a link per thread, held in registers; up to 64 links: wave 0 alone, without barriers
    if (ne > 64 || tid < 64) {
      int si, di; float a; bool keep;
      const bool ok = load(pb + tid, tid < ne, si, di, a);
      for (;;) {
        const bool ch = step(ok, si, di, a, keep);
        const bool any = ne <= 64 ? __ballot(ch) != 0ull : block_or(ch);
        if (!any) break;
      }
      record(pb + tid, ok, keep);
    }
  } else {
    for (;;) {
      bool ch = false;
      for (int l = pb + tid; l < pe; l += NT) { int si, di; float a; bool keep; const bool ok = load(l, true, si, di, a); ch |= step(ok, si, di, a, keep); }
      if (!block_or(ch)) break;
    }
    for (int base = pb; base < pe; base += NT) {
      const int l = base + tid;
      int si, di; float a; bool keep;
      const bool ok = load(l, l < pe, si, di, a);
      (void)step(ok, si, di, a, keep);      // (at the fixed point: changes nothing)
      record(l, ok, keep);
    }
  }
}

// frames whose links must be re-evaluated because the tokens of the next frame changed since
// they were last evaluated (or they never were).  One bit per frame in LDS; frames beyond the
// bitmap always count as dirty.  Thread 0 writes, everybody reads after a barrier.
__device__ __forceinline__ bool frame_dirty(const Shared &sh, int f) {
  return f >= DIRTY_FRAMES || ((sh.dirty[f >> 5] >> (f & 31)) & 1u);
}
__device__ __forceinline__ void set_dirty(Shared &sh, int f, bool d) {
  if (f < 0 || f >= DIRTY_FRAMES) return;
  if (d) sh.dirty[f >> 5] |= 1u << (f & 31); else sh.dirty[f >> 5] &= ~(1u << (f & 31));
}

// PruneForwardLinks for one frame (lattice-faster-decoder.cc:300-371), evaluated to its exact
// fixed point (the frame's own iteration does not use delta, so that a frame whose successor has
// not moved since is known to be settled).  tmp of the frame's tokens must hold +inf on entry
// and holds +inf again on exit.  Leaves the frame's survivor counts in its FrameRec.  Returns
// (uniformly) bit 0: some extra_cost moved by more than delta (the reference's "changed", :358),
// bit 1: some extra_cost moved at all.
VBH_KI __device__ int prune_frame(const Slot &s, int f, float lattice_beam, float delta, Shared &sh, int &pp) {
  const int tid = threadIdx.x;
  FrameRec *frp = s.fr + f;
  const FrameRec fr = ld_frame(frp);
  const int tb = fr.tok_b, te = fr.tok_e, eb = fr.el_b, ee = fr.el_e, pb = fr.pl_b, pe = fr.pl_e;
  const int flags = fr.flags;
  gp<i32x4> wt = uptr(s.w_tok), ht = uptr(s.h_tok);
  gp<i32x4> tok = uptr((flags & 1) ? s.w_tok : s.h_tok);
  gp<i32x4> elk = uptr((flags & 2) ? s.wl : s.hl), plk = uptr((flags & 4) ? s.wl : s.hl);
  // epsilon links of a frame are created right before its emitting links: normally one range
  const bool one_range = pe == eb && ((flags >> 1) & 1) == ((flags >> 2) & 1);
#if VBH_CHECKS
  {   // (uniform: everybody read the same record)
    const int tcap = (flags & 1) ? s.w_tok_cap : s.h_tok_cap, ecap = (flags & 2) ? s.w_link_cap : s.h_link_cap, pcap = (flags & 4) ? s.w_link_cap : s.h_link_cap;
    const bool good = tb >= 0 && tb <= te && te <= tcap && eb >= 0 && eb <= ee && ee <= ecap && pb >= 0 && pb <= pe && pe <= pcap;
    if (!good) { if (threadIdx.x == 0) { chk_fail(sh, 20, f, te); } __syncthreads(); return 0; }
  }
#endif
  // Frames of up to COST_CAP tokens accumulate the link minima in LDS (sh.cost, all +inf between
  // sweeps) instead of the tokens' tmp fields: no global atomics to wait for, and the old
  // extra_costs are fetched while the links are in flight.
  const bool lds_tmp = te - tb <= COST_CAP;
  unsigned *ltmp = sh.arena;
  int result = 0;
#if VBH_SURVIVOR_BITS
  // (window ranges start on multiples of 64; a frame whose links lie in two arrays has its emitting links at the start of W)
  gp<unsigned long long> tbits = (flags & 1) ? uptr(s.w_tok_alive) : nullptr;
  gp<unsigned long long> ebits = (flags & 2) ? uptr(s.w_lk_alive) : nullptr, pbits = (flags & 4) ? uptr(s.w_lk_alive) : nullptr;
#else
  gp<unsigned long long> tbits = nullptr, ebits = nullptr, pbits = nullptr;
#endif
  // VBH_LAZY_DST: the slot -> token table of frame f + 1, for this frame's emitting links while they are in the raw window.  It is in LDS
  // already when frame f + 1 was the last one swept (it left its table there on its way out) or this is the first frame of the pass
  // (prune_backward); otherwise it is fetched now
  const uint2 *g2 = nullptr;
  int g2_tb = 0;
  if (VBH_LAZY_DST && (flags & 2) && sh.use_lds_map) {
    // (uniform: everybody reads the word its last writer left behind a barrier -- and thread 0 must not write it again before everybody
    // HAS read it: without the first barrier a late wave saw the new value, skipped the block and its barrier, and translated slots with a
    // table that was not there yet; found by the two-workgroups-per-CU build, tests/test_gpu_variant.py, as a memory fault)
    const bool fetch = sh.grp2_frame != f + 1;
    if (fetch) {
      __syncthreads();
      gp<const unsigned long long> src = uptr(s.w_grp) + (size_t)((f + 1) % GRP_RING) * (HCAP / 32);
      if (tid < HCAP / 32) { const unsigned long long w_ = src[tid]; uint2 gq; gq.x = (unsigned)w_; gq.y = (unsigned)(w_ >> 32); sh.grp2[tid] = gq; }
      if (tid == 0) { sh.grp2_frame = f + 1; sh.grp2_tb = ((gp<const int>)(s.fr + f + 1))[0]; }
      __syncthreads();
    }
    g2 = sh.grp2; g2_tb = sh.grp2_tb;
  }
  // ... and this frame's own table, for the sweep over frame f - 1: fetched now, put into LDS on the way out (when nobody reads f + 1's any more)
  unsigned long long own_g = 0;
  const bool leave_table = VBH_LAZY_DST && (flags & 1) && sh.use_lds_map;
  if (leave_table && tid < HCAP / 32) own_g = (uptr(s.w_grp) + (size_t)(f % GRP_RING) * (HCAP / 32))[tid];
  // the survivor bits of the tokens the emitting links point to (frame f + 1, swept just before this one), if they are in LDS (VBH_DST_BITS)
  const bool d_ok = VBH_DST_BITS && ebits != nullptr && sh.dst_frame == f + 1;
  const unsigned long long *dbits = d_ok ? sh.dstbits[(f + 1) & 1] : nullptr;
  const int d_tb = sh.dst_tb, d_n = sh.dst_n;
  unsigned long long *mybits = sh.dstbits[f & 1];
  // VBH_EPS_RELAX (frames with their link minima in LDS): ONE sweep per visit.  The emitting links (their destinations, frame f + 1, do not
  // move during the visit) leave E(t) = the minimum over a token's emitting links in ltmp; the frame's epsilon links -- a handful per frame
  // in an HCLG -- are then relaxed on top of that, ltmp[src] = min(ltmp[src], ltmp[dst] + (tot - cost[dst])), until nothing moves: the
  // same fixed point as sweeping the whole frame again for every level of epsilon links (:300-371 iterates the frame; 1.7 sweeps per visit
  // measured, each with the latency of a full link pass), reached from above, so every step is an atomicMin.
  const bool relax = VBH_EPS_RELAX && lds_tmp && (VBH_EPS_RELAX != 2 || pe - pb <= 64);
  const int iters_max = relax ? 1 : MAX_EPS_ITERS;
  // VBH_SPARSE_TOK: a window frame swept in one sweep, all of whose tokens the bitmap in LDS covers
  const bool sparse = VBH_SPARSE_TOK && VBH_DST_BITS && tbits != nullptr && relax && (te - tb) <= 64 * DSTBITS_WORDS;
  if (relax && pe - pb > 64 && tid == 0) { sh.sel_prefix = 0; sh.sel_mask = 0; sh.sel_k = 0; }      // relax_eps_links' flags (ordered by the barrier behind the link pass)
  for (int iter = 0; iter < iters_max; iter++) {
    i32x4 R0[U];            // first round of the token pass, prefetched
    if (lds_tmp && !sparse) {
#pragma unroll
      for (int u = 0; u < U; u++) { const int t = tb + u * NT + tid; R0[u] = tok[t < te ? t : tb]; }
    }
    unsigned long long kept;
    if (lds_tmp) {
      if (one_range) kept = eval_links<true>(elk, pb, ee, pe, wt, ht, lattice_beam, ltmp, tb, sh, s, ebits, iter > 0, 0, 128, dbits, d_tb, d_n, g2, g2_tb, relax);
      else {
        kept = (relax && !pbits) ? 0ull : eval_links<true>(plk, pb, pe, pe, wt, ht, lattice_beam, ltmp, tb, sh, s, pbits, iter > 0, 0, 64, nullptr, 0, 0, nullptr, 0, relax);
        kept += eval_links<true>(elk, eb, ee, eb, wt, ht, lattice_beam, ltmp, tb, sh, s, ebits, iter > 0, 64, 64, dbits, d_tb, d_n, g2, g2_tb);
      }
    } else {
      if (one_range) kept = eval_links<false>(elk, pb, ee, pe, wt, ht, lattice_beam, ltmp, tb, sh, s, ebits, iter > 0, 0, 128, dbits, d_tb, d_n, g2, g2_tb);
      else {
        kept = eval_links<false>(plk, pb, pe, pe, wt, ht, lattice_beam, ltmp, tb, sh, s, pbits, iter > 0, 0, 64);
        kept += eval_links<false>(elk, eb, ee, eb, wt, ht, lattice_beam, ltmp, tb, sh, s, ebits, iter > 0, 64, 64, dbits, d_tb, d_n, g2, g2_tb);
      }
    }
    if ((tid & 63) == 0) {
      if ((unsigned)kept) atomicAdd(&sh.cnt_el[pp], (int)(unsigned)kept);
      if (kept >> 32) atomicAdd(&sh.cnt_pl[pp], (int)(kept >> 32));
    }
    __syncthreads();
    // the [pp] entries are written below / above and read after the next barrier; the [pp^1]
    // entries (read by their consumers before they arrived at the barrier above) are cleared for
    // the following use
    if (tid == 0) { sh.changed2[pp ^ 1] = 0; sh.cnt_tok[pp ^ 1] = 0; sh.cnt_el[pp ^ 1] = 0; sh.cnt_pl[pp ^ 1] = 0; }
    if (relax && pe > pb) {      // (uniform)
      relax_eps_links(plk, pb, pe, wt, ht, lattice_beam, ltmp, tb, te, sh, s, pbits, &sh.cnt_pl[pp]);
      __syncthreads();
    }
    int ch = 0, alive = 0;
    for (int base = tb; base < te; base += NT * U) {
      i32x4 R[U]; bool ok[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int t = base + u * NT + tid;
        ok[u] = t < te;
        if (sparse) { R[u].x = 0; R[u].y = 0; R[u].z = 0; R[u].w = 0; }      // (old extra_cost: what the token was created with; the record is not read)
        else if (lds_tmp && base == tb) R[u] = R0[u];
        else R[u] = tok[ok[u] ? t : tb];
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int t = base + u * NT + tid;
        unsigned v = (unsigned)R[u].w;
        const unsigned o = (unsigned)R[u].z;
        if (lds_tmp) {
          v = INF_BITS;
          if (ok[u]) { v = ltmp[t - tb]; if (v != INF_BITS) ltmp[t - tb] = INF_BITS; }
          if (ok[u] && v != o && !(sparse && v == INF_BITS)) ((gp<int>)&tok[t])[2] = (int)v;      // (sparse: a dead token's record is left alone)
        } else if (ok[u] && (v != o || v != INF_BITS)) {
          i32x2 nv; nv.x = (int)v; nv.y = (int)INF_BITS;
          *(gp<i32x2>)((gp<int>)&tok[t] + 2) = nv;
        }
        if (ok[u] && v != o) {
          ch |= 2;
          if (R[u].x & PLDST) ch |= 4;
          // "changed" as the reference counts it (:358): |new - old| > delta; inf vs inf is no change
          if (fabsf(__uint_as_float(v) - __uint_as_float(o)) > delta) ch |= 1;
        }
        const unsigned long long am = __ballot(ok[u] && v != INF_BITS);
        alive += __popcll(am);
#if VBH_SURVIVOR_BITS
        { const int first = base + u * NT + (tid & ~63);      // (tb is a multiple of 64 in W)
          if (tbits && (tid & 63) == 0 && first < te) { tbits[first >> 6] = am; if (VBH_DST_BITS && ((first - tb) >> 6) < DSTBITS_WORDS) mybits[(first - tb) >> 6] = am; } }
#endif
      }
    }
    if (ch) atomicOr(&sh.changed2[pp], ch);
    if ((tid & 63) == 0 && alive) atomicAdd(&sh.cnt_tok[pp], alive);
    __syncthreads();
    const int any = sh.changed2[pp];
    // another sweep can only differ if a token that some epsilon link of this frame points to moved
    const bool last = relax || !(any & 4) || pe == pb;
    if (last && tid == 0) {
      i32x4 c; c.x = sh.cnt_tok[pp]; c.y = sh.cnt_el[pp]; c.z = sh.cnt_pl[pp]; c.w = 0;
      ((gp<i32x4>)frp)[2] = c;
    }
    pp ^= 1;
    result |= any;
#ifdef VBH_DIAG
    if (tid == 0) sh.diag[4]++;        // sweeps (a frame with epsilon links is swept until its destinations stop moving: 1.9 per visit)
#endif
    if (last) break;
  }
  if (leave_table && tid < HCAP / 32) { uint2 gq; gq.x = (unsigned)own_g; gq.y = (unsigned)(own_g >> 32); sh.grp2[tid] = gq; }      // (everybody is past the last sweep's link pass: the barrier behind it)
  if (tid == 0) {
    if (VBH_LAZY_DST) { sh.grp2_frame = leave_table ? f : -1; sh.grp2_tb = tb; }
    set_dirty(sh, f, false); if (result & 2) set_dirty(sh, f - 1, true);
    // whose survivor bits sh.dstbits[f & 1] now holds (window tokens only: the bitmap is the window's)
    if (VBH_DST_BITS) { sh.dst_frame = tbits ? f : -1; sh.dst_tb = tb; sh.dst_n = (te - tb) < 64 * DSTBITS_WORDS ? (te - tb) : 64 * DSTBITS_WORDS; }
  }
  return result;
}

// backward pass from frame `from` down (PruneActiveTokens :503-530).  Intermediate passes
// (delta > 0): frames >= first_unpruned are always processed, older ones while the frame above
// moved by more than delta, and never below stop_at.  Whatever such a pass leaves unpropagated is
// remembered as a dirty frame.  The final pass (delta = 0, final = true) visits exactly the
// frames that are dirty or become so on the way down: a frame whose successor has not moved
// since its last evaluation would be recomputed to the same values.  extra_costs only grow as
// the frontier advances, so values left stale by an intermediate pass are lower bounds and
// pruning with them never removes anything the final pass keeps.
VBH_KI __device__ VBH_PRUNE_ATTR void prune_backward(const Slot &s, int from, int first_unpruned, int stop_at, float lattice_beam, float delta,
                               bool final, Shared &sh, int &pp) {
  int visited = 0;
  int above = 3;   // result of the frame above: the frontier counts as changed
  for (int i = threadIdx.x; i < COST_CAP; i += NT) sh.arena[i] = INF_BITS;   // see prune_frame
  if (VBH_DST_BITS && threadIdx.x == 0) sh.dst_frame = -1;                   // (no frame's survivor bits are in LDS yet)
  if (VBH_LAZY_DST && threadIdx.x == 0) sh.grp2_frame = -1;                  // (... nor any frame's slot -> token table)
  __syncthreads();
  for (int f = from; f >= 0 && f >= stop_at; f--) {
    if (final) {
      if (!(above & 2) && !frame_dirty(sh, f)) { above = 0; continue; }
    } else if (f < first_unpruned && !(above & 1)) break;
    above = prune_frame<KI>(s, f, lattice_beam, delta, sh, pp);
    __syncthreads();   // thread 0's dirty-bit updates before anybody tests the next frame
    visited++;
  }
  if (threadIdx.x == 0) sh.n_visited += visited;   // diagnostic
}

// largest j < n with beg[j] <= x (beg ascending; empty ranges repeat a value and are skipped)
__device__ __forceinline__ int seg_find(const int *beg, int n, int x) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (beg[mid] <= x) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// Move the survivors listed by the job in LDS (token frames sh.tbeg[0..j_ntf], link segments
// sh.sbeg[0..j_nseg], destinations sh.tbase/sh.sbase sized by the survivor counts of the last
// pruning sweep) into H at destination + stage.  FROM_W: the source is the raw window (token
// ids carry WBIT, remap table w_remap), else a block of H (remap table h_remap).  H token ids
// >= remap_lo are translated through h_remap.  Flat over all frames: a survivor finds its
// frame by binary search in LDS.
template <int KI, bool FROM_W>
__device__ void compact_move(const Slot &s, Shared &sh, int stage_t, int stage_l, int remap_lo, int dense_tok = 0, int dense_link = 0) {
  const int tid = threadIdx.x;
  gp<i32x4> wt = uptr(s.w_tok), ht = uptr(s.h_tok);
  gp<int> w_remap = uptr(s.w_remap), h_remap = uptr(s.h_remap);
  gp<const i32x4> src_tok = FROM_W ? wt : ht;
  gp<int> remap = FROM_W ? w_remap : h_remap;
  const int ntf = sh.j_ntf, nseg = sh.j_nseg;
  Job &J = job_of(sh);
  // With survivor bitmaps (FROM_W only): the window's tokens below dense_tok and links below dense_link belong to frames the pruning
  // pass has swept, and the bits of their last sweep say which are alive; the frontier frame's tokens and epsilon links (never
  // swept) follow, one contiguous range each, and are read as before.
  const bool bits = FROM_W && VBH_SURVIVOR_BITS;
  auto move_tok = [&](int t, const i32x4 Rt) {
    int nid = -1;
    if ((unsigned)Rt.z != INF_BITS) {
      const int j = seg_find(J.tbeg, ntf, t);
      const int c = atomicAdd(&J.tcnt[j], 1);
      if (c < J.tup[j]) {
        nid = J.tbase[j] + c;
        i32x4 r = Rt; r.w = (int)INF_BITS;
        ht[nid + stage_t] = r;
      } else sh.error = ERR_H_TOK;     // more survivors than the sweep counted: never expected
    }
    remap[t] = nid;
  };
  if (ntf > 0) {
    const int t1 = J.tbeg[ntf];
    int t0 = J.tbeg[0];
    if (bits) {
      gp<const unsigned long long> tb_ = uptr(s.w_tok_alive);
      const int tend = dense_tok < t1 ? dense_tok : t1;
      for (int w = (t0 >> 6) + tid; w < ((tend + 63) >> 6); w += NT) {
        for (unsigned long long m = tb_[w]; m; m &= m - 1) {
          const int t = (w << 6) + (__ffsll((long long)m) - 1);
          if (t >= t0 && t < tend) { const i32x4 rec_ = src_tok[t]; move_tok(t, rec_); }
        }
      }
      t0 = t0 > tend ? t0 : tend;
    }
    for (int base = t0; base < t1; base += NT * UM) {
      i32x4 R[UM]; bool ok[UM];
#pragma unroll
      for (int u = 0; u < UM; u++) {
        const int t = base + u * NT + tid;
        ok[u] = t < t1;
        R[u] = src_tok[ok[u] ? t : t0];
      }
#pragma unroll
      for (int u = 0; u < UM; u++) {
        const int t = base + u * NT + tid;
        if (ok[u]) move_tok(t, R[u]);
      }
    }
  }
  __syncthreads();
  if (nseg > 0) {
    gp<const i32x4> src_lk = uptr(FROM_W ? s.wl : s.hl);
    gp<i32x4> hl = uptr(s.hl);
    gp<const unsigned long long> tb_ = uptr(s.w_tok_alive);
    // is window token i alive (swept frames: the bit of the last sweep; frontier frame: its remap entry, written above)?
    auto w_alive = [&](int i) -> bool { return !bits || i >= dense_tok || ((tb_[i >> 6] >> (i & 63)) & 1ull); };
    auto move_link = [&](int l, const i32x4 Lk) {
      const int a = Lk.x, b2 = Lk.y;
      if (a < 0) return;
      int ns, nd;
      if (a & WBIT) ns = w_alive(a ^ WBIT) ? w_remap[a ^ WBIT] : -1;
      else if (a >= remap_lo) ns = h_remap[a];
      else ns = (unsigned)((gp<const int>)(ht + a))[2] != INF_BITS ? a : -1;
      if (b2 & WBIT) nd = w_alive(b2 ^ WBIT) ? w_remap[b2 ^ WBIT] : -1;
      else if (b2 >= remap_lo) nd = h_remap[b2];
      else nd = (unsigned)((gp<const int>)(ht + b2))[2] != INF_BITS ? b2 : -1;
      if (ns < 0 || nd < 0) return;
      const int j = seg_find(J.sbeg, nseg, l);
      const int c = atomicAdd(&J.scnt[j], 1);
      if (c < J.sup[j]) {
        const int nid = J.sbase[j] + c + stage_l;
        i32x4 r = Lk; r.x = ns; r.y = nd;
        hl[nid] = r;
      } else sh.error = ERR_H_LINK;
    };
    const int l1 = J.sbeg[nseg];
    int l0 = J.sbeg[0];
    if (bits) {
      gp<const unsigned long long> lb_ = uptr(s.w_lk_alive);
      const int lend = dense_link < l1 ? dense_link : l1;
      for (int w = (l0 >> 6) + tid; w < ((lend + 63) >> 6); w += NT) {
        for (unsigned long long m = lb_[w]; m; m &= m - 1) {
          const int l = (w << 6) + (__ffsll((long long)m) - 1);
          if (l >= l0 && l < lend) { const i32x4 rec_ = src_lk[l]; move_link(l, rec_); }
        }
      }
      l0 = l0 > lend ? l0 : lend;
    }
    for (int base = l0; base < l1; base += NT * UM) {
      i32x4 L[UM]; bool ok[UM];
#pragma unroll
      for (int u = 0; u < UM; u++) {
        const int l = base + u * NT + tid;
        ok[u] = l < l1;
        L[u] = src_lk[ok[u] ? l : l0];
        ok[u] = ok[u] && L[u].x >= 0;
      }
      // end points: window ids and ids of the block being re-compacted go through the remap
      // tables (-1 = dead); older history tokens keep their id if they are still alive
      int vs[UM], vd[UM];
#pragma unroll
      for (int u = 0; u < UM; u++) {
        const int a = L[u].x, b = L[u].y;
        const bool wa = ok[u] && (a & WBIT) && !w_alive(a ^ WBIT), wb = ok[u] && (b & WBIT) && !w_alive(b ^ WBIT);      // (a window token a swept frame lost)
        gp<const int> pa = (!ok[u] || wa) ? h_remap : (a & WBIT) ? w_remap + (a ^ WBIT) : (a >= remap_lo) ? h_remap + a : (gp<const int>)(ht + a) + 2;
        gp<const int> pb = (!ok[u] || wb) ? h_remap : (b & WBIT) ? w_remap + (b ^ WBIT) : (b >= remap_lo) ? h_remap + b : (gp<const int>)(ht + b) + 2;
        vs[u] = *pa; vd[u] = *pb;
        if (wa) vs[u] = -1;
        if (wb) vd[u] = -1;
      }
#pragma unroll
      for (int u = 0; u < UM; u++) {
        const int l = base + u * NT + tid;
        if (!ok[u]) continue;
        const int a = L[u].x, b = L[u].y;
        const int ns = ((a & WBIT) || a >= remap_lo) ? vs[u] : ((unsigned)vs[u] != INF_BITS ? a : -1);
        const int nd = ((b & WBIT) || b >= remap_lo) ? vd[u] : ((unsigned)vd[u] != INF_BITS ? b : -1);
        if (ns < 0 || nd < 0) continue;
        const int j = seg_find(J.sbeg, nseg, l);
        const int c = atomicAdd(&J.scnt[j], 1);
        if (c < J.sup[j]) {
          const int nid = J.sbase[j] + c + stage_l;
          i32x4 r = L[u]; r.x = ns; r.y = nd;
          hl[nid] = r;
        } else sh.error = ERR_H_LINK;
      }
    }
  }
  __syncthreads();
  // slots the sweep counted but the move did not fill (an end point died without the link being
  // re-evaluated) become dead entries, so that every frame stays one contiguous range
  for (int j = tid; j < ntf; j += NT) {
    const int c = J.tcnt[j] < J.tup[j] ? J.tcnt[j] : J.tup[j];
    for (int i = J.tbase[j] + c; i < J.tbase[j] + J.tup[j]; i++) {
      i32x4 r; r.x = 0; r.y = 0; r.z = (int)INF_BITS; r.w = (int)INF_BITS;
      gptr(s.h_tok)[i + stage_t] = r;
    }
  }
  for (int j = tid; j < nseg; j += NT) {
    const int c = J.scnt[j] < J.sup[j] ? J.scnt[j] : J.sup[j];
    for (int i = J.sbase[j] + c; i < J.sbase[j] + J.sup[j]; i++) {
      i32x4 r; r.x = -1; r.y = 0; r.z = 0; r.w = 0;
      gptr(s.hl)[i + stage_l] = r;
    }
  }
}

// job bookkeeping shared by the two compactions (thread 0)
__device__ __forceinline__ void job_add_tok(Shared &sh, const FrameRec &r, int f) {
  Job &J = job_of(sh);
  const int j = sh.j_ntf++;
  if (j == 0) sh.j_tf0 = f;
  const int n = r.tok_e - r.tok_b, up = r.a_tok < n ? r.a_tok : n;
  J.tbeg[j] = r.tok_b; J.tbeg[j + 1] = r.tok_e; J.tup[j] = up; J.tbase[j] = sh.j_tdst; J.tcnt[j] = 0;
  sh.j_tdst += up;
}
__device__ __forceinline__ void job_add_seg(Shared &sh, int b, int e, int a, int id) {
  Job &J = job_of(sh);
  const int j = sh.j_nseg++;
  const int n = e - b, up = a < n ? a : n;
  J.sbeg[j] = b; J.sbeg[j + 1] = e; J.sup[j] = up; J.sbase[j] = sh.j_ldst; J.scnt[j] = 0; J.sid[j] = id;
  sh.j_ldst += up;
}
// the FrameRecs a job is built from, fetched by everybody into LDS (sh.cost is free between the
// pruning pass and the next frame) instead of one dependent load after the other by thread 0
__device__ __forceinline__ const FrameRec *job_stage_frames(const Slot &s, Shared &sh, int f0, int nf) {
  int *dst = reinterpret_cast<int *>(sh.arena);
  gp<const int> src = (gp<const int>)(s.fr + f0);
  for (int i = threadIdx.x; i < nf * 12; i += NT) dst[i] = src[i];
  __syncthreads();
  return reinterpret_cast<const FrameRec *>(sh.arena);
}
// new positions -> FrameRecs (after compact_move; one thread per frame / segment)
__device__ __forceinline__ void job_publish(const Slot &s, Shared &sh) {
  const int tid = threadIdx.x;
  Job &J = job_of(sh);
  for (int j = tid; j < sh.j_ntf; j += NT) {
    gp<int> r = (gp<int>)(s.fr + (sh.j_tf0 + j));
    r[0] = J.tbase[j]; r[1] = J.tbase[j] + J.tup[j]; r[8] = J.tcnt[j] < J.tup[j] ? J.tcnt[j] : J.tup[j];
  }
  for (int j = tid; j < sh.j_nseg; j += NT) {
    gp<int> r = (gp<int>)(s.fr + (J.sid[j] >> 1));
    const int c = J.scnt[j] < J.sup[j] ? J.scnt[j] : J.sup[j];
    if (J.sid[j] & 1) { r[4] = J.sbase[j]; r[5] = J.sbase[j] + J.sup[j]; r[10] = c; }
    else { r[2] = J.sbase[j]; r[3] = J.sbase[j] + J.sup[j]; r[9] = c; }
  }
}
static_assert(sizeof(FrameRec) == 48, "FrameRec is addressed as 12 ints / 3 vectors");

// Re-compact the block that the previous compaction appended to H (frames (tail_f0, tail_f1]
// and their links).  When it was appended its youngest frames had no look-ahead, so most of
// their tokens and links survived; by now they have been pruned again and nearly all of them are
// dead.  Survivors are staged in the free space behind the block and copied back to its start;
// h_remap[] maps old to new token indices (-1 = dead) for this block.  Everything that refers to
// these tokens is either inside the block or still in the raw window.  Returns false on overflow.
VBH_KI __device__ bool recompact_tail(const Slot &s, Shared &sh) {
  const int tid = threadIdx.x;
  const int tf0 = sh.tail_f0, tf1 = sh.tail_f1;
  if (tf1 < 0) return true;                       // nothing appended yet
  const int fa = tf0 < 0 ? 0 : tf0;               // link frames fa..tf1 (emitting: fa..tf1-1), token frames tf0+1..tf1
  const int tail_tb = sh.tail_tok_b, tail_lb = sh.tail_link_b;
  const int old_nt = sh.h_ntok, old_nl = sh.h_nlink;
  __syncthreads();
  const FrameRec *lf = job_stage_frames(s, sh, fa, tf1 - fa + 1);
  if (tid == 0) {
    sh.j_ntf = 0; sh.j_nseg = 0; sh.j_tf0 = 0; sh.j_tdst = tail_tb; sh.j_ldst = tail_lb;
    for (int f = fa; f <= tf1; f++) {
      const FrameRec r = lf[f - fa];
      if (f > tf0) job_add_tok(sh, r, f);
      if (f > tf0) job_add_seg(sh, r.pl_b, r.pl_e, r.a_pl, (f << 1) | 1);
      if (f < tf1) job_add_seg(sh, r.el_b, r.el_e, r.a_el, f << 1);   // emitting links of the frontier frame are still in W
    }
    // staging area = [old_nt, old_nt + survivors) and [old_nl, old_nl + survivors)
    if (old_nt + (sh.j_tdst - tail_tb) > s.h_tok_cap) sh.error = ERR_H_TOK;
    if (old_nl + (sh.j_ldst - tail_lb) > s.h_link_cap) sh.error = ERR_H_LINK;
    sh.h_ntok = sh.j_tdst; sh.h_nlink = sh.j_ldst;     // new ends of the history
  }
  __syncthreads();
  if (sh.error != ERR_NONE) return false;
  const int new_nt = sh.h_ntok, new_nl = sh.h_nlink;
  const int stage_t = old_nt - tail_tb, stage_l = old_nl - tail_lb;   // staging offset relative to the final position
  compact_move<KI, false>(s, sh, stage_t, stage_l, tail_tb);
  __syncthreads();
  if (sh.error != ERR_NONE) return false;
  // ---- staging -> final position (the two regions never overlap: staging starts at the old end)
  {
    gp<i32x4> ht = uptr(s.h_tok), hl = uptr(s.hl);
    for (int base = tail_tb; base < new_nt; base += NT * U) {
      i32x4 R[U];
#pragma unroll
      for (int u = 0; u < U; u++) { const int i = base + u * NT + tid; R[u] = ht[(i < new_nt ? i : tail_tb) + stage_t]; }
#pragma unroll
      for (int u = 0; u < U; u++) { const int i = base + u * NT + tid; if (i < new_nt) ht[i] = R[u]; }
    }
    for (int base = tail_lb; base < new_nl; base += NT * U) {
      i32x4 R[U];
#pragma unroll
      for (int u = 0; u < U; u++) { const int i = base + u * NT + tid; const int q = (i < new_nl ? i : tail_lb) + stage_l; R[u] = hl[q]; }
#pragma unroll
      for (int u = 0; u < U; u++) { const int i = base + u * NT + tid; if (i < new_nl) hl[i] = R[u]; }
    }
  }
  job_publish(s, sh);
  __threadfence_block();
  __syncthreads();
  return true;
}

// Move the survivors of the raw window into the history: tokens of frames (prev_front, front],
// emitting links from frames [prev_front, front-1], epsilon links within (prev_front, front].
// Returns false (uniformly) on overflow.
VBH_KI __device__ VBH_COLD bool compact_window(const Slot &s, int prev_front, int front, Shared &sh) {
  const int tid = threadIdx.x;
  const int old_tail_tb = sh.tail_tok_b;
  const bool had_tail = sh.tail_f1 >= 0;
  if (!recompact_tail<KI>(s, sh)) return false;
  const int f0 = prev_front < 0 ? 0 : prev_front;
  const int nf = front - f0 + 1;     // frames f0..front  (<= MAX_WIN_FRAMES, checked on the host)
  const FrameRec *lf = job_stage_frames(s, sh, f0, nf);
  if (tid == 0) {
    sh.j_ntf = 0; sh.j_nseg = 0; sh.j_tf0 = 0; sh.j_tdst = sh.h_ntok; sh.j_ldst = sh.h_nlink;
    for (int f = f0; f <= front; f++) {
      const FrameRec r = lf[f - f0];
      if (r.flags & 1) job_add_tok(sh, r, f);
      if (r.flags & 4) job_add_seg(sh, r.pl_b, r.pl_e, r.a_pl, (f << 1) | 1);   // creation order in W: eps links of f, then its emitting links
      if (r.flags & 2) job_add_seg(sh, r.el_b, r.el_e, r.a_el, f << 1);
    }
    if (sh.j_tdst > s.h_tok_cap) sh.error = ERR_H_TOK;
    if (sh.j_ldst > s.h_link_cap) sh.error = ERR_H_LINK;
  }
  __syncthreads();
  if (sh.error != ERR_NONE) return false;
  const int blk_tb = sh.h_ntok, blk_lb = sh.h_nlink;
  // the frontier frame (the job's last) has not been swept: its tokens and epsilon links are read densely, the rest through the bitmaps
  const int dense_tok = (lf[nf - 1].flags & 1) ? lf[nf - 1].tok_b : INT_MAX, dense_link = (lf[nf - 1].flags & 4) ? lf[nf - 1].pl_b : INT_MAX;
  compact_move<KI, true>(s, sh, 0, 0, had_tail ? old_tail_tb : INT_MAX, dense_tok, dense_link);
  __syncthreads();
  if (sh.error != ERR_NONE) return false;
  job_publish(s, sh);
  for (int k = tid; k < nf; k += NT) ((gp<int>)(s.fr + (f0 + k)))[6] = 0;
  if (tid == 0) {
    sh.h_ntok = sh.j_tdst; sh.h_nlink = sh.j_ldst;
    sh.w_ntok = 0; sh.w_nlink = 0;
    sh.tail_tok_b = blk_tb; sh.tail_link_b = blk_lb;
    sh.tail_f0 = prev_front; sh.tail_f1 = front;
  }
  __threadfence_block();
  __syncthreads();
  return true;
}

// ---------------------------------------------------------------------------------------
// State map of the frame being built: state -> (best cost so far, token).  This is the
// reference's per-frame hash (FindOrAddToken, :259-295).  First level: open addressing in LDS,
// tags (state | EPSBIT when the state has epsilon arcs) claimed with atomicCAS, costs lowered with
// atomicMin on the order-preserving key.  A state whose HPROBE probe slots are all taken by other
// states goes to the second level, ent[] in HBM (epoch-stamped keys, never cleared); slots are
// never freed within a frame, so a state always resolves to the same level.
// First-level states get their tokens when the frame is complete (slot_scan): token index =
// frame's first + number of occupied slots before the state's slot, kept per 32-slot group in
// sh.grp, so the map holds 8 bytes per state (16384 states at 1024 threads) and a token's record is
// written once, with its final cost.  Second-level states get a token when they are first reached
// (its index goes to ent[].tokid).
// Probing goes by buckets of four adjacent slots, one 16-byte LDS read each: the probe sequence of a state is its home
// bucket and the HPROBE / 4 - 1 buckets behind it, the slots of a bucket in order.  (A wave's probe loop lasts as long as its
// unluckiest lane's; with one slot per read that was 10-15 dependent LDS round trips per admission in the heavy frames.)
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned map_hash(int st) { return (((unsigned)st * 2654435761u) >> (32 - HBITS)) & ~3u; }
__device__ __forceinline__ int map_find(const Shared &sh, int st) {     // slot, or -1: second level
  if (!sh.use_lds_map) return -1;
  const unsigned *tag = map_tag(sh);
  unsigned b = map_hash(st);
#pragma unroll 1
  for (int p = 0; p < HPROBE / 4; p++, b = (b + 4) & (HCAP - 1)) {
    const u32x4_t t = *reinterpret_cast<const u32x4_t *>(tag + b);
    if ((t.x & STMASK) == (unsigned)st) return (int)b;            // (an empty slot's low bits are no state id)
    if ((t.y & STMASK) == (unsigned)st) return (int)b + 1;
    if ((t.z & STMASK) == (unsigned)st) return (int)b + 2;
    if ((t.w & STMASK) == (unsigned)st) return (int)b + 3;
    if (t.x == HEMPTY || t.y == HEMPTY || t.z == HEMPTY || t.w == HEMPTY) return -1;
  }
  return -1;
}
// lower the cost of the state dsraw & STMASK to tot (dsraw: the arc's next-state field, EPSBIT included).  slot: first-level
// slot or -1; create: this thread is the first to reach the state in this frame; improved: the cost went down.
__device__ __forceinline__ void map_relax(Shared &sh, const EntMap &ent, int dsraw, float tot, unsigned ep, int &slot, bool &create,
                                          bool &improved) {
  const unsigned ck = f2o(tot);
  const int ds = dsraw & STMASK;
  slot = -1; create = false;
  if (sh.use_lds_map) {
    unsigned *tag = map_tag(sh);
    const unsigned want = (unsigned)(dsraw & ~PLDST);
    unsigned b = map_hash(ds);
#pragma unroll 1
    for (int p = 0; p < HPROBE / 4 && slot < 0; p++, b = (b + 4) & (HCAP - 1)) {
      const u32x4_t t = *reinterpret_cast<const u32x4_t *>(tag + b);
      const unsigned tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (slot >= 0) break;
        unsigned cur = tv[j];
        if (cur == HEMPTY) {
          cur = atomicCAS(&tag[b + j], HEMPTY, want);
          if (cur == HEMPTY) { create = true; slot = (int)(b + j); break; }
        }
        if ((cur & STMASK) == (unsigned)ds) { slot = (int)(b + j); break; }
      }
    }
  }
  if (slot >= 0) {
    const unsigned oc = atomicMin(&map_key(sh)[slot], ck);
    improved = ck < oc;
  } else {
    const unsigned long long nk = ((unsigned long long)ep << 32) | ck;
    const unsigned long long old = rmw_min_u64(&ent[ds].key, nk);
    create = (unsigned)(old >> 32) != ep;
    improved = create || nk < old;
  }
}
__device__ __forceinline__ float map_cost(const Shared &sh, const EntMap &ent, int slot, int st) {
  return slot >= 0 ? o2f(map_key(sh)[slot]) : o2f((unsigned)ld_u64(&ent[st].key));
}
// token of a state, frame-relative for first-level states (valid after slot_scan), absolute window index for second-level ones
__device__ __forceinline__ int map_tok(const Shared &sh, const EntMap &ent, int slot, int st, int nb) {
  if (slot < 0) return ld_i32(&ent[st].tokid);
  const uint2 gq = sh.grp[slot >> 5];
  return nb + (int)gq.y + __popc(gq.x & ((1u << (slot & 31)) - 1u));
}
// Epsilon work lists (state ids).  The first FCAP entries of each of the three rotating lists
// live in LDS, so a closure iteration reads and refills its list without a round trip to HBM;
// longer lists continue in the slot's HBM arrays.
__device__ __forceinline__ void front_push(Shared &sh, gp<int> glob, int glob_cap, int k, int q, int st) {
  if (q < FCAP) sh.front_lds[k * FCAP + q] = st;
  else if (q - FCAP < glob_cap) glob[q - FCAP] = st;
  else sh.error = ERR_FRONT;
}
__device__ __forceinline__ int front_get(const Shared &sh, gp<const int> glob, int k, int i) {
  return i < FCAP ? sh.front_lds[k * FCAP + i] : glob[i - FCAP];
}
__device__ __forceinline__ void map_clear(Shared &sh) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 *a = reinterpret_cast<u32x4 *>(sh.arena);
  const u32x4 e = {HEMPTY, HEMPTY, HEMPTY, HEMPTY};        // (an empty tag; as a cost key: larger than any cost)
  for (int i = threadIdx.x; i < 2 * HCAP / 4; i += NT) a[i] = e;
}

// ---------------------------------------------------------------------------------------
// The second half of a frame: ProcessNonemitting (lattice-faster-decoder.cc:796-858) for the tokens of frame fpo, then
// the tokens themselves.  On entry the map holds the states the emitting arcs reached (first level: tag + cost; second level:
// token records [nb, sh.w_ntok) already written by whoever reached them first), work list `cur` (count sh.fcnt[cur]) the
// new states that have input-epsilon arcs, and the emitting links [el_b, el_e) carry a provisional destination (map slot, or
// -1 - state for second-level states).
//   1. closure: iterate over the work lists until no cost changes (arcs through wave_expand);
//   2. slot scan: first-level states are numbered in slot order behind the second-level tokens and their records written
//      with the final cost; second-level tokens get their final cost;
//   3. emitting links get their destination token; the epsilon links are regenerated from the final costs (:824-845) and
//      the tokens they point to tagged (PLDST).
// Returns the first link index of the frame's epsilon links; on return a barrier has made all writes visible.
// VBH_EARLY_EXCISE: a forward link whose cost exceeds the best cost of its destination by more than the lattice beam can never
// survive PruneForwardLinks (:331-347: link_extra_cost = next_tok->extra_cost + (tot - next_tok->tot_cost) with extra_cost >= 0, and
// float addition is monotone), whatever the later frames bring.  It is excised the moment the destination's final cost is known --
// emitting links at their fix-up, epsilon links are not even written -- instead of in the first backward sweep: the sweeps gather no
// destination for it.  The lattice is the same bit for bit.
#ifndef VBH_EARLY_EXCISE
#define VBH_EARLY_EXCISE 1
#endif
#ifdef VBH_DIAG_EPS
#define EPSTAMP(i) { if (threadIdx.x == 0) { const long long now_ = clk(); sh.ept[i] += now_ - sh.ep_last; sh.ep_last = now_; } }
#else
#define EPSTAMP(i)
#endif
VBH_KI __device__ VBH_NONEMIT_ATTR int process_nonemitting(const DevGraph &g, const Slot &s, int nb, float cutoff, unsigned ep,
                                   int cur, Shared &sh, int el_b, int el_e, float lattice_beam, int fpo) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef VBH_DIAG_EPS
  if (tid == 0) sh.ep_last = clk();
#endif
  gp<i32x4> wt = uptr(s.w_tok);
  gp<const i32x4> gst = (gp<const i32x4>)g.st, garcs = (gp<const i32x4>)g.arcs;      // (global, not generic: no FLAT accesses)
  const EntMap ent{uptr(s.ent)};
  const int w_tok_cap = s.w_tok_cap, front_cap = s.front_cap;
  for (int iter = 0;; iter++) {
    const int ncur = sh.fcnt[cur] < FCAP + front_cap ? sh.fcnt[cur] : FCAP + front_cap;
    if (ncur == 0) break;
    if (iter >= MAX_EPS_ITERS) { if (tid == 0) sh.error = ERR_EPS_LOOP; break; }
    const int nxt = cur == 2 ? 0 : cur + 1, spare = nxt == 2 ? 0 : nxt + 1;
    if (tid == 0) sh.fcnt[spare] = 0;      // nobody touches the spare list during this iteration
    gp<const int> fcur = uptr(s.front[cur]);
    gp<int> fnext = uptr(s.front[nxt]);
    int *fcnt_next = &sh.fcnt[nxt];
    const int CH = wave_chunk(ncur);
    for (int base = wave * CH; base < ncur; base += NW * CH) {
      const int i = base + lane;
      const bool have = lane < CH && i < ncur;
      const int st = have ? front_get(sh, fcur, cur, i) : 0;
      const i32x4 sr = gst[st];
      const float c = have ? map_cost(sh, ent, map_find(sh, st), st) : 0.0f;
      const bool exp = have && !(c > cutoff);
      const int a0 = st + 1, cnt = exp ? sr.y : 0;      // (a state's arcs follow its record)
      wave_expand(sh.wx[threadIdx.x >> 6], cnt, [&](bool valid, int owner, int j) {
        const int a = __shfl(a0, owner, 64) + j;
        const float cc = __shfl(c, owner, 64);
        bool create = false, push = false;
        int dsraw = 0, ds = 0, slot = -1;
        if (valid) {
          const i32x4 arc = garcs[a];
          const float tot = cc + asf(arc.y);
          if (tot < cutoff) {
            dsraw = arc.x; ds = dsraw & STMASK;
            bool improved;
            map_relax(sh, ent, dsraw, tot, ep, slot, create, improved);
            push = (create || improved) && dsraw < 0;      // new or improved, and it has epsilon arcs: (re-)expand
          }
        }
        const bool mk = create && slot < 0;                 // second-level state: it gets its token now
        const int nid = wave_alloc(&sh.w_ntok, mk);
        if (mk) {
          if (nid < w_tok_cap) {
            i32x4 r; r.x = dsraw & ~PLDST; r.y = 0; r.z = 0; r.w = (int)INF_BITS;
            wt[nid] = r;
            st_i32(&ent[ds].tokid, nid);
          } else sh.error = ERR_W_TOK;
        }
        const int q = wave_alloc(fcnt_next, push);
        if (push) front_push(sh, fnext, front_cap, nxt, q, ds);
      });
    }
    __syncthreads();
    if (sh.error != ERR_NONE) break;
    cur = nxt;
  }
  __syncthreads();
  EPSTAMP(0)
  // ---- slot scan: thread t numbers the occupied ones of slots [t SPT, (t+1) SPT)
  const int n2 = sh.w_ntok < w_tok_cap ? sh.w_ntok : w_tok_cap;      // second-level tokens: [nb, n2)
  unsigned occ = 0;
  const bool scans = tid < HCAP / SPT && sh.use_lds_map;
  if (scans) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 *tag4 = reinterpret_cast<const u32x4 *>(map_tag(sh) + tid * SPT);
#pragma unroll
    for (int j = 0; j < SPT / 4; j++) {
      const u32x4 t4 = tag4[j];
      occ |= ((unsigned)(t4.x != HEMPTY) | ((unsigned)(t4.y != HEMPTY) << 1) | ((unsigned)(t4.z != HEMPTY) << 2) | ((unsigned)(t4.w != HEMPTY) << 3)) << (4 * j);
    }
  }
  const int mine = __popc(occ);
  const int incl = wave_incl_scan(mine);
  if (lane == 63) sh.scan_part[wave] = incl;
  if (tid == 0) { sh.scan_base = n2; sh.fcnt[0] = 0; sh.fcnt[1] = 0; sh.fcnt[2] = 0; }
  __syncthreads();
  int first = sh.scan_base - nb + incl - mine, total = 0;     // frame-relative index of this thread's first token
#pragma unroll
  for (int i = 0; i < NW; i++) { const int v = sh.scan_part[i]; total += v; if (i < wave) first += v; }
  const int ne_all = sh.scan_base + total;
  if (ne_all > w_tok_cap) { if (tid == 0) sh.error = ERR_W_TOK; }
  else {
    if (scans) {
      // {occupancy of the group's 32 slots, index of the group's first token}: with half a group per thread the lower half's thread writes
      // the index, both their halves of the mask (16-bit stores into the same word)
      if (SPT == 32) {
        uint2 gq; gq.x = occ; gq.y = (unsigned)first; sh.grp[tid] = gq;
        if (VBH_LAZY_DST) (uptr(s.w_grp) + (size_t)(fpo % GRP_RING) * (HCAP / 32))[tid] = ((unsigned long long)gq.y << 32) | gq.x;      // the frame's slot -> token table, for the pruning sweeps
      }
      else {
        reinterpret_cast<unsigned short *>(&sh.grp[tid >> 1].x)[tid & 1] = (unsigned short)occ;
        if (!(tid & 1)) sh.grp[tid >> 1].y = (unsigned)first;
      }
    }
    // token records of the first-level states, final cost included; those with epsilon arcs and a cost within the cutoff go
    // on work list 0 (as slots) for the regeneration of the epsilon links
    const unsigned *tag = map_tag(sh);
    const unsigned *key = map_key(sh);
    int k = nb + first;
    for (unsigned m = occ; __ballot(m != 0) != 0; m &= m - 1) {       // (a wave goes on until its last lane is done)
      bool want = false; int slot = 0;
      if (m) {
        slot = tid * SPT + (__ffs((int)m) - 1);
        const unsigned tg = tag[slot];
        const float c = o2f(key[slot]);
        i32x4 r; r.x = (int)tg; r.y = asi(c); r.z = 0; r.w = (int)INF_BITS;
        wt[k++] = r;
        want = (int)tg < 0 && !(c > cutoff);
      }
      const int q = wave_alloc(&sh.fcnt[0], want);
      if (want) front_push(sh, uptr(s.front[0]), front_cap, 0, q, slot);
    }
    EPSTAMP(1)
    // second-level tokens: final cost; the ones to expand go on work list 1 (as token indices)
    for (int t0 = nb + wave * 64; t0 < n2; t0 += NT) {
      const int t = t0 + lane;
      const bool have = t < n2;
      i32x4 rec; rec.x = 0; rec.y = 0; rec.z = 0; rec.w = 0;
      if (have) rec = wt[t];
      const float c = have ? o2f((unsigned)ld_u64(&ent[rec.x & STMASK].key)) : 0.0f;
      if (have) { gp<int> q = (gp<int>)&wt[t]; q[1] = asi(c); q[2] = 0; }
      const bool want = have && rec.x < 0 && !(c > cutoff);
      const int q = wave_alloc(&sh.fcnt[1], want);
      if (want) front_push(sh, uptr(s.front[1]), front_cap, 1, q, t);
    }
  }
  if (tid == 0) {
    sh.w_ntok = ne_all < w_tok_cap ? ne_all : w_tok_cap;
#if VBH_SURVIVOR_BITS
    sh.w_nlink = (sh.w_nlink + 63) & ~63;      // a frame's links (its epsilon links, then its emitting links) start a 64-bit word of the survivor bitmap
#endif
  }
  __syncthreads();
  EPSTAMP(2)
  const int lb = sh.w_nlink;
  if (sh.error != ERR_NONE) return lb;
  __syncthreads();        // (everybody has read w_nlink before the links below move it)
  gp<i32x4> wl = uptr(s.wl);
#if VBH_LAZY_DST
  // ---- the emitting links' destinations: a link to a first-level state keeps the slot pass 2 gave it (the pruning sweeps translate it,
  //      eval_links); those to second-level states -- listed by pass 2 in sh.hist, 256 of them; more: every link is looked at -- get
  //      their tokens from the states' entries now
  {
    static_assert(SPT == 32, "the slot -> token table is written by the one-group-per-thread scan");
    const int n_l2 = sh.n_l2;
    if (n_l2 <= 256) {
      for (int i = tid; i < n_l2; i += NT) {
        gp<int> py = (gp<int>)&wl[sh.hist[i]] + 1;
        const int d_ = ld_i32(&ent[-1 - *py].tokid);
        (void)VBH_CHK(sh, 30, d_ - nb, ne_all - nb);
        *py = d_ | WBIT;
      }
    } else {
      for (int l = el_b + tid; l < el_e; l += NT) {
        gp<int> py = (gp<int>)&wl[l] + 1;
        const int y = *py;
        if (y < 0) { const int d_ = ld_i32(&ent[-1 - y].tokid); (void)VBH_CHK(sh, 30, d_ - nb, ne_all - nb); *py = d_ | WBIT; }
      }
    }
  }
#else
  // ---- the emitting links' destinations (VBH_FX links per thread in flight: the pass is a chain of one load per round otherwise)
  for (int l0 = el_b + tid; l0 < el_e; l0 += NT * VBH_FX) {
    i32x2 yz[VBH_FX];
#pragma unroll
    for (int u = 0; u < VBH_FX; u++) { const int l = l0 + u * NT; yz[u] = *(gp<i32x2>)((gp<int>)&wl[l < el_e ? l : el_b] + 1); }
#pragma unroll
    for (int u = 0; u < VBH_FX; u++) {
      const int l = l0 + u * NT;
      if (l >= el_e) continue;
      gp<int> py = (gp<int>)&wl[l] + 1;
      const int y = yz[u].x;
#if VBH_EARLY_EXCISE
      const float dcost = y >= 0 ? o2f(map_key(sh)[y]) : o2f((unsigned)ld_u64(&ent[-1 - y].key));
      if (asf(yz[u].y) - dcost > lattice_beam) {
#ifdef VBH_DIAG_EPS
        atomicAdd((unsigned long long *)&sh.ept[6], 1ull);
#endif
        py[-1] = -1; continue;
      }
#endif
      const int d = y >= 0 ? map_tok(sh, ent, y, 0, nb) : ld_i32(&ent[-1 - y].tokid);
      (void)VBH_CHK(sh, 30, d - nb, ne_all - nb);
      *py = d | WBIT;
    }
  }
#endif
  EPSTAMP(3)
  // ---- epsilon links regenerated from the final costs; what they point to is tagged
  {
    const int w_link_cap = s.w_link_cap;
    const unsigned *tag = map_tag(sh);
    const unsigned *key = map_key(sh);
    for (int pass = 0; pass < 2; pass++) {
      const int n_list = sh.fcnt[pass] < FCAP + front_cap ? sh.fcnt[pass] : FCAP + front_cap;
      gp<const int> fl = uptr(s.front[pass]);
      const int CH = wave_chunk(n_list);
      for (int base = wave * CH; base < n_list; base += NW * CH) {
        const int i = base + lane;
        const bool have = lane < CH && i < n_list;
        const int e = have ? front_get(sh, fl, pass, i) : 0;
        int st = 0, src = 0; float c = 0.0f;
        if (have) {
          if (pass == 0) { st = (int)(tag[e] & STMASK); c = o2f(key[e]); src = map_tok(sh, ent, e, 0, nb); }
          else { const i32x4 rec = wt[e]; st = rec.x & STMASK; c = asf(rec.y); src = e; }
        }
        int a0 = 0, cnt = 0;
        if (have) { const i32x4 sr = gst[st]; a0 = st + 1; cnt = sr.y; }
        wave_expand(sh.wx[threadIdx.x >> 6], cnt, [&](bool valid, int owner, int j) {
          const int a = __shfl(a0, owner, 64) + j;
          const float cc = __shfl(c, owner, 64);
          const int so = __shfl(src, owner, 64);
          bool mk = false; int dtok = 0; i32x4 arc; float tot = 0.0f;
          if (valid) {
            arc = garcs[a];
            tot = cc + asf(arc.y);
            if (tot < cutoff) {
              mk = true;
              const int ds = arc.x & STMASK;
              const int dslot = map_find(sh, ds);
              dtok = map_tok(sh, ent, dslot, ds, nb);
              mk = VBH_CHK(sh, 31, dtok - nb, ne_all - nb) && VBH_CHK(sh, 32, so - nb, ne_all - nb);
#if VBH_EARLY_EXCISE
              if (tot - map_cost(sh, ent, dslot, ds) > lattice_beam) mk = false;
#endif
            }
          }
          const int l = wave_alloc(&sh.w_nlink, mk);
          if (mk) {
            if (l < w_link_cap) {
              i32x4 r; r.x = so | WBIT; r.y = dtok | WBIT; r.z = asi(tot); r.w = a;
              if (VBH_NT_EPSLINKS) st_link(wl + l, r); else wl[l] = r;
              (void)__hip_atomic_fetch_or((gp<int>)&wt[dtok], PLDST, __ATOMIC_RELAXED, VBH_RMW_SCOPE);
            } else sh.error = ERR_W_LINK;
          }
        });
      }
    }
  }
  EPSTAMP(4)
  __syncthreads();
  EPSTAMP(5)
  return lb;
}

// Search workspaces ("slots") of one device are a pool shared by every decoder created on it with
// vbh_decoder_create_shared: the persistent grids of several batches are resident at the same time (the grid of batch k
// drains while those of batches k+1, k+2 fill the freed CUs), so a workgroup cannot own slot blockIdx.x; it claims a free
// slot when it starts and gives it back when it leaves.  At most WG_PER_CU workgroups per CU are resident; a pool made for full
// batches holds that many slots, so a resident workgroup finds one at once.  A pool made for small batches (--decode-batch below
// the CU count) that several decoders share holds fewer: a workgroup then waits for a slot -- the holders are resident and
// finish on their own, so the wait ends; it is bounded all the same (a few seconds), after which the workgroup leaves and its
// grid's utterances fall to the grid's other workgroups, or are searched again by the host (ERR_NOT_RUN).  The pool is partitioned by XCD (HW_REG_XCC_ID) and a
// workgroup looks in its own XCD's partition first: the slot's working set then stays in that XCD's L2.  That is a speed
// matter only -- hand-over is an agent-scope release by the leaving workgroup and an agent-scope acquire by the claiming
// one, and the only data a new owner reads before writing it are the epoch-stamped state-map entries (older versions carry
// older epochs and lose every comparison) and the slot's epoch counter (an agent-scope atomic).
struct SlotPoolDev { int *flags; int n_per_xcd, n_xcd; };

__device__ __forceinline__ int claim_slot(const SlotPoolDev &sp) {
  const int n = sp.n_per_xcd, nx = sp.n_xcd;
  int x = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xF);     // HW_REG_XCC_ID[3:0]
  if (x >= nx) x %= nx;
  const int start = (int)((blockIdx.x / 8u) % (unsigned)n);
  for (int round = 0; round < (1 << 20); round++) {
    // own partition first, then the others
    for (int dx = 0; dx < (round == 0 ? 1 : nx); dx++) {
      const int part = (x + dx) % nx;
      for (int k = 0; k < n; k++) {
        int i = start + k; if (i >= n) i -= n;
        int *f = sp.flags + part * n + i;
        if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) continue;
        int expected = 0;
        if (__hip_atomic_compare_exchange_strong(f, &expected, 1, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return part * n + i;
      }
    }
    if (round < 64) __builtin_amdgcn_s_sleep(16); else __builtin_amdgcn_s_sleep(127);
  }
  return -1;      // (seconds without a free slot: see above)
}

// ADM: ProcessEmitting's admission rule, a template argument so that the rule not in use costs the frame loop no registers.
//   VBH_ADMIT_EXACT (0): tot <= the frame's final next_cutoff, whatever the order (rounds 1-3; oracle exact_cutoff = 1);
//   VBH_ADMIT_ORDERED (1): the reference's online tightening (:747-760) as a prefix minimum over a canonical order: tot <= min(initial
//   next_cutoff, tot' + adaptive_beam of every arc leaving a token of a lower bucket) (oracle exact_cutoff = 3, adm_bucket_bits = 6).
template <int ADM>
__global__ __launch_bounds__(NT, VBH_WAVES_PER_EU) void decode_kernel(DevGraph g, DecCfg cfg, const Slot *slots, SlotPoolDev spool,
                                                     const float *__restrict__ loglikes, int P,
                                                     const UttDesc *__restrict__ utts, const int *__restrict__ work_list,
                                                     int n_work, int *work_counter, UttResult *results, const Pool *pool_p) {
  __shared__ Shared sh;
  // The slot's pointers live in LDS: by value they overflow the SGPR file, and read through
  // the global pointer the compiler re-fetches them with vector loads inside every loop.
  __shared__ Slot s_slot;
  __shared__ Pool s_pool;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int my_slot = blockIdx.x;
  if (spool.flags) {
    if (tid == 0) sh.utt = claim_slot(spool);
    __syncthreads();
    my_slot = sh.utt;
    __syncthreads();
    if (my_slot < 0) return;       // (uniform)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  {
    const int *src = reinterpret_cast<const int *>(&slots[my_slot]);
    int *dst = reinterpret_cast<int *>(&s_slot);
    for (int i = tid; i < (int)(sizeof(Slot) / sizeof(int)); i += NT) dst[i] = src[i];
    const int *psrc = reinterpret_cast<const int *>(pool_p);
    int *pdst = reinterpret_cast<int *>(&s_pool);
    for (int i = tid; i < (int)(sizeof(Pool) / sizeof(int)); i += NT) pdst[i] = psrc[i];
  }
#ifdef VBH_LDS_CANARY
  if (tid < 4) { sh.canary_lo[tid] = 0xC0FFEE00u + tid; sh.canary_hi[tid] = 0xFACADE00u + tid; }
  const unsigned long long slot_sum0 = (unsigned long long)slots[my_slot].ent ^ (unsigned long long)slots[my_slot].w_tok ^ (unsigned long long)slots[my_slot].hl;
#endif
  __syncthreads();
  const Slot &s = s_slot;
  const Pool &pool = s_pool;
  gp<const i32x4> gst = (gp<const i32x4>)g.st, garcs = (gp<const i32x4>)g.arcs;      // (global, not generic: no FLAT accesses)
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
    // epochs start at 1: the high word 0xFFFFFFFF is the cleared (memset 0xFF) state.  The slot's epoch counter is advanced
    // with an agent-scope atomic: the next owner of the slot (maybe on another XCD) continues after this utterance's epochs
    if (tid == 0) sh.flag = __hip_atomic_fetch_add(gptr(s.state), T + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int epoch_base = sh.flag + 1;
    __syncthreads();
#if VBH_X1
    int c_tok_exp = 0, c_arc_exp = 0;              // per-thread counters, reduced at the end (32 bits: an utterance's totals are tens of millions)
#else
    long long c_tok_exp = 0, c_arc_exp = 0;        // per-thread counters, reduced at the end
#endif
    long long t_last = clk();
    const long long rt_start = (long long)__builtin_amdgcn_s_memrealtime();
#ifdef VBH_DIAG
    int f_heavy = 0;
#ifdef VBH_DIAG_P1
#define PHASE(i) { if (tid == 0) { const long long now_ = clk(); sh.tph[i] += now_ - t_last; t_last = now_; } }
#else
#define PHASE(i) { if (tid == 0) { const long long now_ = clk(); sh.tph[i] += now_ - t_last; if (f_heavy) sh.hph[i] += now_ - t_last; t_last = now_; } }
#endif
#else
#if VBH_X2
#define PHASE(i) { const long long now_ = clk(); if (tid == 0) sh.tph[i] += now_ - t_last; t_last = now_; }      // (t_last stays wave-uniform: scalar registers)
#else
#define PHASE(i) { if (tid == 0) { const long long now_ = clk(); sh.tph[i] += now_ - t_last; t_last = now_; } }
#endif
#endif
#ifdef VBH_DIAG
#define PHASE_D(i) PHASE(i)
#define DIAG_ADD(i, pred) { const int c_ = __popcll(__ballot(pred)); if (lane == 0 && c_) atomicAdd(&sh.diag[i], c_); }
    if (tid < 8) { sh.diag[tid] = 0; sh.fcl[tid] = 0; }
    if (tid < 12) sh.hph[tid] = 0;
#else
#define PHASE_D(i)
#define DIAG_ADD(i, pred)
#endif
    if (tid == 0) {
      for (int i = 0; i < 12; i++) sh.tph[i] = 0;
      sh.w_ntok = 0; sh.w_nlink = 0; sh.h_ntok = 0; sh.h_nlink = 0; sh.error = ERR_NONE;
      sh.n_max_bound = 0; sh.n_min_bound = 0; sh.n_cand = 0; sh.cand_bound = 0xFFFFFFFFu; sh.n_visited = 0; sh.n_l2 = 0; sh.grp2_frame = -1;
      if (ADM == VBH_ADMIT_ORDERED) sh.init_bound = 0xFFFFFFFFu;
      sh.fcnt[0] = 0; sh.fcnt[1] = 0; sh.fcnt[2] = 0;
      sh.changed2[0] = 0; sh.changed2[1] = 0;
      sh.cnt_tok[0] = 0; sh.cnt_tok[1] = 0; sh.cnt_el[0] = 0; sh.cnt_el[1] = 0; sh.cnt_pl[0] = 0; sh.cnt_pl[1] = 0;
      sh.tail_tok_b = 0; sh.tail_link_b = 0; sh.tail_f0 = -1; sh.tail_f1 = -2;
      sh.chk[0] = 0; sh.chk[1] = 0; sh.chk[2] = 0; sh.chk[3] = 0;
    }
#ifdef VBH_DIAG_EPS
    if (tid < 12) sh.ept[tid] = 0;
#endif
    for (int i = tid; i < DIRTY_FRAMES / 32; i += NT) sh.dirty[i] = ~0u;   // no frame has been evaluated yet
    if (ADM == VBH_ADMIT_ORDERED && tid < 64) sh.bmin[tid] = 0xFFFFFFFFu;
    map_clear(sh);
    if (tid == 0) sh.use_lds_map = cfg.use_lds_map;
    pp = 0;
    __syncthreads();

    // ---- InitDecoding (:56-73): start token, then the epsilon closure with cutoff = beam
    int cur_b, cur_e;        // token range of the current frame (window or history indices)
    bool cur_w = true;
    {
      const unsigned ep = (unsigned)(0xFFFFFFFFu - (unsigned)(epoch_base + 0));
      if (tid == 0) {
        const i32x4 sr = gst[g.start];
        int slot; bool create, improved;
        const int sraw = g.start | (sr.y > 0 ? EPSBIT : 0);
        const EntMap ent0{gptr(s.ent)};
        map_relax(sh, ent0, sraw, 0.0f, ep, slot, create, improved);
        if (slot < 0) {        // second-level state: its token is made here (first-level ones by the slot scan)
          i32x4 r; r.x = sraw; r.y = 0; r.z = 0; r.w = (int)INF_BITS;
          gptr(s.w_tok)[0] = r;
          st_i32(&ent0[g.start].tokid, 0);
          sh.w_ntok = 1;
        }
        if (sr.y > 0) { front_push(sh, gptr(s.front[0]), s.front_cap, 0, 0, g.start); sh.fcnt[0] = 1; }
      }
      __threadfence_block();
      __syncthreads();
      const int plb = process_nonemitting<ADM>(g, s, 0, cfg.beam, ep, 0, sh, 0, 0, cfg.lattice_beam, 0);
      cur_b = 0; cur_e = sh.w_ntok < s.w_tok_cap ? sh.w_ntok : s.w_tok_cap;
      if (tid == 0) {
        FrameRec r;
        r.tok_b = 0; r.tok_e = cur_e; r.el_b = 0; r.el_e = 0;
        r.pl_b = plb; r.pl_e = sh.w_nlink < s.w_link_cap ? sh.w_nlink : s.w_link_cap;
        r.flags = 1 | 4; r.cost_offset = 0.0f; r.a_tok = cur_e; r.a_el = 0; r.a_pl = r.pl_e - r.pl_b; r.pad = 0;
        st_frame(s.fr, r);
        sh.fcnt[0] = 0; sh.fcnt[1] = 0; sh.fcnt[2] = 0;
      }
      __syncthreads();
    }
    long long c_tok_new = cur_e, c_link_new = sh.w_nlink;

    int prev_front = -1;      // frontier frame at the previous pruning (-1: none yet)
    int frames_done = 0;
    PHASE(6)
    for (int frame = 0; frame < T && sh.error == ERR_NONE; frame++) {
#ifdef VBH_DIAG
      const long long f_t0 = clk(); const int f_n0 = cur_e - cur_b;
#ifndef VBH_DIAG_CLASS
#define VBH_DIAG_CLASS 3
#endif
      f_heavy = (f_n0 < 512 ? 0 : f_n0 < 2048 ? 1 : f_n0 < 6144 ? 2 : 3) == VBH_DIAG_CLASS;
#endif
#ifdef VBH_LDS_CANARY
      if (tid == 0) {
        bool bad = false;
        for (int i = 0; i < 4; i++) bad |= sh.canary_lo[i] != 0xC0FFEE00u + i || sh.canary_hi[i] != 0xFACADE00u + i;
        const unsigned long long sum = (unsigned long long)s_slot.ent ^ (unsigned long long)s_slot.w_tok ^ (unsigned long long)s_slot.hl;
        if (bad || sum != slot_sum0) sh.error = ERR_LDS;
      }
      __syncthreads();
      if (sh.error != ERR_NONE) break;
#endif
      // ---- periodic pruning (:85-87); frame 0 has nothing behind it
      if (frame > 0 && frame % cfg.prune_interval == 0) {
        __syncthreads();
        // intermediate passes walk back at most one interval beyond the unpruned frames: a bound on
        // the work (long-lived parallel paths would otherwise drag every pass back to frame 0);
        // whatever is left stale is a lower bound and is settled by the final pass
        prune_backward<ADM>(s, frame - 1, prev_front < 0 ? 0 : prev_front, (prev_front < 0 ? 0 : prev_front) - cfg.prune_back,
                       cfg.lattice_beam, cfg.lattice_beam * 0.1f, /*final=*/false, sh, pp);
        PHASE(4)
        if (!compact_window<ADM>(s, prev_front, frame, sh)) break;
        prev_front = frame;
        { const FrameRec fr = ld_frame(s.fr + frame); cur_b = fr.tok_b; cur_e = fr.tok_e; }
        cur_w = false;
        PHASE(5)
      }
#if VBH_SURVIVOR_BITS
      // the tokens of frame + 1 start a word of the survivor bitmap (nobody reads the counter between the barrier that ended the
      // previous frame and the one inside the cutoff phase below)
      if (tid == 0) sh.w_ntok = (sh.w_ntok + 63) & ~63;
#endif
      const int tb = cur_b, n = cur_e - cur_b;
      if (n >= (1 << 26)) { if (tid == 0) sh.error = ERR_W_TOK; __syncthreads(); break; }      // (uniform; see w_tok_cap: the candidates' source-token word)
      gp<const i32x4> tok = uptr(cur_w ? s.w_tok : s.h_tok) + tb;
      gp<i32x4> cand = uptr(s.cand);
      const int cand_cap = s.cand_cap;
      const int enc = cur_w ? WBIT : 0;
      const float *llrow = loglikes + (ud.row0 + frame) * (long long)P;
      const unsigned ep = (unsigned)(0xFFFFFFFFu - (unsigned)(epoch_base + frame + 1));
      const unsigned long long EPK = (unsigned long long)ep << 32;

      // ---- GetCutoff (:618-685); the costs go to LDS on the way.  The state map is emptied for the
      //      frame about to be built (its last readers finished before the barrier that ended the
      //      previous frame; its next writers come after this phase's reduction barriers)
      const bool ll_lds = VBH_LL_LDS && P <= LLCAP;
      float llr[(LLCAP + NT - 1) / NT];
      if (ll_lds) {
#pragma unroll
        for (int u = 0; u < (LLCAP + NT - 1) / NT; u++) { const int i = u * NT + tid; llr[u] = i < P ? llrow[i] : 0.0f; }
      }
      unsigned long long bk = ~0ull;
      unsigned wk = 0u;                                 // the largest cost key (VBH_SEL_SKIP)
      for (int i0 = 0; i0 < n; i0 += NT * 4) {          // (four records per thread in flight; only {state, cost} of each is read)
        i32x2 r[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int i = i0 + u * NT + tid; r[u] = *reinterpret_cast<gp<const i32x2>>(tok + (i < n ? i : 0)); }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int i = i0 + u * NT + tid;
          if (i < n) {
            if (i < COST_CAP) lds_cost(sh)[i] = asf(r[u].y);
            const unsigned long long k = ((unsigned long long)f2o(asf(r[u].y)) << 32) | (unsigned)(r[u].x & STMASK);
            bk = k < bk ? k : bk;
            if (VBH_SEL_SKIP) wk = (unsigned)(k >> 32) > wk ? (unsigned)(k >> 32) : wk;
          }
        }
      }
      if (ll_lds) {      // (its readers come after the reduction barrier below; its last readers were pass 1 of the previous frame)
#pragma unroll
        for (int u = 0; u < (LLCAP + NT - 1) / NT; u++) { const int i = u * NT + tid; if (i < P) sh.ll[i] = llr[u]; }
      }
      if (VBH_SEL_SKIP) {
        // (the maximum rides on the minimum's barrier: its partials in sh.hist, idle between the previous frame's second-level link list
        // and the select, which empties it behind a barrier of its own)
        const unsigned wmax = ~wave_min_u32(~wk);
        if (lane == 0) sh.hist[par * NW + wave] = wmax;
      }
      const int par_cut = par;
      bk = block_min_u64(bk, sh, par);
      if (VBH_SEL_SKIP) {
        wk = sh.hist[par_cut * NW];
#pragma unroll
        for (int i = 1; i < NW; i++) wk = sh.hist[par_cut * NW + i] > wk ? sh.hist[par_cut * NW + i] : wk;
      }
      PHASE_D(9)
      const float best = n > 0 ? o2f((unsigned)(bk >> 32)) : INF;
      const int best_state = (int)(unsigned)bk;
      float cur_cutoff, adaptive_beam;
      if (cfg.max_active == INT_MAX && cfg.min_active == 0) {
        adaptive_beam = cfg.beam; cur_cutoff = best + cfg.beam;
      } else {
        const float beam_cutoff = best + cfg.beam;
        float mac = INF, mic = INF;
        unsigned long long cnt = 0;   // (#cost < beam_cutoff) << 32 | (#cost <= beam_cutoff)
        {     // (four costs per LDS read, as in select_kth)
          typedef float f32x4 __attribute__((ext_vector_type(4)));
          const int nl = n < COST_CAP ? n : COST_CAP, n4 = nl & ~3;
          const f32x4 *c4 = reinterpret_cast<const f32x4 *>(lds_cost(sh));
          for (int i = tid * 4; i < n4; i += NT * 4) {
            const f32x4 c = c4[i >> 2];
            cnt += ((unsigned long long)((int)(c.x < beam_cutoff) + (int)(c.y < beam_cutoff) + (int)(c.z < beam_cutoff) + (int)(c.w < beam_cutoff)) << 32) |
                   (unsigned long long)((int)(c.x <= beam_cutoff) + (int)(c.y <= beam_cutoff) + (int)(c.z <= beam_cutoff) + (int)(c.w <= beam_cutoff));
          }
          for (int i = n4 + tid; i < n; i += NT) {
            const float c = frame_cost(sh, tok, i);
            cnt += ((unsigned long long)(c < beam_cutoff) << 32) | (unsigned long long)(c <= beam_cutoff);
          }
        }
        cnt = block_sum_u64(cnt, sh, par);
        const int cnt_lt = (int)(cnt >> 32), cnt_le = (int)(unsigned)cnt;
        // the order statistic is only needed when it decides the comparison
        if (n > cfg.max_active && cnt_lt > cfg.max_active) mac = select_kth<ADM>(tok, n, cfg.max_active, sh, (unsigned)(bk >> 32), wk);
        if (mac < beam_cutoff) {
          adaptive_beam = mac - best + cfg.beam_delta; cur_cutoff = mac;
          if (tid == 0) sh.n_max_bound++;
        } else {
          if (n > cfg.min_active) {
            if (cfg.min_active == 0) mic = best;
            else if (cnt_le <= cfg.min_active) mic = select_kth<ADM>(tok, n, cfg.min_active, sh, (unsigned)(bk >> 32), wk);
            else mic = beam_cutoff;   // some value <= beam_cutoff: the comparison below is false
          }
          if (mic > beam_cutoff) {
            adaptive_beam = mic - best + cfg.beam_delta; cur_cutoff = mic;
            if (tid == 0) sh.n_min_bound++;
          } else { adaptive_beam = cfg.beam; cur_cutoff = beam_cutoff; }
        }
      }
      // ---- cost_offset (:706-711).  The initial next_cutoff from the best token's arcs (:712-724) is taken in pass 1 by the lanes
      //      that expand that token: min over arcs of (new_weight + adaptive_beam), in the reference's own association order
      const float cost_offset = n > 0 ? -best : 0.0f;
      if (tid == 0) ((gp<float>)(s.fr + frame))[7] = cost_offset;
      PHASE(0)
      // ---- the state map is emptied for the frame about to be built (the costs the cutoff phase kept in its memory were last
      //      read before that phase's last reduction barrier; the map's first writers come after pass 1's reduction barrier)
      map_clear(sh);
      // The ends of the token and link windows, i.e. where frame + 1 will start.  Read HERE, not behind pass 1's reduction barrier where
      // they are first used: pass 2 moves both counters (second-level tokens, every admitted link), and a wave that left that barrier
      // late -- its SIMD busy with another workgroup's waves -- read them after a faster wave's first allocation: two waves of one
      // frame then disagreed on the frame's first token / link.  With one workgroup per CU the window is a few instructions against a
      // global round trip and was never seen hit; with two workgroups per CU (NT = 512) five utterances in a thousand came out different
      // from run to run, some with cyclic lattices (r03).  Nothing moves the counters between here and pass 2.
      const int nb = __builtin_amdgcn_readfirstlane(sh.w_ntok);     // first token of frame+1
      const int lb = __builtin_amdgcn_readfirstlane(sh.w_nlink);
      // ---- pass 1 (arc-parallel): every candidate arc -> scratch list; min of tot + adaptive_beam
#ifdef VBH_DIAG_P1
#define P1STAMP(i) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (tid == 0) { const long long now_ = clk(); sh.hph[i] += now_ - p1_t; p1_t = now_; } }
      long long p1_t = clk();
#else
#define P1STAMP(i)
#endif
      float mymin = INF;
      {
        // Rounds of <= 64 tokens per wave, software-pipelined: while the arcs of round r are expanded, the state records of round
        // r + 1 and the token records of round r + 2 are already on their way (a round is otherwise a chain of four dependent
        // loads: token, state, arc, score).
        const int CH = wave_chunk(n), RS = NW * CH;
        auto ld_tok = [&](int b) -> i32x2 {      // {state | flags, cost} of this lane's token of the round starting at b
          const int i_ = b + lane;
          i32x2 r_; r_.x = 0; r_.y = 0x7FC00000;      // (no token: a NaN cost fails every comparison)
          if (b < n) { const i32x4 t_ = VBH_NT_P1TOK ? __builtin_nontemporal_load(tok + ((lane < CH && i_ < n) ? i_ : 0)) : tok[(lane < CH && i_ < n) ? i_ : 0]; if (lane < CH && i_ < n) { r_.x = t_.x; r_.y = t_.y; } }
          return r_;
        };
        auto ld_st = [&](const i32x2 &r_) -> i32x4 { return ld_graph(gst + ((asf(r_.y) <= cur_cutoff) ? (r_.x & STMASK) : 0)); };
        const int b0 = wave * CH;
        i32x2 recC = ld_tok(b0), recB = ld_tok(b0 + RS);
        i32x4 srC = ld_st(recC);
        for (int base = b0; base < n; base += RS) {
          P1STAMP(0)
          const i32x4 srB = ld_st(recB);
          const i32x2 recA = ld_tok(base + 2 * RS);
          const i32x2 rec = recC;
          const i32x4 sr = srC;
          recC = recB; srC = srB; recB = recA;
          const float c = asf(rec.y);
          const bool exp = c <= cur_cutoff;       // (false for lanes without a token)
          // (the sign bit of the first-arc index marks the frame's best token)
          const int a0 = ((rec.x & STMASK) + 1 + sr.y) | ((exp && (rec.x & STMASK) == best_state) ? EPSBIT : 0), cnt = exp ? sr.z - sr.y : 0;
          // ordered admission: the token's bucket and, as a filter, its bucket's bound as it stands at the start of the round, in one
          // word for the lanes that expand its arcs: the key with its six low bits giving way to the bucket (read back with those
          // bits set: a bound rounded UP, and an unset bound reads as NaN, which no cost exceeds)
          int bkw = 0;
          if (ADM == VBH_ADMIT_ORDERED) {
            const int bk = adm_bucket(sr.x & STMASK);      // (of the graph's own state id: oracle/vb_oracle_decode.c VBO_ADM_BUCKET)
            const unsigned pre = adm_bounds(sh.bmin, &sh.init_bound);
            bkw = (int)(((unsigned)__shfl((int)pre, bk, 64) & ~63u) | (unsigned)bk);
          }
          c_tok_exp += exp;
          wave_expand_n<UA>(sh.wx[wave], cnt, [&](const bool (&valid)[UA], const int (&owner)[UA], const int (&j)[UA]) {
            P1STAMP(1)
            int a[UA]; float cc[UA]; bool isb[UA]; i32x4 arc[UA]; float sc[UA]; int bko[UA];
#pragma unroll
            for (int u = 0; u < UA; u++) {
              const int ao = __shfl(a0, owner[u], 64);
              a[u] = (ao & ~EPSBIT) + j[u]; isb[u] = ao < 0;
              cc[u] = __shfl(c, owner[u], 64);
              bko[u] = (ADM == VBH_ADMIT_ORDERED && !VBH_ADM_LATE) ? __shfl(bkw, owner[u], 64) : 0;
            }
#pragma unroll
            for (int u = 0; u < UA; u++) arc[u] = ld_graph(garcs + (valid[u] ? a[u] : 0));
            P1STAMP(2)
#pragma unroll
            for (int u = 0; u < UA; u++) sc[u] = cfg.acwt * llrow[valid[u] ? arc[u].z : 0];
            P1STAMP(3)
            float ac[UA], tot[UA];
#pragma unroll
            for (int u = 0; u < UA; u++) {
              ac[u] = cost_offset - sc[u];
              tot[u] = (cc[u] + ac[u]) + asf(arc[u].y);
              if (valid[u]) {
                mymin = fminf(mymin, tot[u] + adaptive_beam);
                if (isb[u]) mymin = fminf(mymin, (((asf(arc[u].y) + cost_offset) - sc[u]) + best) + adaptive_beam);
                c_arc_exp++;
              }
            }
            // An arc whose tot already exceeds a bound on next_cutoff -- the smallest tot + adaptive_beam anybody in the workgroup has
            // seen so far, kept in LDS -- cannot be admitted (the final next_cutoff is the minimum over all arcs, so it is no larger):
            // it is not written to the candidate list.  (What the reference does one arc at a time, :758-760, as a filter only:
            // the admitted set is still exactly {tot <= final next_cutoff}.)
            bool keep[UA];
            if (ADM == VBH_ADMIT_ORDERED) {
              // Ordered admission: every arc lowers its bucket's minimum (and the best token's arcs the initial bound); what an arc is
              // admitted against is only known when all arcs have (pass 2), but the bounds as they stand now are no smaller, so an arc
              // above its bucket's present bound is not written
#if VBH_ADM_REFRESH
              const unsigned pre2 = adm_bounds(sh.bmin, &sh.init_bound);      // the bounds as they stand now, not at the start of the round
#endif
#pragma unroll
              for (int u = 0; u < UA; u++) {
                if (VBH_ADM_LATE) bko[u] = __shfl(bkw, owner[u], 64);
#if VBH_ADM_REFRESH
                const float bnd = fminf(o2f((unsigned)bko[u] | 63u), o2f((unsigned)__shfl((int)pre2, bko[u] & 63, 64)));
#else
                const float bnd = o2f((unsigned)bko[u] | 63u);
#endif
                keep[u] = valid[u] && !(tot[u] > bnd);
                bko[u] &= 63;
                if (valid[u]) {
                  // Only an arc that gets below the bound of its OWN bucket can lower the bound of a later one (those are minima over
                  // this bucket's bound and more), so every other arc -- all but the frame's few best -- spares its atomic: bmin[] need
                  // not hold the buckets' exact minima, only the prefix minima over it must come out exact.  (An unset bound is a NaN:
                  // the comparison is written so that it then fails and the atomic is done.)
                  const float v = tot[u] + adaptive_beam;
                  if (!(v >= bnd)) atomicMin(&sh.bmin[bko[u]], f2o(v));
                  if (isb[u]) atomicMin(&sh.init_bound, f2o((((asf(arc[u].y) + cost_offset) - sc[u]) + best) + adaptive_beam));
                }
              }
            } else {
              float it_min = INF;
#pragma unroll
              for (int u = 0; u < UA; u++) if (valid[u]) it_min = fminf(it_min, tot[u] + adaptive_beam);
              const unsigned it_key = wave_min_u32(f2o(it_min));      // (the order-preserving key: min of keys = key of the min)
              if (lane == 0) atomicMin(&sh.cand_bound, it_key);
              const float bnd = o2f(sh.cand_bound);
#pragma unroll
              for (int u = 0; u < UA; u++) keep[u] = valid[u] && !(tot[u] > bnd);
            }
            int q[UA];
            wave_alloc_n<UA>(&sh.n_cand, keep, q);
            P1STAMP(4)
#pragma unroll
            for (int u = 0; u < UA; u++) {
              if (keep[u]) {
                if (q[u] < cand_cap) {
                  i32x4 ca; ca.x = asi(tot[u]); ca.y = a[u]; ca.z = (int)((unsigned)(base + owner[u]) | ((unsigned)bko[u] << 26)); ca.w = arc[u].x;
                  cand[q[u]] = ca;
                } else sh.error = ERR_CAND;
              }
            }
            P1STAMP(5)
          });
        }
      }
      P1STAMP(6)
      mymin = block_min_f(mymin, sh, par);
      P1STAMP(7)
      const float next_cutoff = mymin;
      const int ncand = sh.n_cand < cand_cap ? sh.n_cand : cand_cap;      // (pass 2 does not move n_cand)
      // ordered admission: the buckets' final bounds (every arc's minimum is in: the reduction's barrier), lane b holds bucket b's
      const unsigned adm_pre = ADM == VBH_ADMIT_ORDERED ? adm_bounds(sh.bmin, &sh.init_bound) : 0u;
      PHASE(1)
      // ---- pass 2 (candidate-parallel): admit tot <= next_cutoff: the state's cost in the map is lowered (the thread whose
      //      compare-and-swap claims a first-level slot, or that meets a stale epoch in the second level, is the first to reach
      //      the state), and the forward link (:765-767) is written with a provisional destination -- map slot, or -1 - state
      //      for a second-level state -- that process_nonemitting replaces by the token once tokens are numbered
      {
        const EntMap ent{uptr(s.ent)};
        gp<i32x4> wt = uptr(s.w_tok);
        gp<int> front0 = uptr(s.front[0]);
        gp<i32x4> wl = uptr(s.wl);
        const int w_tok_cap = s.w_tok_cap, front_cap = s.front_cap, w_link_cap = s.w_link_cap;
#if VBH_CAND_PF
        // (the next round's candidate records are fetched before this round's are worked on: a round is otherwise a load the whole wave
        // waits for, then a stretch of LDS atomics during which nothing is in flight)
        i32x4 CAn[UC];
#pragma unroll
        for (int u = 0; u < UC; u++) { const int q = u * NT + tid; CAn[u] = cand[q < ncand ? q : 0]; }
#endif
        for (int base = 0; base < ncand; base += NT * UC) {
          i32x4 CA[UC]; bool adm[UC];         // {tot, arc, source token, next state | EPSBIT}
#if VBH_CAND_PF
#pragma unroll
          for (int u = 0; u < UC; u++) CA[u] = CAn[u];
          if (base + NT * UC < ncand) {
#pragma unroll
            for (int u = 0; u < UC; u++) { const int q = base + NT * UC + u * NT + tid; CAn[u] = cand[q < ncand ? q : 0]; }
          }
#endif
#pragma unroll
          for (int u = 0; u < UC; u++) {
            const int q = base + u * NT + tid;
            const bool ok = q < ncand;
#if !VBH_CAND_PF
            CA[u] = cand[ok ? q : 0];
#endif
            if (ADM == VBH_ADMIT_ORDERED) {
              const float bnd = o2f((unsigned)__shfl((int)adm_pre, (int)((unsigned)CA[u].z >> 26), 64));
              adm[u] = ok && !(asf(CA[u].x) > bnd);
              CA[u].z &= 0x03FFFFFF;
            } else adm[u] = ok && !(asf(CA[u].x) > next_cutoff);
          }
          int slot[UC]; bool create[UC], mk2[UC]; int nid[UC];
#pragma unroll
          for (int u = 0; u < UC; u++) {
            bool improved;
            slot[u] = -1; create[u] = false;
            if (adm[u]) map_relax(sh, ent, CA[u].w, asf(CA[u].x), ep, slot[u], create[u], improved);
            mk2[u] = create[u] && slot[u] < 0;
            DIAG_ADD(0, adm[u] && slot[u] < 0)
            DIAG_ADD(1, adm[u])
          }
          wave_alloc_n<UC>(&sh.w_ntok, mk2, nid);
          bool pushf[UC]; int fq[UC], lq[UC];
#pragma unroll
          for (int u = 0; u < UC; u++) {
            pushf[u] = create[u] && CA[u].w < 0;
            if (mk2[u]) {       // second-level state: its token is made now
              if (nid[u] < w_tok_cap) {
                i32x4 r; r.x = CA[u].w & ~PLDST; r.y = 0; r.z = 0; r.w = (int)INF_BITS;
                wt[nid[u]] = r;
                st_i32(&ent[CA[u].w & STMASK].tokid, nid[u]);
              } else { sh.error = ERR_W_TOK; pushf[u] = false; }
            }
          }
          wave_alloc_n<UC>(&sh.fcnt[0], pushf, fq);
          wave_alloc_n<UC>(&sh.w_nlink, adm, lq);
#if VBH_LAZY_DST
          {     // the links whose destination is a second-level state: listed for process_nonemitting (the others keep their slot)
            bool l2[UC]; int q2[UC];
#pragma unroll
            for (int u = 0; u < UC; u++) l2[u] = adm[u] && slot[u] < 0;
            wave_alloc_n<UC>(&sh.n_l2, l2, q2);
#pragma unroll
            for (int u = 0; u < UC; u++) if (l2[u] && q2[u] < 256) sh.hist[q2[u]] = (unsigned)lq[u];
          }
#endif
#pragma unroll
          for (int u = 0; u < UC; u++) {
            if (pushf[u]) front_push(sh, front0, front_cap, 0, fq[u], CA[u].w & STMASK);
            if (adm[u]) {
              if (lq[u] < w_link_cap) {
                i32x4 r; r.x = (tb + CA[u].z) | enc; r.y = slot[u] >= 0 ? slot[u] : -1 - (CA[u].w & STMASK); r.z = CA[u].x; r.w = CA[u].y;
                st_link(wl + lq[u], r);
              } else sh.error = ERR_W_LINK;
            }
          }
        }
      }
      __syncthreads();
      PHASE_D(8)
      const int le_emit = sh.w_nlink < s.w_link_cap ? sh.w_nlink : s.w_link_cap;
      if (sh.error != ERR_NONE) break;
      PHASE(2)
      // ---- ProcessNonemitting(next_cutoff)
      const int plb = process_nonemitting<ADM>(g, s, nb, next_cutoff, ep, 0, sh, lb, le_emit, cfg.lattice_beam, frame + 1);
      cur_b = nb; cur_e = sh.w_ntok < s.w_tok_cap ? sh.w_ntok : s.w_tok_cap; cur_w = true;
      const int pl_end = sh.w_nlink < s.w_link_cap ? sh.w_nlink : s.w_link_cap;
      if (tid == 0) {
        gp<int> r0 = (gp<int>)(s.fr + frame);
        r0[2] = lb; r0[3] = le_emit; r0[6] = r0[6] | 2; r0[9] = le_emit - lb;
        FrameRec r;
        r.tok_b = cur_b; r.tok_e = cur_e; r.el_b = 0; r.el_e = 0; r.pl_b = plb; r.pl_e = pl_end;
        r.flags = 1 | 4; r.cost_offset = 0.0f; r.a_tok = cur_e - cur_b; r.a_el = 0; r.a_pl = pl_end - plb; r.pad = 0;
        st_frame(s.fr + (frame + 1), r);
        sh.tph[11] += sh.n_cand;      // candidates pass 1 wrote, over the utterance (reported in UttResult::t_phase[12]; no room for a word of its own: the two-workgroups-per-CU build's LDS ends 16 bytes under 80 KB)
        sh.n_cand = 0; sh.cand_bound = 0xFFFFFFFFu; sh.fcnt[0] = 0; sh.fcnt[1] = 0; sh.fcnt[2] = 0; sh.n_l2 = 0;
      }
      if (ADM == VBH_ADMIT_ORDERED) {      // (their last readers: before pass 2; the barrier below)
        if (tid >= NT - 64) sh.bmin[tid - (NT - 64)] = 0xFFFFFFFFu;
        if (tid == 0) sh.init_bound = 0xFFFFFFFFu;
      }
#ifdef VBH_DIAG
      if (tid == 0) {
        if (cur_e - cur_b > (HCAP * 3) / 4) sh.diag[2]++;
        if (ncand > 2 * HCAP) sh.diag[3]++;
        const int cl = f_n0 < 512 ? 0 : f_n0 < 2048 ? 1 : f_n0 < 6144 ? 2 : 3;
        sh.fcl[cl] += clk() - f_t0; sh.fcl[4 + cl] += 1;
      }
#endif
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
      __threadfence_block();
      __syncthreads();
      const FrameRec fr = ld_frame(s.fr + F);
      const int tb = fr.tok_b, te = fr.tok_e;
      gp<i32x4> wt = uptr(s.w_tok), ht = uptr(s.h_tok);
      gp<i32x4> tok = uptr((fr.flags & 1) ? s.w_tok : s.h_tok);
      gp<i32x4> plk = uptr((fr.flags & 4) ? s.wl : s.hl);
      // ComputeFinalCosts (:532-572)
      float bc = INF, bwf = INF; unsigned long long anyfinal = 0;
      for (int t = tb + tid; t < te; t += NT) {
        const i32x4 r = tok[t];
        const float fc = asf(gst[r.x & STMASK].w), c = asf(r.y);
        bc = fminf(bc, c); bwf = fminf(bwf, c + fc);
        anyfinal += fc != INF;
      }
      bc = block_min_f(bc, sh, par); bwf = block_min_f(bwf, sh, par);
      anyfinal = block_sum_u64(anyfinal, sh, par);
      final_empty = anyfinal == 0;
      final_relative = (bc == INF && bwf == INF) ? INF : bwf - bc;
      final_best = (bwf != INF) ? bwf : bc;
      // PruneForwardLinksFinal (:376-458), to the fixed point
      const int pb = fr.pl_b, pe = fr.pl_e;
      for (int iter = 0; iter < MAX_EPS_ITERS; iter++) {
        for (int t = tb + tid; t < te; t += NT) {
          const i32x4 r = tok[t];
          const float fc = final_empty ? 0.0f : asf(gst[r.x & STMASK].w);
          const float x = asf(r.y) + fc - final_best;    // >= 0 by construction
          ((gp<int>)&tok[t])[3] = asi(x);
        }
        __syncthreads();
        const unsigned long long cp = eval_links<false>(plk, pb, pe, pe, wt, ht, cfg.lattice_beam, nullptr, 0, sh, s);
        if ((tid & 63) == 0 && (cp >> 32)) atomicAdd(&sh.cnt_pl[pp], (int)(cp >> 32));
        __syncthreads();
        if (tid == 0) { sh.changed2[pp ^ 1] = 0; sh.cnt_tok[pp ^ 1] = 0; sh.cnt_el[pp ^ 1] = 0; sh.cnt_pl[pp ^ 1] = 0; }
        int ch = 0, alive = 0;
        for (int t = tb + tid; t < te; t += NT) {
          const i32x4 r = tok[t];
          float v = asf(r.w);
          if (v > cfg.lattice_beam) v = INF;
          const unsigned vb = __float_as_uint(v);
          i32x2 nv; nv.x = (int)vb; nv.y = (int)INF_BITS;
          *(gp<i32x2>)((gp<int>)&tok[t] + 2) = nv;
          if (vb != (unsigned)r.z) ch = 1;
          alive += vb != INF_BITS;
        }
        if (ch) sh.changed2[pp] = 1;
        if (alive) atomicAdd(&sh.cnt_tok[pp], alive);
        __syncthreads();
        const bool any = sh.changed2[pp] != 0;
        const bool last = !any || pe == pb;
        if (last && tid == 0) {
          i32x4 c; c.x = sh.cnt_tok[pp]; c.y = 0; c.z = sh.cnt_pl[pp]; c.w = 0;
          ((gp<i32x4>)(s.fr + F))[2] = c;
        }
        pp ^= 1;
        if (last) break;
      }
      __syncthreads();
      prune_backward<ADM>(s, F - 1, /*first_unpruned=*/0, /*stop_at=*/0, cfg.lattice_beam, 0.0f, /*final=*/true, sh, pp);   // delta 0 (:606-611)
      PHASE(4)
      compact_window<ADM>(s, prev_front, F, sh);
      PHASE(5)
    }
    __syncthreads();

    // ---- write the surviving lattice to the dense pool (content of GetRawLattice :110-192).
    //      One wave per frame; two traversals (number + count, then write) so that the pool
    //      space is reserved with one atomic per utterance.
    int r_ok = 0, r_error = sh.error, r_reached_final = 0, r_n_tok = 0, r_n_link = 0;
    long long r_tok_off = 0, r_link_off = 0, r_cost_off = 0;
    if (search_ok && sh.error == ERR_NONE) {
      const int F = T;
      gp<const i32x4> ht = uptr(s.h_tok), hl = uptr(s.hl);
      gp<int> h_remap = uptr(s.h_remap);
      if (tid == 0) sh.flag = 0;
      __syncthreads();
      // dense index per alive history token (h_remap; -1 = dead)
      unsigned long long alive_last = 0;
      for (int f = wave; f <= F; f += NW) {
        const FrameRec fr = ld_frame(s.fr + f);
        for (int b0 = fr.tok_b; b0 < fr.tok_e; b0 += 64) {
          const int t = b0 + lane;
          const bool in = t < fr.tok_e;
          const bool alive = in && (unsigned)ht[in ? t : fr.tok_b].z != INF_BITS;
          const int d = wave_alloc(&sh.flag, alive);
          if (in) h_remap[t] = alive ? d : -1;
          if (f == F) alive_last += alive;
        }
      }
      alive_last = block_sum_u64(alive_last, sh, par);   // (barrier: h_remap and sh.flag complete)
      const unsigned long long na = (unsigned long long)sh.flag;
      unsigned long long nl = 0;
      for (int f = wave; f <= F; f += NW) {
        const FrameRec fr = ld_frame(s.fr + f);
        for (int pass = 0; pass < 2; pass++) {
          const int lb = pass ? fr.pl_b : fr.el_b, le_ = pass ? fr.pl_e : fr.el_e;
          for (int b0 = lb; b0 < le_; b0 += 64) {
            const int l = b0 + lane;
            const bool in = l < le_;
            const i32x4 L = hl[in ? l : lb];
            const bool live = in && L.x >= 0;
            const int a = h_remap[live ? L.x : 0], b = h_remap[live ? L.y : 0];
            nl += live && a >= 0 && b >= 0;
          }
        }
      }
      nl = block_sum_u64(nl, sh, par);
      if (tid == 0) sh.u0 = atomicAdd(&pool.counters[0], na);
      __syncthreads();
      const long long toff = (long long)sh.u0;
      __syncthreads();
      if (tid == 0) sh.u0 = atomicAdd(&pool.counters[1], nl);
      __syncthreads();
      const long long loff = (long long)sh.u0;
      __syncthreads();
      if (tid == 0) { sh.u0 = atomicAdd(&pool.counters[2], (unsigned long long)T); sh.flag = 0; }
      __syncthreads();
      const long long coff_off = (long long)sh.u0;
      const bool fits = toff + (long long)na <= pool.tok_cap && loff + (long long)nl <= pool.link_cap && coff_off + T <= pool.cost_cap;
      if (fits) {
        for (int f = wave; f <= F; f += NW) {
          const FrameRec fr = ld_frame(s.fr + f);
          for (int b0 = fr.tok_b; b0 < fr.tok_e; b0 += 64) {
            const int t = b0 + lane;
            const bool in = t < fr.tok_e;
            const i32x4 r = ht[in ? t : fr.tok_b];
            const int d = h_remap[in ? t : fr.tok_b];
            if (in && d >= 0) {
              const long long o = toff + d;
              const int st = r.x & STMASK;
              const i32x4 sr = gst[st];
              pool.tok_frame[o] = f; pool.tok_state[o] = sr.x; pool.tok_cost[o] = asf(r.y); pool.tok_extra[o] = asf(r.z);      // (the graph's own state id)
              float fin = INF;
              if (f == F) fin = final_empty ? 0.0f : asf(sr.w);
              pool.tok_final[o] = fin;
            }
          }
          for (int pass = 0; pass < 2; pass++) {
            const int lb = pass ? fr.pl_b : fr.el_b, le_ = pass ? fr.pl_e : fr.el_e;
            const float coff = (!pass && f < T) ? fr.cost_offset : 0.0f;
            for (int b0 = lb; b0 < le_; b0 += 64) {
              const int l = b0 + lane;
              const bool in = l < le_;
              const i32x4 L = hl[in ? l : lb];
              const int arc_id = in ? L.w : 0;
              const bool live = in && L.x >= 0;
              const int a = h_remap[live ? L.x : 0], b = h_remap[live ? L.y : 0];
              const bool w = live && a >= 0 && b >= 0;
              const int d = wave_alloc(&sh.flag, w);
              if (w) {
                const long long o = loff + d;
                const int il = ((gp<const int>)g.arc_il)[arc_id];
                const i32x4 arc = garcs[arc_id];
                pool.link_src[o] = a; pool.link_dst[o] = b;
                pool.link_il[o] = il; pool.link_ol[o] = arc.w; pool.link_gr[o] = asf(arc.y);
                // the acoustic cost as pass 1 computed it (cost_offset - acwt * score), then the offset taken out again (:148-152)
                float acst = 0.0f;
                if (il != 0) { acst = coff - cfg.acwt * loglikes[(ud.row0 + f) * (long long)P + arc.z]; acst = acst - coff; }
                pool.link_ac[o] = acst;
              }
            }
          }
        }
        for (int f = tid; f < T; f += NT) pool.cost_offsets[coff_off + f] = ((gp<const float>)(s.fr + f))[7];
        r_n_tok = (int)na; r_n_link = (int)nl; r_tok_off = toff; r_link_off = loff; r_cost_off = coff_off;
        r_reached_final = final_relative != INF;
        // Decode() returns whether the last frame still has tokens (:93-95)
        r_ok = alive_last > 0;
      } else r_error = ERR_POOL;
    }
    PHASE(6)
#ifdef VBH_DIAG_EPS
    if (tid == 0 && utt < 2)
      printf("utt %d: process_nonemitting cycles/frame: closure %lld, scan + token records %lld, second-level tokens %lld, link fix-up %lld, epsilon links %lld, last barrier %lld; emitting links excised at the fix-up %lld per frame\n",
             utt, sh.ept[0] / (T + 1), sh.ept[1] / (T + 1), sh.ept[2] / (T + 1), sh.ept[3] / (T + 1), sh.ept[4] / (T + 1), sh.ept[5] / (T + 1), sh.ept[6] / (T + 1));
#endif
    {
      const unsigned long long te_ = block_sum_u64((unsigned long long)(unsigned)c_tok_exp, sh, par);
      const unsigned long long ae_ = block_sum_u64((unsigned long long)(unsigned)c_arc_exp, sh, par);
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
        for (int i = 8; i < 16; i++) r->t_phase[i] = 0;
        r->t_phase[12] = sh.tph[11];
        if (r_error == ERR_INTERNAL) { r->t_phase[8] = sh.chk[0]; r->t_phase[9] = sh.chk[1]; r->t_phase[10] = sh.chk[2]; r->t_phase[11] = sh.chk[3]; }
#ifdef VBH_DIAG
        r->t_phase[8] = sh.tph[8]; r->t_phase[9] = sh.tph[9];
        for (int i = 0; i < 4; i++) r->t_phase[10 + i] = sh.diag[i];
        r->t_phase[13] = sh.diag[4];      // (replaces the candidates-per-slot counter: sweeps of the pruning passes)
        // [14], [15]: cycles and frames of the four frame classes, 16 bits of kilocycles-per-frame / frames each
        r->t_phase[14] = 0; r->t_phase[15] = 0;
        // heavy frames' phase cycles replace [0..6], [8], [9] when VBH_DIAG_HEAVY is defined
#ifdef VBH_DIAG_HEAVY
        for (int i = 0; i < 7; i++) r->t_phase[i] = sh.hph[i];
        r->t_phase[8] = sh.hph[8]; r->t_phase[9] = sh.hph[9];
#endif
#ifdef VBH_DIAG_P1
        for (int i = 0; i < 7; i++) r->t_phase[i] = sh.hph[i];
        r->t_phase[7] = sh.hph[7];
#endif
        for (int i = 0; i < 4; i++) {
          const long long kc = sh.fcl[4 + i] ? sh.fcl[i] / sh.fcl[4 + i] / 1000 : 0;
          r->t_phase[14] |= (kc > 65535 ? 65535 : kc) << (16 * i);
          r->t_phase[15] |= (sh.fcl[4 + i] > 65535 ? 65535 : sh.fcl[4 + i]) << (16 * i);
        }
#endif
        r->t_start = rt_start; r->t_end = (long long)__builtin_amdgcn_s_memrealtime(); r->wg = blockIdx.x; r->pad2 = 0;
      }
    }
    __syncthreads();
  }
#undef PHASE
#undef PHASE_D
#undef DIAG_ADD
  if (spool.flags) {
    // every wave waits for its OWN stores (and writes back its XCD's L2) before the barrier: a workgroup barrier orders the waves of a CU
    // through the in-order L1 path, which says nothing to the next owner of the slot on another CU
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_store(spool.flags + my_slot, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

#if !VBH_LAYOUT2
// per batch: pdf of every emitting arc (arc.z = tid2pdf[ilabel]; -1 for input-epsilon arcs)
__global__ void fill_arc_pdf_kernel(i32x4 *__restrict__ arcs, const int *__restrict__ arc_il, long long A,
                                    const int *__restrict__ tid2pdf) {
  for (long long a = blockIdx.x * (long long)blockDim.x + threadIdx.x; a < A; a += (long long)gridDim.x * blockDim.x) {
    const int il = arc_il[a];
    if (il >= 0) reinterpret_cast<int *>(&arcs[a])[2] = il == 0 ? -1 : tid2pdf[il];      // (-1: a state's record or padding of the packed array)
  }
}
#endif

// ---- the two builds of this file.  The library holds the search kernel twice (voicebridge_amd/build.py):
//   layout 1 = one workgroup of 1024 threads per CU, a 16384-slot state map: the default.  A heavy search (max-active 7000) is bound by
//              what one CU's units get through, and an utterance wants all of them (two half-size workgroups per CU: 977 against 925 ms);
//   layout 2 = two workgroups of 512 threads per CU, an 8192-slot map (this file with -DVBH_LAYOUT2 -DVBH_DECODE_NT=512
//              -DVBH_WAVES_PER_EU=4 -DVBH_HBITS=13).  A LIGHT search -- DecodeFmllr's first pass at beam 10 / max-active 2000: 670
//              tokens per frame -- is bound by the ~25 dependent phases of a frame, and two utterances per CU fill each other's waits:
//              392 -> 279 ms per 1024 x 30 s, same box (round 5).  Until then this build was the race detector of tests/test_gpu_variant.py
//              (waves skew far more beside another workgroup's); it still is, now inside the product.
// The second build consists of the device code above and the two entry points below; the host side (first build only) picks a layout
// per decoder (vbh_decode_layout_for: max_active <= 4000 -> 2; VBH_DECODE_LAYOUT=1|2 overrides) and sizes its slot pools by it.
struct LaunchArgs {
  int admission, grid; hipStream_t st;
  DevGraph dg; DecCfg dc; const Slot *slots; SlotPoolDev spool; const float *loglikes; int P;
  const UttDesc *utts; const int *work_list; int n_work; int *work_counter; UttResult *results; const Pool *pool;
};
void launch_decode(const LaunchArgs &a) {
  if (a.admission == VBH_ADMIT_ORDERED)
    hipLaunchKernelGGL(decode_kernel<VBH_ADMIT_ORDERED>, dim3(a.grid), dim3(NT), 0, a.st, a.dg, a.dc, a.slots, a.spool, a.loglikes, a.P, a.utts, a.work_list,
                       a.n_work, a.work_counter, a.results, a.pool);
  else
    hipLaunchKernelGGL(decode_kernel<VBH_ADMIT_EXACT>, dim3(a.grid), dim3(NT), 0, a.st, a.dg, a.dc, a.slots, a.spool, a.loglikes, a.P, a.utts, a.work_list,
                       a.n_work, a.work_counter, a.results, a.pool);
}

}  // namespace

#if VBH_LAYOUT2
extern "C" void vbh_decode_l2_launch(const void *args) { launch_decode(*static_cast<const LaunchArgs *>(args)); }
extern "C" void vbh_decode_l2_info(int *out3) { out3[0] = NT; out3[1] = HCAP; out3[2] = WG_PER_CU; }
#else
extern "C" void vbh_decode_l2_launch(const void *args);
extern "C" void vbh_decode_l2_info(int *out3);

// ============================================================================ host side
struct Layout { int id, nt, hcap, wg_per_cu; void (*launch)(const void *); };
static void launch_l1(const void *args) { launch_decode(*static_cast<const LaunchArgs *>(args)); }
static const Layout *layout_by_id(int id) {
  static const Layout l1{1, NT, HCAP, WG_PER_CU, launch_l1};
  static const Layout l2 = [] { int v[3]; vbh_decode_l2_info(v); return Layout{2, v[0], v[1], v[2], vbh_decode_l2_launch}; }();
  return id == 2 ? &l2 : &l1;
}
#ifndef VBH_LIGHT_MAX_ACTIVE
#define VBH_LIGHT_MAX_ACTIVE 4000      // (round 5, beam 13, ms per 1024 x 30 s in layout 1 / 2: max-active 1000: 358 / 238; 3000: 492 / 409; 4000: 562 / 521; 5000: 624 / 636; 7000: 752 / 977)
#endif
// Slots of a decoder's pool: one per workgroup that can be resident, or fewer for small batches.  A pool is shared by the decoders of a
// process (vbh_decoder_create_shared), and in layout 2 a job of 256 utterances fills only half of the device: the pool is made for the
// whole device from 128 utterances up, so that the grids of several job threads run side by side instead of spinning for a slot
// (four jobs of 256 sharing 256 slots gained nothing over layout 1: DecodeFmllr's first pass, round 5).
static int pool_slots(const Layout *lay, int max_batch, int dev_cus) {
  const int cap = lay->wg_per_cu * dev_cus;
  return (lay->id == 2 && max_batch >= 128) ? cap : std::min(max_batch, cap);
}
static int layout_for(const vbh_decode_cfg *cfg) {
  if (const char *e = getenv("VBH_DECODE_LAYOUT")) { if (!strcmp(e, "1")) return 1; if (!strcmp(e, "2")) return 2; }
  return cfg->max_active <= VBH_LIGHT_MAX_ACTIVE ? 2 : 1;
}
struct vbh_graph {
  // Only the device is remembered, not the creating context: a graph is shared by the threads of a process (the jobs of a
  // step), and the context of the thread that happened to build it may be gone by the time the last user lets go of it
  int device;
  int S, start; long long A; int max_ilabel, max_degree;
  DevBuf<i32x4> pk;            // the packed array (DevGraph): what both searches read
  DevBuf<int> pk_il;           // an arc's ilabel (-1: a state's record or padding)
  long long PKN = 0; int start_pk = 0;
  std::vector<int> h_pos;      // [S] a state's record in the packed array (per-utterance start states of the 1-best search are translated on the host)
  // tid2pdf table the arcs' pdf fields currently hold (empty: none) and the searches in flight that read them: decoders on
  // several host threads / streams may share a graph as long as they use the same transition model at the same time
  mutable std::vector<int> filled_t2p;
  mutable std::mutex pdf_mu;
  mutable int pdf_users = 0;
  std::vector<float> h_final;
};

void vbh_graph_view_get(const vbh_graph *g, vbh_graph_view *v) {
  v->S = g->S; v->start = g->start; v->A = g->A; v->max_ilabel = g->max_ilabel; v->max_degree = g->max_degree;
  v->PKN = g->PKN; v->start_pk = g->start_pk; v->pk = g->pk.p; v->pk_il = g->pk_il.p; v->h_pos = g->h_pos.data();
}

// The arcs' pdf fields depend on tid2pdf only: refill when the table changes.  A search that reads them holds a use count
// from vbh_graph_fill_pdf until vbh_graph_release_pdf; a different table is refused while another search is in flight.
int vbh_graph_fill_pdf(const vbh_graph *g, hipStream_t st, const int32_t *h_t2p, int32_t num_tids) {
  std::vector<int> t2p(h_t2p, h_t2p + num_tids + 1);
  t2p[0] = 0;
  std::lock_guard<std::mutex> lk(g->pdf_mu);
  if (g->A <= 0 || g->filled_t2p == t2p) { g->pdf_users++; return 0; }
  VBH_CHECK(g->pdf_users == 0, "the decoding graph is being searched with another transition model (tid2pdf table): decoders that share a "
                               "vbh_graph must use one table at a time");
  DevBuf<int> d_t2p;
  if (d_t2p.upload(t2p.data(), t2p.size(), st)) return -1;
  hipLaunchKernelGGL(fill_arc_pdf_kernel, dim3(1024), dim3(256), 0, st, g->pk.p, g->pk_il.p, g->PKN, d_t2p.p);
  VBH_HIP(hipGetLastError());
  VBH_HIP(hipStreamSynchronize(st));
  g->filled_t2p = t2p;
  g->pdf_users++;
  return 0;
}
void vbh_graph_release_pdf(const vbh_graph *g) {
  std::lock_guard<std::mutex> lk(g->pdf_mu);
  if (g->pdf_users > 0) g->pdf_users--;
}

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
  // the pool form (see claim_slot): n = n_xcd * n_per_xcd slots claimed through flags; flags.p == nullptr: slot = blockIdx.x
  DevBuf<int> flags;
  int n_per_xcd = 0, n_xcd = 1;
  int device = 0, max_frames = 0, graph_states = 0;
  int layout = 1;           // the build of the kernel these slots were made for (their slot -> token tables hold its map size)
  vbh_decode_cfg cfg{};
  std::vector<void *> allocs;
  SlotPoolDev dev() const { return SlotPoolDev{flags.p, n_per_xcd, n_xcd}; }
  ~SlotSet() {
    if (allocs.empty()) return;
    (void)hipSetDevice(device);
    (void)hipDeviceSynchronize();
    for (void *p : allocs) (void)hipFree(p);
  }
};

struct vbh_decoder {
  vbh_ctx *ctx; const vbh_graph *graph; vbh_decode_cfg cfg;
  const Layout *lay = nullptr;
  int max_frames;
  std::shared_ptr<SlotSet> small_set;   // the device's slot pool (shared by the decoders created with vbh_decoder_create_shared)
  SlotSet big_set;                      // generous capacities, private, allocated on first overflow
  std::vector<HostLattice> results;
  std::vector<UttResult> h_res;
  int n_retried = 0, n_pool_grown = 0, n_failed = 0;   // last batch: searched again in a big slot / pool enlargements / given up
  // output pool (device) and its pinned host mirror: kept across batches, grown on demand
  DevBuf<int> p_tf, p_ts, p_ls, p_ld, p_li, p_lo; DevBuf<float> p_tc, p_te, p_tfin, p_lg, p_la, p_co; DevBuf<unsigned long long> p_cnt;
  DevBuf<Pool> d_pool;
  long long pool_tok_cap = 0, pool_link_cap = 0, pool_cost_cap = 0;
  void *pin = nullptr; size_t pin_bytes = 0;
  float t_kernel_ms = 0, t_copy_ms = 0, t_host_ms = 0, t_total_ms = 0;   // last vbh_decode_batch
  // a batch between vbh_decode_batch_begin and vbh_decode_batch_end
  struct Pending {
    bool active = false;
    bool holds_pdf = false;           // a use count on the graph's pdf fields (vbh_graph_fill_pdf)
    int n_utt = 0, num_pdfs = 0;
    const float *d_loglikes = nullptr;
    long long tok_cap = 0, link_cap = 0;
    DevGraph dg{}; DecCfg dc{};
    std::chrono::steady_clock::time_point t_begin;
  } pend;
  DevBuf<UttDesc> d_utts; DevBuf<int> d_t2p, d_counter; DevBuf<UttResult> d_res;
  size_t utts_cap = 0, t2p_cap = 0;
  std::vector<UttDesc> h_utts; std::vector<int> h_t2p; Pool h_pool{};
  char *stage = nullptr; size_t stage_bytes = 0;   // pinned copy of the three above: source of the (truly) asynchronous uploads
  DevBuf<int> d_order; size_t order_cap = 0; std::vector<int> h_order;     // longest-first work order (see lpt_order)
  float t_probe_ms = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

// the pools of this process (weak: a pool lives as long as a decoder uses it), see vbh_decoder_create_shared
static std::mutex g_pool_mu;
static std::mutex g_pool_create_mu;
static std::vector<std::weak_ptr<SlotSet>> g_pools;

static int slot_alloc(vbh_decoder *d, SlotSet *set, size_t bytes, void **p, int memset_val = -2) {
  VBH_HIP(hipMalloc(p, bytes ? bytes : 8));
  if (getenv("VBH_DEBUG_ALLOC")) fprintf(stderr, "[vbh alloc] slot array [%p, %p) %zu B\n", *p, (void *)((char *)*p + bytes), bytes);
  set->allocs.push_back(*p);
  if (memset_val != -2) VBH_HIP(hipMemsetAsync(*p, memset_val, bytes, d->ctx->stream));
  return 0;
}

static const char *err_name(int e);

// Frames between two intermediate prunings.  The final lattice does not depend on this schedule
// (see the header), so the kernel prunes no more often than every 300 frames even when
// cfg.prune_interval asks for less: every visit of a frame costs the same handful of dependent
// passes however few links it holds, re-visits per frame fall with the interval (r02, same box: 251 / 247 / 244 / 241 / 243 ms per
// utterance at 100 / 140 / 180 / 300 / 450), and HBM has room for the longer raw window (~115 MB per search workspace).
// VBH_PRUNE_EVERY overrides (tuning, and tests of the pruning paths).
// VBH_ADMISSION=exact|ordered overrides the configuration (A/B runs of programs that do not expose the field)
static int admission_of(const vbh_decode_cfg *cfg) {
  int a = cfg->admission == VBH_ADMISSION_EXACT ? VBH_ADMIT_EXACT : VBH_ADMIT_ORDERED;
  if (const char *e = getenv("VBH_ADMISSION")) a = !strcmp(e, "exact") ? VBH_ADMIT_EXACT : !strcmp(e, "ordered") ? VBH_ADMIT_ORDERED : a;
  return a;
}
static int effective_prune_interval(const vbh_decode_cfg *cfg) {
  int v = std::max(cfg->prune_interval, 300);
  if (const char *e = getenv("VBH_PRUNE_EVERY")) { const int x = atoi(e); if (x > 0) v = x; }
  return std::min(v, MAX_WIN_FRAMES - 2);
}

// Regular slots are sized for the common case, not for the worst one: the raw window of forward links is cut to
// VBH_DEFAULT_LINK_SCALE of its worst-case size (a frame may leave 2 links per new token and 2 x max_active new tokens; over a
// pruning interval the bench's heaviest utterances stay under a sixth of that), the token window keeps its full size (it is
// the smaller array and the one heavy utterances do fill).  What outgrows a regular slot is searched again in a big slot,
// which costs that utterance a second search -- rare by construction; the memory saved is what lets a pool of two slots per
// CU and several batches of scores be resident on one device.
#ifndef VBH_DEFAULT_LINK_SCALE
#define VBH_DEFAULT_LINK_SCALE 0.35
#endif
static int alloc_slot_set(vbh_decoder *d, SlotSet *set, int n, bool big) {
  const vbh_graph *graph = d->graph;
  // (big slots follow the pool's configuration, which vbh_decoder_set_cfg cannot exceed)
  const vbh_decode_cfg *cfg = (big && d->small_set && d->small_set->n > 0) ? &d->small_set->cfg : &d->cfg;
  const int max_frames = d->max_frames;
  const long long S = graph->S;
  const long long ma = std::min<long long>(cfg->max_active, S);
  // new tokens per frame are bounded by the arcs leaving the expanded tokens (~max_active * out-degree)
  const long long per_frame_tok = std::min<long long>(S, big ? std::max<long long>(8192, 8 * ma) : std::max<long long>(4096, 2 * ma));
  long long per_frame_link = std::min<long long>(std::max<long long>(graph->A, 1), 2 * per_frame_tok);
  // VBH_SLOT_SCALE (tuning, and the tests of the retry path): scales the per-frame token, link and candidate capacities of
  // the small slots; an utterance that outgrows them is searched again in a big slot
  double slot_scale = 1.0;
  if (const char *e = getenv("VBH_SLOT_SCALE")) { const double x = atof(e); if (x >= 0.01 && x <= 4.0) slot_scale = x; }
  long long per_frame_tok_s = per_frame_tok;
  if (!big) {
    per_frame_link = std::max<long long>(1024, (long long)(per_frame_link * slot_scale * VBH_DEFAULT_LINK_SCALE));
    if (slot_scale < 1.0) per_frame_tok_s = std::max<long long>(256, (long long)(per_frame_tok * slot_scale));
  }
  const long long surv_tok = big ? 2048 : 160, surv_link = big ? 4096 : 320;   // lattice survivors per frame
  const int pi = effective_prune_interval(cfg);
  const int win = pi + 1;
  Slot p{};
  // (every frame's tokens and links start a 64-bit word of the survivor bitmaps, VBH_SURVIVOR_BITS: a frame takes whole words, which
  // for a graph of a dozen states -- config 1's monophone loop -- is several times what its tokens need; tests/test_gpu_config1.py)
  // (a FRAME holds fewer than 2^26 tokens -- a candidate's source token, an index within its frame, shares its word with the six bits of
  // the token's admission bucket, decode_kernel pass 1: checked per frame in the kernel -- the window as many as its indices allow)
  p.w_tok_cap = (int)std::min<long long>((((per_frame_tok_s + 63) & ~63LL) + 64) * win, 1LL << 28);
  p.w_link_cap = (int)std::min<long long>((((per_frame_link + per_frame_tok_s + 63) & ~63LL) + 64) * win, 1LL << 28);
  // history: one unpruned frontier frame per interval plus the survivors
  const long long n_int = max_frames / pi + 2;
  const long long front_avg = big ? per_frame_tok : per_frame_tok / 2;
  p.h_tok_cap = (int)std::min<long long>(front_avg * n_int + surv_tok * (max_frames + 1) + per_frame_tok, 1LL << 29);
  p.h_link_cap = (int)std::min<long long>(front_avg * n_int + surv_link * (max_frames + 1) + per_frame_link, 1LL << 29);
  p.front_cap = (int)(2 * per_frame_tok);
  // candidates of one frame = arcs leaving the expanded tokens: at most ~max_active tokens are
  // expanded (GetCutoff), each with at most max_degree arcs
  p.cand_cap = (int)std::min<long long>(std::max<long long>(graph->A, 1),
                                        std::min<long long>((ma + 256) * std::max(graph->max_degree, 1),
                                                            big ? (1LL << 24) : (long long)((1LL << 21) * std::min(1.0, slot_scale))));
  set->proto = p; set->n = n;
  set->layout = d->lay->id;
  set->device = d->ctx->device; set->max_frames = max_frames; set->graph_states = (int)(graph->PKN / 2 + 1); set->cfg = *cfg;      // (what the slots' second-level maps are sized by)
  set->h.assign(n, p);
  const size_t F = (size_t)max_frames + 2;
  // one allocation per field, covering all slots (slot i owns [i*count, (i+1)*count))
  const size_t ns = (size_t)n;
#define A_(field, type, count, ms)                                                                   \
  {                                                                                                  \
    type *base = nullptr;                                                                            \
    if (slot_alloc(d, set, sizeof(type) * (size_t)(count) * ns, reinterpret_cast<void **>(&base), ms)) return -1; \
    for (size_t i = 0; i < ns; i++) set->h[i].field = base + i * (size_t)(count);                    \
  }
  A_(ent, Ent, graph->PKN / 2 + 1, 0xFF)      // (EntMap)
  A_(w_tok, i32x4, p.w_tok_cap, -2) A_(w_remap, int, p.w_tok_cap, 0) A_(wl, i32x4, p.w_link_cap, -2)
  A_(w_tok_alive, unsigned long long, p.w_tok_cap / 64 + 2, 0) A_(w_lk_alive, unsigned long long, p.w_link_cap / 64 + 2, 0)
  A_(w_grp, unsigned long long, (size_t)GRP_RING * (size_t)(d->lay->hcap / 32), 0)
  A_(h_tok, i32x4, p.h_tok_cap, -2) A_(h_remap, int, p.h_tok_cap, 0) A_(hl, i32x4, p.h_link_cap, -2)
  A_(fr, FrameRec, F, 0)
  A_(cand, i32x4, p.cand_cap, -2)
  A_(front[0], int, p.front_cap, -2) A_(front[1], int, p.front_cap, -2) A_(front[2], int, p.front_cap, -2)
  A_(state, int, 4, 0)
#undef A_
  if (set->d.upload(set->h.data(), set->h.size(), d->ctx->stream)) return -1;
  if (!big) {
    set->n_xcd = 8; set->n_per_xcd = n / 8;       // (n is a multiple of 8, see vbh_decoder_create_shared)
    if (set->flags.alloc((size_t)n)) return -1;
    VBH_HIP(hipMemsetAsync(set->flags.p, 0, sizeof(int) * (size_t)n, d->ctx->stream));
  }
  return 0;
}

extern "C" {

void vbh_decode_cfg_default(vbh_decode_cfg *c) {
  // VB/scr/kaldi_scr.h:152-157 (Decode defaults) over lattice-faster-decoder.h:59-67
  c->beam = 13.0f; c->max_active = 7000; c->min_active = 200; c->lattice_beam = 6.0f;
  c->prune_interval = 25; c->beam_delta = 0.5f; c->acoustic_scale = 0.083333f;
  c->admission = VBH_ADMISSION_ORDERED;
}

int vbh_graph_create(vbh_ctx *ctx, int32_t num_states, int32_t start, const float *h_final,
                     const int64_t *h_arc_offsets, const vbh_arc *h_arcs, vbh_graph **out) {
  VBH_CHECK(ctx && h_final && h_arc_offsets && out, "vbh_graph_create: NULL argument");
  VBH_CHECK(num_states > 0 && start >= 0 && start < num_states, "vbh_graph_create: bad start state %d / %d states", start, num_states);
  VBH_CHECK(num_states <= STMASK, "vbh_graph_create: %d states, at most %d are supported (two tag bits ride on state ids)", num_states, STMASK);
  const long long A = h_arc_offsets[num_states];
  VBH_CHECK(h_arc_offsets[0] == 0 && A >= 0 && A < 2147483647LL, "vbh_graph_create: bad arc offsets");
  VBH_CHECK(A == 0 || h_arcs, "vbh_graph_create: arcs missing");
  if (vbh_use_device(ctx)) return -1;
  // (the two passes over the arcs run on a few host threads: a decoding graph has millions of them, and every job of a step waits
  // for this call)
  const int n_thr = (int)std::max(1u, std::min(8u, std::min(std::thread::hardware_concurrency(), (unsigned)(A / 200000 + 1))));
  auto parallel_states = [&](auto body) {        // body(thread index, first state, end state)
    std::vector<std::thread> th;
    for (int t = 1; t < n_thr; t++)
      th.emplace_back([&, t]() { body(t, (int)((long long)num_states * t / n_thr), (int)((long long)num_states * (t + 1) / n_thr)); });
    body(0, 0, (int)((long long)num_states / n_thr));
    for (auto &x : th) x.join();
  };
  std::vector<int> off(num_states + 1), nieps(num_states, 0);
  for (int s = 0; s <= num_states; s++) {
    VBH_CHECK(s == 0 || h_arc_offsets[s] >= h_arc_offsets[s - 1], "vbh_graph_create: arc offsets not monotone at state %d", s);
    off[s] = (int)h_arc_offsets[s];
  }
  std::vector<int> t_max_il(n_thr, 0), t_max_deg(n_thr, 0), t_bad_arc(n_thr, -1), t_bad_kind(n_thr, 0);
  parallel_states([&](int t, int s0, int s1) {
    int mi = 0, md = 0;
    for (int s = s0; s < s1; s++) {
      if (off[s + 1] - off[s] > md) md = off[s + 1] - off[s];
      int ne = 0;
      for (int a = off[s]; a < off[s + 1]; a++) {
        if (h_arcs[a].nextstate < 0 || h_arcs[a].nextstate >= num_states) { if (t_bad_arc[t] < 0) { t_bad_arc[t] = a; t_bad_kind[t] = 1; } continue; }
        if (h_arcs[a].ilabel < 0) { if (t_bad_arc[t] < 0) { t_bad_arc[t] = a; t_bad_kind[t] = 2; } continue; }
        ne += h_arcs[a].ilabel == 0;
        if (h_arcs[a].ilabel > mi) mi = h_arcs[a].ilabel;
      }
      nieps[s] = ne;
    }
    t_max_il[t] = mi; t_max_deg[t] = md;
  });
  int max_il = 0, max_deg = 0;
  for (int t = 0; t < n_thr; t++) {
    VBH_CHECK(t_bad_kind[t] != 1, "vbh_graph_create: arc %d leaves the graph", t_bad_arc[t]);
    VBH_CHECK(t_bad_kind[t] != 2, "vbh_graph_create: negative ilabel on arc %d", t_bad_arc[t]);
    max_il = std::max(max_il, t_max_il[t]); max_deg = std::max(max_deg, t_max_deg[t]);
  }
  // device records: per state {first arc, #epsilon arcs, #arcs, final cost}; arcs re-ordered
  // epsilon-first within a state (stable), next state tagged when it has epsilon arcs itself
  const size_t An = (size_t)std::max<long long>(A, 1);
  std::vector<i32x4> h_st(num_states);
  std::unique_ptr<i32x4[]> h_ar(new i32x4[An]);       // (every element is written below: no zero-fill of ~100 MB first)
  std::unique_ptr<int[]> h_il(new int[An]);
  if (A == 0) { h_ar[0] = i32x4{0, 0, 0, 0}; h_il[0] = 0; }
  parallel_states([&](int, int s0, int s1) {
    for (int s = s0; s < s1; s++) {
      int fb; memcpy(&fb, &h_final[s], 4);
      i32x4 r; r.x = off[s]; r.y = nieps[s]; r.z = off[s + 1] - off[s]; r.w = fb;
      h_st[s] = r;
      int pe = off[s], pn = off[s] + nieps[s];
      for (int a = off[s]; a < off[s + 1]; a++) {
        const int p = h_arcs[a].ilabel == 0 ? pe++ : pn++;
        i32x4 x; x.x = h_arcs[a].nextstate | (nieps[h_arcs[a].nextstate] > 0 ? EPSBIT : 0);
        int wb; memcpy(&wb, &h_arcs[a].weight, 4);
        x.y = wb; x.z = -1; x.w = h_arcs[a].olabel;
        h_ar[p] = x; h_il[p] = h_arcs[a].ilabel;
      }
    }
  });
  // the packed array (DevGraph): where every state's record goes -- a block of a record and its arcs that fits a 64-byte line (four
  // records) does not straddle one, a larger block starts one
  std::vector<int> pos(num_states);
  long long pkn = 0;
  for (int s = 0; s < num_states; s++) {
    const int k = std::max(2, 1 + (off[s + 1] - off[s]));      // (at least two records: EntMap)
    if ((pkn & 3) + k > 4 && (pkn & 3) != 0) pkn = (pkn + 3) & ~3LL;
    VBH_CHECK(pkn + k <= (long long)STMASK, "vbh_graph_create: %d states and %lld arcs do not fit the %d records the lattice search's packed graph can address",
              num_states, A, STMASK);
    pos[s] = (int)pkn;
    pkn += k;
  }
  std::unique_ptr<i32x4[]> h_pk(new i32x4[(size_t)pkn]);
  std::unique_ptr<int[]> h_pil(new int[(size_t)pkn]);
  parallel_states([&](int, int s0, int s1) {
    for (int s = s0; s < s1; s++) {
      const int p0 = pos[s], k = off[s + 1] - off[s];
      i32x4 r = h_st[s]; r.x = s;
      h_pk[p0] = r; h_pil[p0] = -1;
      for (int j = 0; j < k; j++) {
        i32x4 x = h_ar[off[s] + j];      // (already epsilon-first)
        x.x = pos[x.x & STMASK] | (x.x & EPSBIT);
        h_pk[p0 + 1 + j] = x; h_pil[p0 + 1 + j] = h_il[off[s] + j];
      }
      const long long end = s + 1 < num_states ? pos[s + 1] : pkn;
      for (long long q = (long long)p0 + 1 + k; q < end; q++) { h_pk[q] = i32x4{0, 0, 0, 0}; h_pil[q] = -1; }      // padding: never addressed
    }
  });
  vbh_graph *g = new vbh_graph();
  g->device = ctx->device; g->S = num_states; g->start = start; g->A = A; g->max_ilabel = max_il; g->max_degree = max_deg;
  g->PKN = pkn; g->start_pk = pos[start]; g->h_pos = pos;
  g->h_final.assign(h_final, h_final + num_states);
  hipStream_t st = ctx->stream;
  if (g->pk.upload(h_pk.get(), (size_t)pkn, st) || g->pk_il.upload(h_pil.get(), (size_t)pkn, st)) { delete g; return -1; }
  VBH_HIP(hipStreamSynchronize(st));
  *out = g;
  return 0;
}

void vbh_graph_destroy(vbh_graph *g) {
  if (!g) return;
  // (no context to drain: whoever still searches this graph holds a decoder on it, and decoders are destroyed first; the
  // device-wide wait covers streams of contexts that are gone)
  (void)hipSetDevice(g->device);
  (void)hipDeviceSynchronize();
  delete g;
}

int vbh_decoder_create_shared(vbh_ctx *ctx, const vbh_graph *graph, const vbh_decode_cfg *cfg, int32_t max_batch,
                              int32_t max_frames, vbh_decoder *share_slots_with, vbh_decoder **out) {
  VBH_CHECK(ctx && graph && cfg && out, "vbh_decoder_create: NULL argument");
  VBH_CHECK(cfg->beam > 0.0f && cfg->max_active > 1 && cfg->lattice_beam > 0.0f && cfg->prune_interval > 0 && cfg->beam_delta > 0.0f,
            "vbh_decoder_create: invalid config (LatticeFasterDecoderConfig::Check)");
  VBH_CHECK(cfg->min_active >= 0, "vbh_decoder_create: min_active < 0");
  VBH_CHECK(max_batch >= 1 && max_frames >= 0, "vbh_decoder_create: bad batch/frames");
  if (vbh_use_device(ctx)) return -1;
  vbh_decoder *d = new vbh_decoder();
  d->ctx = ctx; d->graph = graph; d->cfg = *cfg; d->max_frames = max_frames;
  d->lay = share_slots_with ? share_slots_with->lay : layout_by_id(layout_for(cfg));
  if (share_slots_with) {
    const SlotSet *o = share_slots_with->small_set.get();
    const bool ok = o && o->device == ctx->device && o->graph_states == (int)(graph->PKN / 2 + 1) && o->max_frames >= max_frames &&
                    o->cfg.max_active >= cfg->max_active && effective_prune_interval(&o->cfg) == effective_prune_interval(cfg);
    if (!ok) { delete d; vbh_set_error("vbh_decoder_create_shared: the other decoder's search slots do not fit (device, graph size, max_frames, max_active or pruning interval differ)"); return -1; }
    d->small_set = share_slots_with->small_set;
    *out = d;
    return 0;
  }
  // Decoders of one process share the device's pool whenever an existing one fits (same graph size, capacities that cover
  // this configuration, at least as many slots): the jobs of a step running on several host threads, or a caller's decoders
  // on several streams, then hold ONE set of workspaces between them.  VBH_PRIVATE_SLOTS=1: every decoder its own.
  // Looking for a pool and creating one is one critical section: job threads that start together must end up with one pool
  // between them, not one each (four of ~20 GB each, twice per DecodeFmllr, did not fit the device).
  std::unique_lock<std::mutex> create_lk(g_pool_create_mu, std::defer_lock);
  if (!getenv("VBH_PRIVATE_SLOTS")) {
    create_lk.lock();
    int dev_cus0 = ctx->num_cus;
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess) dev_cus0 = prop.multiProcessorCount; }
    const int want_n = (pool_slots(d->lay, max_batch, dev_cus0) + 7) & ~7;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (auto it = g_pools.begin(); it != g_pools.end();) {
      std::shared_ptr<SlotSet> o = it->lock();
      if (!o) { it = g_pools.erase(it); continue; }
      if (o->layout == d->lay->id && o->device == ctx->device && o->graph_states == (int)(graph->PKN / 2 + 1) && o->max_frames >= max_frames && o->cfg.max_active >= cfg->max_active &&
          effective_prune_interval(&o->cfg) == effective_prune_interval(cfg) && o->n >= want_n) {
        d->small_set = o;
        *out = d;
        return 0;
      }
      ++it;
    }
  }
  // Two size classes.  Regular slots: sized for the common case, one per workgroup that can be resident on the device (the
  // pool claim_slot() draws from), or fewer for small batches.  "big": the generous worst-case sizes, a handful of slots,
  // allocated only when an utterance overflows a regular slot; such utterances are searched again there (vbh_decode_batch),
  // so capacity never truncates a result.
  int dev_cus = ctx->num_cus;
  { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess) dev_cus = prop.multiProcessorCount; }
  int n = pool_slots(d->lay, max_batch, dev_cus);
  n = (n + 7) & ~7;
  d->small_set = std::make_shared<SlotSet>();
  if (alloc_slot_set(d, d->small_set.get(), n, /*big=*/false)) { vbh_decoder_destroy(d); return -1; }
  VBH_HIP(hipStreamSynchronize(ctx->stream));
  if (!getenv("VBH_PRIVATE_SLOTS")) { std::lock_guard<std::mutex> lk(g_pool_mu); g_pools.push_back(d->small_set); }
  *out = d;
  return 0;
}

int vbh_decoder_create(vbh_ctx *ctx, const vbh_graph *graph, const vbh_decode_cfg *cfg, int32_t max_batch,
                       int32_t max_frames, vbh_decoder **out) {
  return vbh_decoder_create_shared(ctx, graph, cfg, max_batch, max_frames, nullptr, out);
}

void vbh_decoder_destroy(vbh_decoder *d) {
  if (!d) return;
  (void)hipSetDevice(d->ctx->device);
  (void)hipStreamSynchronize(d->ctx->stream);
  d->small_set.reset();
  if (d->pin) (void)hipHostFree(d->pin);
  if (d->stage) (void)hipHostFree(d->stage);
  if (d->ev0) (void)hipEventDestroy(d->ev0);
  if (d->ev1) (void)hipEventDestroy(d->ev1);
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
    case ERR_NOT_RUN: return "no workgroup searched the utterance";
    case ERR_LDS: return "the workgroup's LDS was overwritten from outside (canary, diagnostic build)";
    case ERR_INTERNAL: return "internal consistency check failed (an index out of its array)";
    default: return "unknown";
  }
}

extern "C" {

// the search kernel of the decoder's layout, one instantiation per admission rule (DecCfg::admission)
#define VBH_LAUNCH_DECODE(grid_, dg_, dc_, slots_, spool_, ll_, P_, utts_, list_, n_work_, counter_, res_, pool_)                          \
  {                                                                                                                                     \
    const LaunchArgs a_{admission_of(&d->cfg), (int)(grid_), st, dg_, dc_, slots_, spool_, ll_, P_, utts_, list_, n_work_, counter_, res_, pool_}; \
    d->lay->launch(&a_);                                                                                                                \
  }
static int decode_batch_begin_impl(vbh_decoder *d, const float *d_loglikes, int32_t num_pdfs, const int64_t *h_frame_offsets,
                                   int32_t n_utt, const int32_t *h_tid2pdf, int32_t num_tids) {
  VBH_CHECK(d && h_frame_offsets && h_tid2pdf, "vbh_decode_batch: NULL argument");
  VBH_CHECK(n_utt >= 0 && num_pdfs > 0 && num_tids >= 0, "vbh_decode_batch: bad sizes");
  VBH_CHECK(!d->pend.active, "vbh_decode_batch_begin: the previous batch has not been ended");
  if (vbh_use_device(d->ctx)) return -1;
  d->results.clear(); d->h_res.clear();
  d->t_kernel_ms = d->t_copy_ms = d->t_host_ms = d->t_total_ms = 0;
  d->pend = vbh_decoder::Pending();
  d->pend.t_begin = std::chrono::steady_clock::now();
  d->pend.n_utt = n_utt;
  d->pend.active = true;
  if (n_utt == 0) return 0;
  d->pend.active = false;      // set again once the launch has succeeded
  if (!d->ev0) { VBH_HIP(hipEventCreate(&d->ev0)); VBH_HIP(hipEventCreate(&d->ev1)); }
  const vbh_graph *g = d->graph;
  // the graph's ilabels must be valid transition-ids ("Likely graph/model mismatch",
  // decodable-am-diag-gmm.cc:32-33) -- checked once per batch on the tid2pdf table
  VBH_CHECK(g->max_ilabel <= num_tids, "vbh_decode_batch: graph uses transition-id %d but the model has %d (graph/model mismatch)", g->max_ilabel, num_tids);
  for (int t = 1; t <= num_tids; t++)
    VBH_CHECK(h_tid2pdf[t] >= 0 && h_tid2pdf[t] < num_pdfs, "vbh_decode_batch: tid2pdf[%d] = %d out of range", t, h_tid2pdf[t]);
  std::vector<UttDesc> &utts = d->h_utts;
  utts.assign(n_utt, UttDesc{0, 0, 0});
  long long total_frames = 0;
  for (int u = 0; u < n_utt; u++) {
    const long long T = h_frame_offsets[u + 1] - h_frame_offsets[u];
    VBH_CHECK(T >= 0 && T <= d->max_frames, "vbh_decode_batch: utterance %d has %lld frames (decoder built for %d)", u, T, d->max_frames);
    utts[u] = UttDesc{h_frame_offsets[u], (int)T, 0};
    total_frames += T;
  }
  VBH_CHECK(total_frames == 0 || d_loglikes, "vbh_decode_batch: NULL loglikes");
  hipStream_t st = d->ctx->stream;
  std::vector<int> &t2p = d->h_t2p;
  t2p.assign(h_tid2pdf, h_tid2pdf + num_tids + 1);
  t2p[0] = 0;
  if ((size_t)n_utt > d->utts_cap) { if (d->d_utts.alloc(n_utt) || d->d_res.alloc(n_utt)) return -1; d->utts_cap = n_utt; }
  if (t2p.size() > d->t2p_cap) { if (d->d_t2p.alloc(t2p.size())) return -1; d->t2p_cap = t2p.size(); }
  if (!d->d_counter.p && d->d_counter.alloc(1)) return -1;
  // uploads come from pinned memory: a copy from pageable memory blocks the host until the GPU
  // has run it, i.e. behind whatever another context's search is doing to the CUs
  const size_t utt_b = (sizeof(UttDesc) * n_utt + 255) & ~(size_t)255, t2p_b = (sizeof(int) * t2p.size() + 255) & ~(size_t)255;
  if (utt_b + t2p_b + 256 > d->stage_bytes) {
    if (d->stage) (void)hipHostFree(d->stage);
    d->stage = nullptr; d->stage_bytes = 0;
    VBH_HIP(hipHostMalloc(reinterpret_cast<void **>(&d->stage), utt_b + t2p_b + 4096, hipHostMallocDefault));
    d->stage_bytes = utt_b + t2p_b + 4096;
  }
  memcpy(d->stage, utts.data(), sizeof(UttDesc) * n_utt);
  memcpy(d->stage + utt_b, t2p.data(), sizeof(int) * t2p.size());
  VBH_HIP(hipMemcpyAsync(d->d_utts.p, d->stage, sizeof(UttDesc) * n_utt, hipMemcpyHostToDevice, st));
  VBH_HIP(hipMemcpyAsync(d->d_t2p.p, d->stage + utt_b, sizeof(int) * t2p.size(), hipMemcpyHostToDevice, st));
  VBH_HIP(hipMemsetAsync(d->d_counter.p, 0, sizeof(int), st));
  // output pool: the pruned lattice is a small multiple of the frame count
  Pool &pool = d->h_pool;
  pool = Pool{};
  // (an estimate: a batch that needs more gets an enlarged pool in vbh_decode_batch_end; VBH_POOL_SCALE shrinks it for the tests of that path)
  double pool_scale = 1.0;
  if (const char *e = getenv("VBH_POOL_SCALE")) { const double x = atof(e); if (x > 0.0 && x <= 16.0) pool_scale = x; }
  const long long tok_cap = std::max<long long>(4096, (long long)(pool_scale * 64 * (total_frames + n_utt))),
                  link_cap = std::max<long long>(8192, (long long)(pool_scale * 128 * (total_frames + n_utt)));
  pool.tok_cap = tok_cap; pool.link_cap = link_cap; pool.cost_cap = total_frames + 1;
  DevBuf<int> &p_tf = d->p_tf, &p_ts = d->p_ts, &p_ls = d->p_ls, &p_ld = d->p_ld, &p_li = d->p_li, &p_lo = d->p_lo;
  DevBuf<float> &p_tc = d->p_tc, &p_te = d->p_te, &p_tfin = d->p_tfin, &p_lg = d->p_lg, &p_la = d->p_la, &p_co = d->p_co;
  DevBuf<unsigned long long> &p_cnt = d->p_cnt;
  if (tok_cap > d->pool_tok_cap) {
    if (p_tf.alloc(tok_cap) || p_ts.alloc(tok_cap) || p_tc.alloc(tok_cap) || p_te.alloc(tok_cap) || p_tfin.alloc(tok_cap)) return -1;
    d->pool_tok_cap = tok_cap;
  }
  if (link_cap > d->pool_link_cap) {
    if (p_ls.alloc(link_cap) || p_ld.alloc(link_cap) || p_li.alloc(link_cap) || p_lo.alloc(link_cap) || p_lg.alloc(link_cap) || p_la.alloc(link_cap)) return -1;
    d->pool_link_cap = link_cap;
  }
  if (pool.cost_cap > d->pool_cost_cap) { if (p_co.alloc(pool.cost_cap)) return -1; d->pool_cost_cap = pool.cost_cap; }
  if (!p_cnt.p && p_cnt.alloc(4)) return -1;
  VBH_HIP(hipMemsetAsync(p_cnt.p, 0, 4 * sizeof(unsigned long long), st));
  pool.tok_frame = p_tf.p; pool.tok_state = p_ts.p; pool.tok_cost = p_tc.p; pool.tok_extra = p_te.p; pool.tok_final = p_tfin.p;
  pool.link_src = p_ls.p; pool.link_dst = p_ld.p; pool.link_il = p_li.p; pool.link_ol = p_lo.p; pool.link_gr = p_lg.p; pool.link_ac = p_la.p;
  pool.cost_offsets = p_co.p; pool.counters = p_cnt.p;
  if (vbh_graph_fill_pdf(g, st, h_tid2pdf, num_tids)) return -1;
  d->pend.holds_pdf = true;
  DevBuf<Pool> &d_pool = d->d_pool;
  if (!d_pool.p && d_pool.alloc(1)) return -1;
  const DevGraph dg0{(int)g->PKN, g->start_pk, g->pk.p, g->pk.p, g->pk_il.p};
  const DecCfg dc0{d->cfg.beam, d->cfg.lattice_beam, d->cfg.beam_delta, d->cfg.acoustic_scale, d->cfg.max_active, d->cfg.min_active, effective_prune_interval(&d->cfg), getenv("VBH_NO_LDS_MAP") ? 0 : 1,
                      getenv("VBH_PRUNE_BACK") ? std::max(0, atoi(getenv("VBH_PRUNE_BACK"))) : effective_prune_interval(&d->cfg)};
  // Optional longest-first order for the persistent grid (VBH_LPT_PROBE_FRAMES=n, off by default).
  // Search times differ several-fold between utterances of equal length (how well the audio matches
  // the model decides how many tokens stay in the beam), and with a few utterances per workgroup one
  // heavy utterance picked up late leaves CUs idle at the end (35 % of workgroup-time at 1024
  // utterances).  With the option the grid first searches the first n frames of every utterance with
  // the output pool closed and the real search takes the utterances in order of decreasing probe
  // time.  Measured: 64 frames predict too little (busy 66 % -> 68 %), 128 frames help (80 %) but cost
  // what they gain, and the probe blocks the host; overlapping consecutive batches on two streams
  // (vbh_decode_batch_begin/_end) removes the idle tail without guessing, so that is what bench.py does.
  const int *d_work_list = nullptr;
  d->t_probe_ms = 0;
  {
    int probe_frames = 0;
    if (const char *e = getenv("VBH_LPT_PROBE_FRAMES")) probe_frames = atoi(e);
    if (probe_frames > 0 && n_utt > d->small_set->n) {
      const auto tp0 = std::chrono::steady_clock::now();
      std::vector<UttDesc> probe(utts);
      for (auto &u : probe) u.T = std::min(u.T, probe_frames);
      Pool closed = pool; closed.tok_cap = 0; closed.link_cap = 0; closed.cost_cap = 0;
      VBH_HIP(hipMemcpyAsync(d->d_utts.p, probe.data(), sizeof(UttDesc) * n_utt, hipMemcpyHostToDevice, st));
      VBH_HIP(hipMemcpyAsync(d->d_pool.p, &closed, sizeof(Pool), hipMemcpyHostToDevice, st));
      VBH_LAUNCH_DECODE(std::min(d->lay->wg_per_cu * d->ctx->num_cus, n_utt), dg0, dc0, d->small_set->d.p, d->small_set->dev(), d_loglikes,
                         (int)num_pdfs, d->d_utts.p, (const int *)nullptr, n_utt, d->d_counter.p, d->d_res.p, d->d_pool.p);
      VBH_HIP(hipGetLastError());
      std::vector<UttResult> pr(n_utt);
      VBH_HIP(hipMemcpyAsync(pr.data(), d->d_res.p, sizeof(UttResult) * n_utt, hipMemcpyDeviceToHost, st));
      VBH_HIP(hipStreamSynchronize(st));
      d->h_order.resize(n_utt);
      for (int u = 0; u < n_utt; u++) d->h_order[u] = u;
      // probe time per frame searched, scaled to the utterance's length
      std::vector<double> w(n_utt);
      for (int u = 0; u < n_utt; u++) w[u] = (double)(pr[u].t_end - pr[u].t_start) / std::max(1, probe[u].T) * std::max(1, utts[u].T);
      std::stable_sort(d->h_order.begin(), d->h_order.end(), [&](int a, int b) { return w[a] > w[b]; });
      if ((size_t)n_utt > d->order_cap) { if (d->d_order.alloc(n_utt)) return -1; d->order_cap = n_utt; }
      VBH_HIP(hipMemcpyAsync(d->d_order.p, d->h_order.data(), sizeof(int) * n_utt, hipMemcpyHostToDevice, st));
      VBH_HIP(hipMemcpyAsync(d->d_utts.p, utts.data(), sizeof(UttDesc) * n_utt, hipMemcpyHostToDevice, st));
      VBH_HIP(hipMemsetAsync(d->d_counter.p, 0, sizeof(int), st));
      VBH_HIP(hipMemsetAsync(p_cnt.p, 0, 4 * sizeof(unsigned long long), st));
      d_work_list = d->d_order.p;
      d->t_probe_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - tp0).count();
      if (getenv("VBH_DEBUG_TIMES")) fprintf(stderr, "[vbh] %-28s %8.2f ms\n", "longest-first probe", d->t_probe_ms);
    }
  }
  memcpy(d->stage + utt_b + t2p_b, &pool, sizeof(Pool));
  VBH_HIP(hipMemcpyAsync(d->d_pool.p, d->stage + utt_b + t2p_b, sizeof(Pool), hipMemcpyHostToDevice, st));
  d->pend.dg = dg0;
  d->pend.dc = dc0;
  d->pend.d_loglikes = d_loglikes; d->pend.num_pdfs = num_pdfs; d->pend.tok_cap = tok_cap; d->pend.link_cap = link_cap;
  {
    // one workgroup per slot the context's CUs can hold (a CU-masked context counts its own CUs), never more than the pool has
    int grid = std::min(std::min(d->small_set->n, d->lay->wg_per_cu * d->ctx->num_cus), n_utt);
    // VBH_SEARCH_MAX_WGS (diagnostic): a search grid smaller than the device.  A persistent search workgroup holds its CU until the grid has no
    // utterance left, so this keeps CUs free for the kernels other job threads issue beside a search (statistics, rescoring, the next job's
    // scoring).  Measured r04 on DecodeFmllr, 4 jobs of 256 utterances on one GPU: 3.56 s uncapped, 4.00 / 4.27 / 4.24 s at 224 / 192 / 160
    // workgroups -- a job is one utterance per CU, and the utterances that do not fit the grid double the job's search time.  Not a default.
    if (const char *e = getenv("VBH_SEARCH_MAX_WGS")) { const int cap_wgs = atoi(e); if (cap_wgs > 0) grid = std::max(1, std::min(grid, cap_wgs)); }
    VBH_HIP(hipMemsetAsync(d->d_res.p, 0xFF, sizeof(UttResult) * (size_t)n_utt, st));     // error = -1: "no workgroup searched it" until one does
    VBH_HIP(hipEventRecord(d->ev0, st));
    VBH_LAUNCH_DECODE(grid, d->pend.dg, d->pend.dc, d->small_set->d.p, d->small_set->dev(), d_loglikes, (int)num_pdfs,
                       d->d_utts.p, d_work_list, n_utt, d->d_counter.p, d->d_res.p, d_pool.p);
    VBH_HIP(hipGetLastError());
    VBH_HIP(hipEventRecord(d->ev1, st));
  }
  d->pend.active = true;
  return 0;
}

int vbh_decode_batch_begin(vbh_decoder *d, const float *d_loglikes, int32_t num_pdfs, const int64_t *h_frame_offsets,
                           int32_t n_utt, const int32_t *h_tid2pdf, int32_t num_tids) {
  const int rc = decode_batch_begin_impl(d, d_loglikes, num_pdfs, h_frame_offsets, n_utt, h_tid2pdf, num_tids);
  if (rc != 0 && d && d->pend.holds_pdf) { vbh_graph_release_pdf(d->graph); d->pend.holds_pdf = false; }
  return rc;
}

static int decode_batch_end_impl(vbh_decoder *d);
int vbh_decode_batch_end(vbh_decoder *d) {
  const int rc = decode_batch_end_impl(d);
  if (d && d->pend.holds_pdf) { vbh_graph_release_pdf(d->graph); d->pend.holds_pdf = false; }
  return rc;
}

static int decode_batch_end_impl(vbh_decoder *d) {
  VBH_CHECK(d, "vbh_decode_batch_end: NULL argument");
  VBH_CHECK(d->pend.active, "vbh_decode_batch_end: no batch has been begun");
  if (vbh_use_device(d->ctx)) return -1;
  d->pend.active = false;
  const int n_utt = d->pend.n_utt;
  if (n_utt == 0) return 0;
  const vbh_graph *g = d->graph;
  hipStream_t st = d->ctx->stream;
  const long long tok_cap = d->pend.tok_cap, link_cap = d->pend.link_cap;
  const float *d_loglikes = d->pend.d_loglikes; const int num_pdfs = d->pend.num_pdfs;
  const DevGraph dg = d->pend.dg; const DecCfg dc = d->pend.dc;
  const auto t_begin = d->pend.t_begin;
  hipEvent_t ev0 = d->ev0, ev1 = d->ev1;
  DevBuf<UttDesc> &d_utts = d->d_utts; DevBuf<int> &d_counter = d->d_counter; DevBuf<UttResult> &d_res = d->d_res;
  DevBuf<Pool> &d_pool = d->d_pool;
  DevBuf<int> &p_tf = d->p_tf, &p_ts = d->p_ts, &p_ls = d->p_ls, &p_ld = d->p_ld, &p_li = d->p_li, &p_lo = d->p_lo;
  DevBuf<float> &p_tc = d->p_tc, &p_te = d->p_te, &p_tfin = d->p_tfin, &p_lg = d->p_lg, &p_la = d->p_la;
  DevBuf<unsigned long long> &p_cnt = d->p_cnt;
  const bool dbg_t = getenv("VBH_DEBUG_TIMES") != nullptr;
  auto t_lap = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!dbg_t) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[vbh] %-28s %8.2f ms\n", what, std::chrono::duration<float, std::milli>(now - t_lap).count());
    t_lap = now;
  };
  d->h_res.resize(n_utt);
  {
    VBH_HIP(hipMemcpyAsync(d->h_res.data(), d_res.p, sizeof(UttResult) * n_utt, hipMemcpyDeviceToHost, st));
    VBH_HIP(hipStreamSynchronize(st));
    float ms = 0; VBH_HIP(hipEventElapsedTime(&ms, ev0, ev1)); d->t_kernel_ms += ms;
  }
  lap("search kernel + results D2H");
  // Capacities are estimates, results are not: an utterance that outgrew a small slot is searched again in a big one, and
  // when the lattices of a batch outgrow the output pool the pool is enlarged (its content kept) and the utterances that
  // did not fit are searched again.  What still fails after that -- a big slot too small, an epsilon cycle -- fails alone:
  // the utterance is reported with ok = 0 / error set and the rest of the batch is delivered (the reference logs
  // "Failed to decode" for one utterance and goes on, K/decoder/decoder-wrappers.cc:227-231).
  auto is_cap = [](int e) { return e == ERR_W_TOK || e == ERR_W_LINK || e == ERR_H_TOK || e == ERR_H_LINK || e == ERR_FRONT || e == ERR_CAND || e == ERR_NOT_RUN; };
  for (int u = 0; u < n_utt; u++) if (d->h_res[u].error < 0) { d->h_res[u] = UttResult{}; d->h_res[u].error = ERR_NOT_RUN; }
  long long tok_cap_now = tok_cap, link_cap_now = link_cap;
  std::vector<char> tried_big(n_utt, 0);
  d->n_retried = 0; d->n_pool_grown = 0; d->n_failed = 0;
  for (int round = 0; round < 6; round++) {
    std::vector<int> retry;
    bool pool_short = false, need_big = false;
    for (int u = 0; u < n_utt; u++) {
      const int e = d->h_res[u].error;
      if (e == ERR_POOL) { retry.push_back(u); pool_short = true; }
      else if (is_cap(e) && !tried_big[u]) { retry.push_back(u); need_big = true; }
    }
    if (retry.empty()) break;
    if (getenv("VBH_DEBUG")) {
      int cnt[16] = {0};
      for (int u : retry) cnt[d->h_res[u].error & 15]++;
      fprintf(stderr, "[vbh] round %d: %zu of %d utterances are searched again:", round, retry.size(), n_utt);
      for (int e = 1; e < 9; e++) if (cnt[e]) fprintf(stderr, " %s x%d;", err_name(e), cnt[e]);
      fprintf(stderr, " (frames done of first: %d)\n", d->h_res[retry[0]].num_frames);
    }
    if (pool_short) {
      // counters hold the demand so far, including the reservations that did not fit; the utterances searched again
      // reserve once more, so twice the demand is enough for this round
      unsigned long long used[4];
      VBH_HIP(hipMemcpyAsync(used, p_cnt.p, sizeof(used), hipMemcpyDeviceToHost, st));
      VBH_HIP(hipStreamSynchronize(st));
      const long long want_t = std::max<long long>(2 * (long long)used[0] + 4096, tok_cap_now), want_l = std::max<long long>(2 * (long long)used[1] + 8192, link_cap_now);
      auto grow = [&](auto &buf, long long old_cap, long long new_cap) -> int {
        if (new_cap <= old_cap) return 0;
        typename std::remove_reference<decltype(buf)>::type nb;
        if (nb.alloc((size_t)new_cap)) return -1;
        VBH_HIP(hipMemcpyAsync(nb.p, buf.p, sizeof(*buf.p) * (size_t)old_cap, hipMemcpyDeviceToDevice, st));
        VBH_HIP(hipStreamSynchronize(st));
        std::swap(buf.p, nb.p); std::swap(buf.n, nb.n);
        return 0;
      };
      if (grow(p_tf, tok_cap_now, want_t) || grow(p_ts, tok_cap_now, want_t) || grow(p_tc, tok_cap_now, want_t) || grow(p_te, tok_cap_now, want_t) ||
          grow(p_tfin, tok_cap_now, want_t)) return -1;
      if (grow(p_ls, link_cap_now, want_l) || grow(p_ld, link_cap_now, want_l) || grow(p_li, link_cap_now, want_l) || grow(p_lo, link_cap_now, want_l) ||
          grow(p_lg, link_cap_now, want_l) || grow(p_la, link_cap_now, want_l)) return -1;
      // the per-frame cost offsets are reserved per attempt as well
      {
        const long long want_c = 2 * (long long)used[2] + 1;
        if (want_c > d->pool_cost_cap) { if (grow(d->p_co, d->pool_cost_cap, want_c)) return -1; d->pool_cost_cap = want_c; }
        d->h_pool.cost_cap = d->pool_cost_cap; d->h_pool.cost_offsets = d->p_co.p;
      }
      tok_cap_now = std::max(tok_cap_now, want_t); link_cap_now = std::max(link_cap_now, want_l);
      d->pool_tok_cap = std::max(d->pool_tok_cap, tok_cap_now); d->pool_link_cap = std::max(d->pool_link_cap, link_cap_now);
      Pool &pool = d->h_pool;
      pool.tok_cap = tok_cap_now; pool.link_cap = link_cap_now;
      pool.tok_frame = p_tf.p; pool.tok_state = p_ts.p; pool.tok_cost = p_tc.p; pool.tok_extra = p_te.p; pool.tok_final = p_tfin.p;
      pool.link_src = p_ls.p; pool.link_dst = p_ld.p; pool.link_il = p_li.p; pool.link_ol = p_lo.p; pool.link_gr = p_lg.p; pool.link_ac = p_la.p;
      VBH_HIP(hipMemcpyAsync(d_pool.p, &pool, sizeof(Pool), hipMemcpyHostToDevice, st));
      VBH_HIP(hipStreamSynchronize(st));
      d->n_pool_grown++;
    }
    // capacity overflows go to the big slots; utterances that only lacked pool space run where they ran before
    for (int pass = 0; pass < 2; pass++) {
      std::vector<int> list;
      for (int u : retry) {
        const bool cap_err = is_cap(d->h_res[u].error);
        if ((pass == 0) == (cap_err || tried_big[u])) list.push_back(u);
      }
      if (list.empty()) continue;
      const bool big = pass == 0;
      if (big && d->big_set.n == 0) {
        if (alloc_slot_set(d, &d->big_set, std::min<int>(16, (int)list.size()), /*big=*/true)) return -1;
      }
      SlotSet &set = big ? d->big_set : *d->small_set;
      DevBuf<int> d_list;
      if (d_list.upload(list.data(), list.size(), st)) return -1;
      VBH_HIP(hipMemsetAsync(d_counter.p, 0, sizeof(int), st));
      const int grid = std::min<int>(big ? set.n : std::min(set.n, d->lay->wg_per_cu * d->ctx->num_cus), (int)list.size());
      VBH_HIP(hipEventRecord(ev0, st));
      VBH_LAUNCH_DECODE(grid, dg, dc, set.d.p, set.dev(), d_loglikes, (int)num_pdfs,
                         d_utts.p, d_list.p, (int)list.size(), d_counter.p, d_res.p, d_pool.p);
      VBH_HIP(hipGetLastError());
      VBH_HIP(hipEventRecord(ev1, st));
      VBH_HIP(hipStreamSynchronize(st));
      float ms = 0; VBH_HIP(hipEventElapsedTime(&ms, ev0, ev1)); d->t_kernel_ms += ms;
      if (big) for (int u : list) { if (!tried_big[u]) d->n_retried++; tried_big[u] = 1; }
    }
    (void)need_big;
    VBH_HIP(hipMemcpyAsync(d->h_res.data(), d_res.p, sizeof(UttResult) * n_utt, hipMemcpyDeviceToHost, st));
    VBH_HIP(hipStreamSynchronize(st));
  }
  unsigned long long used[4];
  VBH_HIP(hipMemcpyAsync(used, p_cnt.p, sizeof(used), hipMemcpyDeviceToHost, st));
  VBH_HIP(hipStreamSynchronize(st));
  for (int u = 0; u < n_utt; u++) {
    UttResult &r = d->h_res[u];
    if (r.error == ERR_NONE) continue;
    d->n_failed++;
    fprintf(stderr, "WARNING (vbh_decode_batch) utterance %d failed: %s (frame %d of %d); it is reported with ok = 0\n", u, err_name(r.error),
            r.num_frames, (int)d->h_utts[u].T);
    if (r.error == ERR_INTERNAL)
      fprintf(stderr, "        check %lld failed: value %lld, bound %lld\n", r.t_phase[8], r.t_phase[10], r.t_phase[11]);
    r.ok = 0; r.reached_final = 0; r.n_tok = 0; r.n_link = 0; r.tok_off = 0; r.link_off = 0;
  }
  const long long tok_cap_final = tok_cap_now, link_cap_final = link_cap_now;
  // bring the used part of the pool to the host
  lap("retries + checks");
  const auto t_copy0 = std::chrono::steady_clock::now();
  const long long nt = std::min<long long>((long long)used[0], tok_cap_final), nl = std::min<long long>((long long)used[1], link_cap_final);
  // pinned staging: [5 token arrays of nt][6 link arrays of nl], 4 bytes per entry
  const size_t need = (size_t)(5 * nt + 6 * nl) * 4 + 64;
  if (need > d->pin_bytes) {
    if (d->pin) (void)hipHostFree(d->pin);
    d->pin = nullptr; d->pin_bytes = 0;
    const size_t want = need + need / 4;
    VBH_HIP(hipHostMalloc(&d->pin, want, hipHostMallocDefault));
    d->pin_bytes = want;
  }
  int32_t *hb = static_cast<int32_t *>(d->pin);
  int *tf = hb, *ts = hb + nt; float *tc = reinterpret_cast<float *>(hb + 2 * nt), *te = reinterpret_cast<float *>(hb + 3 * nt), *tfin = reinterpret_cast<float *>(hb + 4 * nt);
  int32_t *lbase = hb + 5 * nt;
  int *ls = lbase, *ld = lbase + nl, *li = lbase + 2 * nl, *lo = lbase + 3 * nl;
  float *lg = reinterpret_cast<float *>(lbase + 4 * nl), *la = reinterpret_cast<float *>(lbase + 5 * nl);
#define D2H(dst, src, cnt) if (cnt) VBH_HIP(hipMemcpyAsync(dst, src.p, 4 * (size_t)(cnt), hipMemcpyDeviceToHost, st));
  D2H(tf, p_tf, nt) D2H(ts, p_ts, nt) D2H(tc, p_tc, nt) D2H(te, p_te, nt) D2H(tfin, p_tfin, nt)
  D2H(ls, p_ls, nl) D2H(ld, p_ld, nl) D2H(li, p_li, nl) D2H(lo, p_lo, nl) D2H(lg, p_lg, nl) D2H(la, p_la, nl)
#undef D2H
  VBH_HIP(hipStreamSynchronize(st));
  Pool hp{};
  hp.tok_frame = tf; hp.tok_state = ts; hp.tok_cost = tc; hp.tok_extra = te; hp.tok_final = tfin;
  hp.link_src = ls; hp.link_dst = ld; hp.link_il = li; hp.link_ol = lo; hp.link_gr = lg; hp.link_ac = la;
  lap("pool D2H");
  const auto t_host0 = std::chrono::steady_clock::now();
  d->results.resize(n_utt);
  {
    // canonical ordering + best path per utterance: independent, spread over host threads
    int nthr = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
    if (const char *e = getenv("VBH_HOST_THREADS")) { const int x = atoi(e); if (x > 0) nthr = x; }
    nthr = std::min(nthr, n_utt);
    std::atomic<int> next{0};
    auto work = [&]() {
      for (int u = next.fetch_add(1); u < n_utt; u = next.fetch_add(1)) finish_lattice(g, d->h_res[u], hp, &d->results[u]);
    };
    if (nthr <= 1) work();
    else {
      std::vector<std::thread> pool_threads;
      for (int i = 0; i < nthr; i++) pool_threads.emplace_back(work);
      for (auto &t : pool_threads) t.join();
    }
  }
  lap("host lattice finishing");
  const auto t_end = std::chrono::steady_clock::now();
  d->t_copy_ms = std::chrono::duration<float, std::milli>(t_host0 - t_copy0).count();
  d->t_host_ms = std::chrono::duration<float, std::milli>(t_end - t_host0).count();
  d->t_total_ms = std::chrono::duration<float, std::milli>(t_end - t_begin).count();
  return 0;
}

int vbh_decode_batch(vbh_decoder *d, const float *d_loglikes, int32_t num_pdfs, const int64_t *h_frame_offsets,
                     int32_t n_utt, const int32_t *h_tid2pdf, int32_t num_tids) {
  if (vbh_decode_batch_begin(d, d_loglikes, num_pdfs, h_frame_offsets, n_utt, h_tid2pdf, num_tids)) return -1;
  return vbh_decode_batch_end(d);
}

int vbh_decoder_utt_timeline(vbh_decoder *d, int32_t utt, int64_t *out3) {
  VBH_CHECK(d && out3, "vbh_decoder_utt_timeline: NULL argument");
  VBH_CHECK(utt >= 0 && utt < (int)d->h_res.size(), "vbh_decoder_utt_timeline: utterance %d out of range", utt);
  out3[0] = d->h_res[utt].t_start; out3[1] = d->h_res[utt].t_end; out3[2] = d->h_res[utt].wg;
  return 0;
}

int vbh_decoder_diag(vbh_decoder *d, int32_t utt, int64_t *out16) {
  VBH_CHECK(d && out16, "vbh_decoder_diag: NULL argument");
  VBH_CHECK(utt >= 0 && utt < (int)d->h_res.size(), "vbh_decoder_diag: utterance %d out of range", utt);
  for (int i = 0; i < 16; i++) out16[i] = d->h_res[utt].t_phase[i];
  return 0;
}
int vbh_decoder_phase_cycles(vbh_decoder *d, int32_t utt, int64_t *out8) {
  VBH_CHECK(d && out8, "vbh_decoder_phase_cycles: NULL argument");
  VBH_CHECK(utt >= 0 && utt < (int)d->h_res.size(), "vbh_decoder_phase_cycles: utterance %d out of range", utt);
  for (int i = 0; i < 8; i++) out8[i] = d->h_res[utt].t_phase[i];
  return 0;
}

int vbh_decoder_times(vbh_decoder *d, float *out4) {
  VBH_CHECK(d && out4, "vbh_decoder_times: NULL argument");
  out4[0] = d->t_kernel_ms; out4[1] = d->t_copy_ms; out4[2] = d->t_host_ms; out4[3] = d->t_total_ms;
  return 0;
}

int vbh_decoder_set_cfg(vbh_decoder *d, const vbh_decode_cfg *cfg) {
  VBH_CHECK(d && cfg, "vbh_decoder_set_cfg: NULL argument");
  VBH_CHECK(!d->pend.active, "vbh_decoder_set_cfg: a batch is in flight");
  VBH_CHECK(cfg->beam > 0.0f && cfg->max_active > 1 && cfg->lattice_beam > 0.0f && cfg->prune_interval > 0 && cfg->beam_delta > 0.0f && cfg->min_active >= 0,
            "vbh_decoder_set_cfg: invalid config (LatticeFasterDecoderConfig::Check)");
  const SlotSet *o = d->small_set.get();
  VBH_CHECK(o && cfg->max_active <= o->cfg.max_active && effective_prune_interval(cfg) == effective_prune_interval(&o->cfg),
            "vbh_decoder_set_cfg: the search workspaces were made for max_active %d / pruning interval %d: a configuration that needs larger ones "
            "needs a new decoder", o ? o->cfg.max_active : 0, o ? effective_prune_interval(&o->cfg) : 0);
  d->cfg = *cfg;
  return 0;
}

int vbh_decoder_batch_stats(vbh_decoder *d, int32_t *out3) {
  VBH_CHECK(d && out3, "vbh_decoder_batch_stats: NULL argument");
  out3[0] = d->n_retried; out3[1] = d->n_pool_grown; out3[2] = d->n_failed;
  return 0;
}

int vbh_decoder_result(vbh_decoder *d, int32_t utt, vbh_lattice *out) {
  VBH_CHECK(d && out, "vbh_decoder_result: NULL argument");
  VBH_CHECK(utt >= 0 && utt < (int)d->results.size(), "vbh_decoder_result: utterance %d out of range", utt);
  *out = d->results[utt].view;
  return 0;
}

int vbh_decode_layout_for(const vbh_decode_cfg *cfg) { return cfg ? layout_for(cfg) : 1; }
int vbh_decoder_layout(const vbh_decoder *d) { return d && d->lay ? d->lay->id : 0; }

}  // extern "C"
#endif  // !VBH_LAYOUT2
