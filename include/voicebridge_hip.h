/*
 * voicebridge_hip.h -- C ABI of the MI355X (gfx950) implementation of VoiceBridge's
 * acoustic-scoring + decoding hot path.  Plain C, caller-allocated buffers, int status
 * (0 = ok, -1 = error, text via vbh_last_error()), no exceptions cross the boundary, no
 * torch/ATen types.  One vbh_ctx per GPU and host thread; entry points taking the same ctx
 * must not be called concurrently.
 *
 * "d_" parameters are DEVICE pointers (hipMalloc'ed memory of the ctx's device, e.g. from
 * vbh_dev_alloc or a torch tensor's data_ptr()); "h_" parameters are host pointers.  All
 * device work is enqueued on the ctx's stream (vbh_ctx_stream); entry points return after
 * enqueueing unless documented otherwise; vbh_ctx_sync waits.
 *
 * Each entry point names the reference interface it replaces (paths relative to
 * /root/reference; K/ = kaldi-master/src/, VB/ = VoiceBridge/VoiceBridge/kaldi-win/).
 * The argv-style bin functions and the step API of the reference (VB/src/kaldi_src.h:17-101,
 * VB/scr/kaldi_scr.h:71-165) are re-declared in voicebridge_amd/host/voicebridge_api.h and
 * are implemented on top of this ABI; INTEGRATION.md shows the binding a maintainer adds.
 */
#ifndef VOICEBRIDGE_HIP_H_
#define VOICEBRIDGE_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (r03): vbh_best_path has six more fields than in version 1 and vbh_mfcc_opts.dither is applied (version 1 required 0);
 * 3 (r04): vbh_decode_cfg grew by `admission` (0 = the default);
 * 4 (r05): vbh_faster_cfg grew by `admission` (0 = the default) */
#define VBH_ABI_VERSION 4

/* ------------------------------------------------------------------ context / memory */
typedef struct vbh_ctx vbh_ctx;

int vbh_abi_version(void);
const char *vbh_last_error(void);                 /* thread-local message of the last failure */
int vbh_device_count(void);                       /* 0 when no HIP device is visible */
int vbh_ctx_create(int device_id, vbh_ctx **out); /* fails (-1) when there is no such GPU */
/* priority 0 = default; larger = this context's kernels are dispatched before those of lower
 * priority contexts when CUs free up (clamped to what the device offers) */
int vbh_ctx_create_with_priority(int device_id, int priority, vbh_ctx **out);
/* a context whose kernels run only on CUs [first_cu, first_cu + n_cus) of the device's CU mask
 * (consecutive mask bits are spread evenly over the XCDs); kernels that size their grid by the
 * CU count (the searches) see n_cus.  Used to keep CUs free for the work of other contexts while
 * a persistent search grid is resident. */
int vbh_ctx_create_on_cus(int device_id, int first_cu, int n_cus, vbh_ctx **out);
void vbh_ctx_destroy(vbh_ctx *ctx);
int vbh_ctx_sync(vbh_ctx *ctx);
/* work enqueued on ctx after this call starts only when everything enqueued on other so far
 * has completed (stages of a pipeline on different contexts of one device); does not block */
int vbh_ctx_wait_for(vbh_ctx *ctx, vbh_ctx *other);
/* finer than vbh_ctx_wait_for: vbh_ctx_mark records "everything enqueued on ctx so far" under a slot number (0..15; a slot is
 * overwritten by its next mark), vbh_ctx_wait_mark makes ctx wait for that point of other's stream -- a consumer stream
 * waits for its own item of a producer stream that is already working on later items */
int vbh_ctx_mark(vbh_ctx *ctx, int slot);
int vbh_ctx_wait_mark(vbh_ctx *ctx, vbh_ctx *other, int slot);
void *vbh_ctx_stream(vbh_ctx *ctx);               /* hipStream_t all work is enqueued on */
int vbh_ctx_device(vbh_ctx *ctx);
/* Scratch device memory of a context: slot's buffer (slots 0 .. 11), at least `bytes` long, grown when a call asks for more and otherwise
 * handed out again -- for a host's per-call work buffers.  No counterpart in the reference (its programs allocate on the host heap); on the
 * device a free waits for every stream of the device, which a host that runs several contexts side by side (the step API's job threads)
 * cannot afford per call.  The buffer belongs to the context: contents do not survive growth, the memory goes with vbh_ctx_destroy. */
int vbh_ctx_scratch(vbh_ctx *ctx, int32_t slot, size_t bytes, void **d_ptr);
/* gives the slot's memory back now (slot < 0: every slot's); the next vbh_ctx_scratch allocates again.  Waits for the context's stream. */
int vbh_ctx_scratch_release(vbh_ctx *ctx, int32_t slot);
int vbh_dev_alloc(vbh_ctx *ctx, size_t bytes, void **d_ptr);
int vbh_dev_free(vbh_ctx *ctx, void *d_ptr);
int vbh_copy_h2d(vbh_ctx *ctx, void *d_dst, const void *h_src, size_t bytes); /* async on the stream */
int vbh_copy_d2h(vbh_ctx *ctx, void *h_dst, const void *d_src, size_t bytes); /* synchronises */
int vbh_memset(vbh_ctx *ctx, void *d_dst, int value, size_t bytes);
/* HIP-event stopwatch on the ctx stream (used by bench.py for per-kernel device time) */
int vbh_timer_start(vbh_ctx *ctx, int slot);
int vbh_timer_stop(vbh_ctx *ctx, int slot);       /* records the stop event */
int vbh_timer_elapsed_ms(vbh_ctx *ctx, int slot, float *ms); /* waits for the stop event */

/* ------------------------------------------------------------------ MFCC
 * replaces Mfcc::ComputeFeatures / OfflineFeatureTpl<MfccComputer>::Compute
 * (K/feat/feature-common-inl.h:29-98) as driven by ComputeMFCCFeats
 * (VB/src/featbin/compute-mfcc-feats.cpp:103-147).  Field meaning and defaults are those of
 * FrameExtractionOptions (K/feat/feature-window.h:33-101), MelBanksOptions
 * (K/feat/mel-computations.h:43-74) and MfccOptions (K/feat/feature-mfcc.h:38-78). */
enum { VBH_WIN_HANNING = 0, VBH_WIN_HAMMING = 1, VBH_WIN_POVEY = 2, VBH_WIN_RECTANGULAR = 3,
       VBH_WIN_BLACKMAN = 4 };

typedef struct vbh_mfcc_opts {
  float samp_freq, frame_shift_ms, frame_length_ms;
  float dither;               /* reference default 1.0; here drawn from a counter-based generator, see vbh_mfcc_set_dither_stream */
  float preemph_coeff;
  int32_t remove_dc_offset, window_type, round_to_power_of_two;
  float blackman_coeff;
  int32_t snip_edges;
  int32_t num_bins;
  float low_freq, high_freq, vtln_low, vtln_high;
  int32_t htk_mode;
  int32_t num_ceps, use_energy;
  float energy_floor;
  int32_t raw_energy;
  float cepstral_lifter;
  int32_t htk_compat;
} vbh_mfcc_opts;

void vbh_mfcc_opts_default(vbh_mfcc_opts *o);                    /* reference defaults except dither = 0 (bit parity needs it) */
int32_t vbh_mfcc_num_frames(const vbh_mfcc_opts *o, int64_t num_samples); /* K/feat/feature-window.cc:41-87 */

typedef struct vbh_mfcc vbh_mfcc;
/* builds window / mel filterbank / DCT / lifter tables (K/feat/feature-window.cc:109-131,
 * mel-computations.cc:33-144, matrix-functions.cc:592-608) and uploads them */
int vbh_mfcc_create(vbh_ctx *ctx, const vbh_mfcc_opts *opts, float vtln_warp, vbh_mfcc **out);
void vbh_mfcc_destroy(vbh_mfcc *m);
/* Batch of n_utt utterances.  d_wave holds the samples of all utterances back to back
 * (int16 PCM when wave_is_int16, else float in int16 range as WaveData keeps them,
 * K/feat/wave-reader.cc:260-310); utterance u is samples [h_wave_offsets[u], h_wave_offsets[u+1]).
 * h_frame_offsets (n_utt+1, written by the call) gives the row range of each utterance in
 * d_feats ([total_frames x num_ceps] row-major float32, caller sized with
 * vbh_mfcc_num_frames). */
int vbh_mfcc_compute(vbh_mfcc *m, const void *d_wave, int wave_is_int16,
                     const int64_t *h_wave_offsets, int32_t n_utt, float *d_feats,
                     int64_t *h_frame_offsets);
/* Dither (Dither(), K/feat/feature-window.cc:90-98: dither * N(0,1) added to every sample of the extracted window).  The
 * reference draws from libc rand(), a sequential stream no parallel code can replay; here the normal of (utterance, frame,
 * window sample) comes from a counter-based generator (Philox2x32-10 + Box-Muller): same distribution, NOT the same bits --
 * parity tests use dither = 0.  The noise is a pure function of (seed, first_utt_index + position of the utterance in the
 * call, frame, sample), so a job gets the same features however it is cut into calls.  Defaults: a fixed seed, index 0. */
int vbh_mfcc_set_dither_stream(vbh_mfcc *m, uint64_t seed, int64_t first_utt_index);
/* ... or, for the NEXT vbh_mfcc_compute call only, an explicit stream index per utterance (e.g. a hash of the utterance's name: the noise
 * then does not depend on which job the utterance is in, nor on the order or number of the jobs) */
int vbh_mfcc_set_dither_ids(vbh_mfcc *m, const int64_t *h_ids, int32_t n);

/* ------------------------------------------------------------------ feature post-processing
 * replaces the apply-cmvn | add-deltas | [splice-feats | transform-feats final.mat] |
 * [transform-feats trans.JOB] pipeline that Decode() builds
 * (VB/scr/steps/decode_gmm.cpp:201-293): ApplyCmvn (K/transform/cmvn.cc:64-113),
 * ComputeDeltas (K/feat/feature-functions.cc:160-171), SpliceFrames (:205-226) and the
 * body of TransformFeats (VB/src/featbin/transform-feats.cpp:89-107), fused in one pass. */
typedef struct vbh_featpost_cfg {
  int32_t in_dim;            /* 13 */
  int32_t apply_cmvn;        /* 0/1 */
  int32_t norm_vars;         /* cmvn --norm-vars */
  int32_t mode;              /* 0 = none, 1 = deltas, 2 = splice */
  int32_t delta_order, delta_window;   /* 2, 2 */
  int32_t splice_left, splice_right;   /* 4, 4 */
  int32_t lda_rows, lda_cols;          /* global transform after deltas/splice; 0 = none.
                                          cols == dim (linear) or dim+1 (affine) */
  int32_t spk_rows, spk_cols;          /* per-speaker (fMLLR) transform; 0 = none */
} vbh_featpost_cfg;

int32_t vbh_featpost_out_dim(const vbh_featpost_cfg *cfg);
/* h_cmvn_stats: [n_spk][2][in_dim+1] double (K/transform/cmvn.cc:25-47); h_lda: [lda_rows][lda_cols];
 * h_spk_xform: [n_spk][spk_rows][spk_cols]; h_utt2spk: [n_utt] speaker index of each utterance.
 * d_in [total_frames x in_dim], d_out [total_frames x out_dim]. */
int vbh_featpost(vbh_ctx *ctx, const vbh_featpost_cfg *cfg, const float *d_in,
                 const int64_t *h_frame_offsets, int32_t n_utt, const int32_t *h_utt2spk,
                 int32_t n_spk, const double *h_cmvn_stats, const float *h_lda,
                 const float *h_spk_xform, float *d_out);
/* CMVN statistics per speaker (ComputeCmvnStats, K/transform/cmvn.cc:30-62): h_stats_out
 * [n_spk][2][dim+1] double, accumulated on the device, copied back (synchronises). */
int vbh_cmvn_stats(vbh_ctx *ctx, const float *d_feats, int32_t dim, const int64_t *h_frame_offsets,
                   int32_t n_utt, const int32_t *h_utt2spk, int32_t n_spk, double *h_stats_out);

/* ------------------------------------------------------------------ diagonal GMM scoring
 * replaces AmDiagGmm + DecodableAmDiagGmmScaled::LogLikelihood evaluated for every pdf of
 * every frame (K/gmm/decodable-am-diag-gmm.cc:28-72, K/gmm/diag-gmm.cc:528-562,
 * K/matrix/kaldi-vector.cc:757-775).  Model arrays are those of DiagGmm
 * (K/gmm/diag-gmm.h: weights_, means_invvars_, inv_vars_); gconsts are recomputed as
 * DiagGmm::Read does (K/gmm/diag-gmm.cc:114-152, :728-756). */
typedef struct vbh_gmm vbh_gmm;
int vbh_gmm_create(vbh_ctx *ctx, int32_t dim, int32_t num_pdfs, const int32_t *h_pdf_offsets,
                   const float *h_weights, const float *h_means_invvars, const float *h_inv_vars,
                   vbh_gmm **out);
void vbh_gmm_destroy(vbh_gmm *g);
int32_t vbh_gmm_num_pdfs(const vbh_gmm *g);
int32_t vbh_gmm_dim(const vbh_gmm *g);
int vbh_gmm_get_gconsts(const vbh_gmm *g, float *h_gconsts);   /* [G] as computed at create */
/* How the LogSumExp over a pdf's Gaussians is evaluated (K/matrix/kaldi-vector.cc:757-775: the maximum, then the sum of
 * exp(x - max) over the terms not below max + log(FLT_EPSILON), accumulated in double, max + log(sum)):
 *   VBH_LSE_FAST    (default) one pass, running maximum + rescaled f32 sum, one hardware exp per Gaussian, no cutoff;
 *   VBH_LSE_STRICT  the reference's arithmetic: two passes with its cutoff, expf, f64 sum, max + log(sum) in f64, rounded once;
 *   VBH_LSE_TWOPASS the strict mode's two passes and cutoff in f32 with the hardware functions (measurement only: slower).
 * All three are within 1e-6 relative of the reference's gmm-compute-likes on its own features (tests/test_gpu_e2e.py).
 * The environment variable VBH_GMM_LSE=strict|twopass sets the mode of models created afterwards. */
#define VBH_LSE_FAST 0
#define VBH_LSE_STRICT 1
#define VBH_LSE_TWOPASS 2
int vbh_gmm_set_lse_mode(vbh_gmm *g, int32_t mode);
/* d_feats [T x dim] -> d_loglikes [T x num_pdfs] (unscaled log-likelihoods) */
int vbh_gmm_loglikes(vbh_gmm *g, const float *d_feats, int64_t T, float *d_loglikes);
/* The same with rows ld >= num_pdfs floats apart (columns num_pdfs..ld-1 are not written).  With ld = vbh_gmm_padded_pdfs()
 * (num_pdfs rounded up to a multiple of 32: rows start on 128-byte lines) every store of the kernel covers whole 32-byte
 * sectors; the search entry points take such a matrix as it is: their num_pdfs argument is the row length of d_loglikes. */
int vbh_gmm_loglikes_ld(vbh_gmm *g, const float *d_feats, int64_t T, float *d_loglikes, int32_t ld);
int32_t vbh_gmm_padded_pdfs(const vbh_gmm *g);
/* The on-demand form (the reference's semantics, K/gmm/decodable-am-diag-gmm.cc:28-72: a pdf is evaluated when the search or a
 * lattice asks for it): log-likelihoods of the (frame, pdf) pairs [h_pair_offsets[t], h_pair_offsets[t+1]) of every frame t only;
 * h_out [h_pair_offsets[T]] on the host.  Used by gmm-rescore-lattice.  Synchronises. */
int vbh_gmm_loglikes_pairs(vbh_gmm *g, const float *d_feats, int64_t T, const int64_t *h_pair_offsets, const int32_t *h_pair_pdf,
                           float *h_out);
/* algorithmic work of one vbh_gmm_loglikes call: bytes and flops */
int vbh_gmm_work(const vbh_gmm *g, int64_t T, double *bytes, double *flops);

/* ------------------------------------------------------------------ decoding graph + search
 * Graph arrays are those of fst::ConstFst<StdArc> (openfst-win-1.6/src/include/fst/const-fst.h:102-110,
 * arc.h:27-36) as read by ReadFstKaldiGeneric (K/fstext/kaldi-fst-io.cc:45-90). */
typedef struct vbh_arc { int32_t ilabel, olabel; float weight; int32_t nextstate; } vbh_arc;
typedef struct vbh_graph vbh_graph;
int vbh_graph_create(vbh_ctx *ctx, int32_t num_states, int32_t start, const float *h_final,
                     const int64_t *h_arc_offsets, const vbh_arc *h_arcs, vbh_graph **out);
void vbh_graph_destroy(vbh_graph *g);

/* LatticeFasterDecoderConfig (K/decoder/lattice-faster-decoder.h:41-92) */
typedef struct vbh_decode_cfg {
  float beam;            /* 13 */
  int32_t max_active;    /* 7000 */
  int32_t min_active;    /* 200 */
  float lattice_beam;    /* 6 */
  int32_t prune_interval;/* 25 */
  float beam_delta;      /* 0.5 */
  float acoustic_scale;  /* 0.083333 */
  /* ProcessEmitting's admission rule (K/decoder/lattice-faster-decoder.cc:707-760).  The reference admits an arc iff
   * tot <= min(initial next_cutoff from the best token's arcs, tot' + adaptive_beam of every arc it visited earlier), in the order of
   * its token hash list.  VBH_ADMISSION_ORDERED (0, the default): the same prefix minimum over a canonical order -- arcs leaving
   * tokens whose state hashes to a lower one of 64 buckets count as "earlier" -- so the result does not depend on where tokens are
   * stored; statistically the reference's rule.  VBH_ADMISSION_EXACT (1): tot <= the frame's final next_cutoff
   * (the minimum over all arcs), the order-independent subset rounds 1-3 computed.
   * What neither mode is: a replay of the reference's HashList order.  On identical scores ORDERED reproduces the reference's
   * (default --hash-ratio) best path wherever the visiting order does not decide it -- every golden case but the one random graph
   * with max-active binding in every frame, 7 of 8 speech utterances, 491 of 512 end to end (the reference against itself under
   * another --hash-ratio: 505-506) -- and EXACT reproduced 8 of 8 and 495 of 512 on those samples while creating 3 % fewer tokens
   * and running ~5 % faster: a caller that wants speed or per-utterance identity on easy material may prefer EXACT, one that wants
   * the reference's amount of search ORDERED.  The host programs take --admission=ordered|exact. */
  int32_t admission;
} vbh_decode_cfg;
enum { VBH_ADMISSION_ORDERED = 0, VBH_ADMISSION_EXACT = 1 };
void vbh_decode_cfg_default(vbh_decode_cfg *c);

/* Result of decoding one utterance: the lattice-beam-pruned token/arc sets after
 * FinalizeDecoding (K/decoder/lattice-faster-decoder.cc:600-615), i.e. the content of
 * GetRawLattice (:110-192), plus the best path (GetBestPath :100-106 and
 * GetLinearSymbolSequence).  Arrays are owned by the result object. */
typedef struct vbh_lattice {
  int32_t ok, reached_final, num_frames;
  float final_relative_cost;
  int32_t num_toks;
  const int32_t *tok_frame, *tok_state;
  const float *tok_cost, *tok_extra, *tok_final;
  int32_t num_links;
  const int32_t *link_src, *link_dst, *link_ilabel, *link_olabel;
  const float *link_graph, *link_acoustic;
  int32_t num_ali; const int32_t *ali;
  int32_t num_words; const int32_t *words;
  float best_graph, best_acoustic;
  /* per-utterance counters of the run (decoder roofline) */
  int64_t n_tokens_expanded, n_arcs_expanded, n_tokens_created, n_links_created;
  int32_t n_frames_maxactive_bound, n_frames_minactive_bound;
} vbh_lattice;

typedef struct vbh_decoder vbh_decoder;
/* max_frames bounds the longest utterance, max_batch the utterances searched concurrently */
int vbh_decoder_create(vbh_ctx *ctx, const vbh_graph *graph, const vbh_decode_cfg *cfg,
                       int32_t max_batch, int32_t max_frames, vbh_decoder **out);
/* The same, drawing its search workspaces from the pool of another decoder of the same device, graph size and capacity
 * settings (share_slots_with; NULL = a pool of its own): decoders on several contexts/streams of one device then keep
 * several batches in flight out of ONE set of workspaces -- the workgroups of whatever grids are resident claim and return
 * slots -- instead of one set per decoder.  The pool lives as long as any decoder that uses it. */
int vbh_decoder_create_shared(vbh_ctx *ctx, const vbh_graph *graph, const vbh_decode_cfg *cfg,
                              int32_t max_batch, int32_t max_frames, vbh_decoder *share_slots_with, vbh_decoder **out);
/* Another search configuration for the batches that follow (beams, min_active, acoustic scale; max_active up to the value the
 * decoder's workspaces were made for): DecodeFmllr's speaker-independent pass (beam 10, max-active 2000) and its main pass
 * (beam 13, max-active 7000) then share one decoder and one set of workspaces. */
int vbh_decoder_set_cfg(vbh_decoder *d, const vbh_decode_cfg *cfg);
/* The library holds the search kernel in two layouts and a decoder is made for one of them:
 *   1 = one workgroup of 1024 threads per CU (an utterance has the CU to itself): heavy searches, the default;
 *   2 = two workgroups of 512 threads per CU: light searches (max_active <= 4000, e.g. DecodeFmllr's first pass at beam 10 /
 *       max-active 2000), whose frames are a chain of short dependent phases -- two utterances per CU fill each other's waits
 *       (1.4 x the frames per second at those settings; 5 % slower than layout 1 at max-active 7000).
 * Results do not depend on the layout (the same algorithm on a smaller first-level state map; tests/test_gpu_variant.py runs the
 * parity tests in layout 2).  vbh_decode_layout_for() = the layout vbh_decoder_create picks for a configuration (environment
 * VBH_DECODE_LAYOUT=1|2 overrides); vbh_decoder_set_cfg never changes a decoder's layout, so a caller that alternates between a
 * light and a heavy configuration keeps one decoder per layout (voicebridge_amd/host/voicebridge_api.cc does). */
int vbh_decode_layout_for(const vbh_decode_cfg *cfg);
int vbh_decoder_layout(const vbh_decoder *d);
void vbh_decoder_destroy(vbh_decoder *d);
/* Replaces LatticeFasterDecoder::Decode + FinalizeDecoding for a batch of utterances
 * (DecodeUtteranceLatticeFaster, K/decoder/decoder-wrappers.cc:198-293).  d_loglikes
 * [total_frames x num_pdfs] unscaled (num_pdfs is the ROW LENGTH of the matrix: a matrix written by vbh_gmm_loglikes_ld is
 * passed with its ld, the columns past the model's pdfs are never read); h_tid2pdf [num_tids+1] = TransitionModel::id2pdf_id_
 * (K/hmm/transition-model.h:324-328).  Synchronises; results are read with
 * vbh_decoder_result and stay valid until the next decode call. */
int vbh_decode_batch(vbh_decoder *d, const float *d_loglikes, int32_t num_pdfs,
                     const int64_t *h_frame_offsets, int32_t n_utt, const int32_t *h_tid2pdf,
                     int32_t num_tids);
/* The same in two halves, so that a caller can overlap batches on different contexts/streams
 * (the persistent search grid of one batch drains while the next one's fills the freed CUs):
 * _begin enqueues the uploads and the search and returns; _end waits, searches overflowed
 * utterances again in big slots, and brings the lattices to the host.  Buffers passed to _begin
 * (log-likelihoods) must stay valid until _end returns. */
int vbh_decode_batch_begin(vbh_decoder *d, const float *d_loglikes, int32_t num_pdfs, const int64_t *h_frame_offsets,
                           int32_t n_utt, const int32_t *h_tid2pdf, int32_t num_tids);
int vbh_decode_batch_end(vbh_decoder *d);
int vbh_decoder_result(vbh_decoder *d, int32_t utt, vbh_lattice *out);
/* Capacities inside the search are estimates and never decide a result: {utterances of the last batch that outgrew a
 * regular search slot and were searched again in a big one, times the output pool had to be enlarged, utterances given
 * up (reported with ok = 0 and a warning on stderr; the rest of the batch is delivered, as the reference fails one
 * utterance at a time, K/decoder/decoder-wrappers.cc:227-231)} */
int vbh_decoder_batch_stats(vbh_decoder *d, int32_t *out3);
/* diagnostic: shader-clock cycles the utterance's workgroup spent per phase
 * {cutoff, pass1, admit+links, epsilon closure, pruning, compaction, output, other} */
int vbh_decoder_phase_cycles(vbh_decoder *d, int32_t utt, int64_t *out8);
/* diagnostic: the same 8 values, then -- in libraries built with -DVBH_DIAG, zeros otherwise -- {cycles admitting candidates (pass 2;
 * [2] is then the forward links alone), cycles of the cutoff phase up to the first reduction, admissions that went to the HBM level of
 * the state map, admissions, frames that created more than 3/4 of the LDS map's slots, frames with more than two candidates per slot, 0, 0} */
int vbh_decoder_diag(vbh_decoder *d, int32_t utt, int64_t *out16);
/* diagnostic: {device wall clock (100 MHz ticks) when a workgroup took the utterance, when it
 * finished it, workgroup id} -- load balance of the persistent search grid */
int vbh_decoder_utt_timeline(vbh_decoder *d, int32_t utt, int64_t *out3);
/* diagnostic: milliseconds of the last vbh_decode_batch {search kernel(s) on the device,
 * lattice pool device->host, host-side canonical ordering + best path, whole call} */
int vbh_decoder_times(vbh_decoder *d, float *out4);

/* ---- fMLLR sufficient statistics.  Replaces, per batch of utterances,
 * FmllrDiagGmmAccs::AccumulateForGmm / AccumulateFromPosteriors / CommitSingleFrameStats
 * (K/transform/fmllr-diag-gmm.cc:30-44,100-113,562-596) as gmm-est-fmllr drives them
 * (K/gmmbin/gmm-est-fmllr.cc:36-52): frame t of d_feats [T x dim] carries the posterior pairs
 * [h_pair_offsets[t], h_pair_offsets[t+1]) of (pdf-id, weight) -- transition-id posteriors summed
 * per pdf (ConvertPosteriorToPdfs, K/hmm/posterior.cc:322-346).  Component posteriors come from
 * ali_gmm, the linear/quadratic terms from adapt_gmm (the same model twice for gmm-est-fmllr; the
 * alignment model and the adaptation model for the gmm-post-to-gpost | gmm-est-fmllr-gpost pair of
 * DecodeFmllr's first pass); the two models must have the same Gaussians per pdf.  h_utt2spk[u] in
 * [0, n_spk) or -1 (utterance skipped).  Outputs on the host, double: h_beta [n_spk],
 * h_K [n_spk][dim][dim+1], h_G [n_spk][dim][dim+1][dim+1] (full symmetric); h_K and h_G may both be NULL (the statistics stay on the
 * device for vbh_fmllr_update_full).  kernel_ms (optional):
 * HIP-event duration of the accumulation kernel on the context's stream.  Synchronises. */
int vbh_fmllr_accumulate(vbh_ctx *ctx, const vbh_gmm *ali_gmm, const vbh_gmm *adapt_gmm, const float *d_feats, int32_t n_utt,
                         const int64_t *h_frame_offsets, const int32_t *h_utt2spk, int32_t n_spk, const int64_t *h_pair_offsets,
                         const int32_t *h_pair_pdf, const float *h_pair_weight, double *h_beta, double *h_K, double *h_G,
                         float *kernel_ms);
/* The same from explicit Gaussian-level posteriors (gmm-est-fmllr-gpost, K/gmmbin/gmm-est-fmllr-gpost.cc:36-50): pair q carries the
 * posteriors h_gpost[h_gpost_offsets[q] .. h_gpost_offsets[q+1]) of the Gaussians of pdf h_pair_pdf[q] (h_pdf_num_gauss[pdf] of them). */
int vbh_fmllr_accumulate_gpost(vbh_ctx *ctx, const vbh_gmm *adapt_gmm, const float *d_feats, int32_t n_utt, const int64_t *h_frame_offsets,
                               const int32_t *h_utt2spk, int32_t n_spk, const int64_t *h_pair_offsets, const int32_t *h_pair_pdf,
                               const int64_t *h_gpost_offsets, const float *h_gpost, const int32_t *h_pdf_num_gauss, double *h_beta, double *h_K,
                               double *h_G, float *kernel_ms);
/* The row-by-row update on the device (ComputeFmllrMatrixDiagGmmFull / FmllrInnerUpdate / FmllrAuxFuncDiagGmm,
 * K/transform/fmllr-diag-gmm.cc:195-273, 496-508; update type "full") from the statistics the context's LAST vbh_fmllr_accumulate* call
 * left on the device -- pass h_K = h_G = NULL there and the 128 MB of a 256-speaker batch never cross the bus.  One wave per speaker,
 * double throughout.  h_xform [n_spk][dim][dim+1] float: in = the initial transform (the unit transform for gmm-est-fmllr), out = the
 * estimate (unchanged where beta <= min_count or the objective did not increase); h_impr [n_spk] = the objective's improvement (0 there).
 * A singular statistic fails the call ("fMLLR: singular matrix").  Synchronises. */
int vbh_fmllr_update_full(vbh_ctx *ctx, int32_t dim, int32_t n_spk, float min_count, int32_t num_iters, float *h_xform, float *h_impr,
                          float *kernel_ms);
/* algorithmic work of the accumulation kernel for T frames: bytes read, f64 flops issued */
int vbh_fmllr_work(int32_t dim, int64_t T, double *bytes, double *flops);

/* ---- FasterDecoder (K/decoder/faster-decoder.{h,cc}; the search of gmm-decode-faster,
 * VB/src/gmmbin/gmm-decode-faster.cpp): 1-best Viterbi beam search, no lattice.  Same graph and
 * score-matrix conventions as the lattice decoder above. */
typedef struct vbh_faster_cfg {   /* FasterDecoderOptions (faster-decoder.h:31-71) */
  float beam;            /* 16 */
  int32_t max_active;    /* INT32_MAX */
  int32_t min_active;    /* 20 */
  float beam_delta;      /* 0.5 */
  float acoustic_scale;  /* 0.083333 */
  /* ProcessEmitting's admission rule (K/decoder/faster-decoder.cc:241-273), as vbh_decode_cfg::admission: VBH_ADMISSION_ORDERED (0,
   * the default) = the reference's online tightening of next_weight_cutoff applied in a canonical visiting order (64 state-hash
   * buckets); VBH_ADMISSION_EXACT (1) = new_weight < the frame's final next_weight_cutoff, the order-independent subset */
  int32_t admission;
} vbh_faster_cfg;
void vbh_faster_cfg_default(vbh_faster_cfg *cfg);
typedef struct vbh_faster_decoder vbh_faster_decoder;
int vbh_faster_decoder_create(vbh_ctx *ctx, const vbh_graph *graph, const vbh_faster_cfg *cfg, int32_t max_batch,
                              int32_t max_frames, vbh_faster_decoder **out);
/* The same for alignment (AlignUtteranceWrapper, K/decoder/decoder-wrappers.cc:424-516: one small graph per utterance): `graph`
 * is the disjoint union of the utterances' graphs and max_states_per_utt the largest of them, which sizes the per-frame
 * workspaces (0: sized for the whole graph) */
int vbh_faster_decoder_create_ex(vbh_ctx *ctx, const vbh_graph *graph, const vbh_faster_cfg *cfg, int32_t max_batch,
                                 int32_t max_frames, int32_t max_states_per_utt, vbh_faster_decoder **out);
void vbh_faster_decoder_destroy(vbh_faster_decoder *d);
/* decodes n_utt utterances (rows h_frame_offsets[u]..[u+1] of d_loglikes [T x num_pdfs]); blocks */
int vbh_faster_decode_batch(vbh_faster_decoder *d, const float *d_loglikes, int32_t num_pdfs,
                            const int64_t *h_frame_offsets, int32_t n_utt, const int32_t *h_tid2pdf,
                            int32_t num_tids);
/* ... with a start state per utterance (h_start_states, NULL or -1 entries: the graph's own) and, when beam > 0, that beam instead
 * of the configured one (the retry_beam second attempt of AlignUtteranceWrapper, decoder-wrappers.cc:466-473) */
int vbh_faster_decode_batch_ex(vbh_faster_decoder *d, const float *d_loglikes, int32_t num_pdfs,
                               const int64_t *h_frame_offsets, int32_t n_utt, const int32_t *h_tid2pdf,
                               int32_t num_tids, const int32_t *h_start_states, float beam);
typedef struct vbh_best_path {    /* GetBestPath + GetLinearSymbolSequence (faster-decoder.cc:87-147) */
  int32_t ok;                     /* a token survived the last frame (Decode's return value) */
  int32_t reached_final;          /* ReachedFinal() */
  int32_t num_frames;
  int32_t num_ali; const int32_t *ali;       /* transition-ids */
  int32_t num_words; const int32_t *words;
  float best_graph, best_acoustic;           /* LatticeWeight of the path (acoustic still scaled) */
  int64_t n_tokens;                          /* tokens created (diagnostic) */
  /* the path arc by arc, as GetBestPath builds its linear lattice before RemoveEpsLocal (faster-decoder.cc:113-141): labels of the
   * graph arc, graph cost = its weight, acoustic cost = (float)(token cost - previous token's cost) - graph cost (still scaled);
   * final_graph: the final cost of the last state when the path ends in a final state (reached_final), else 0 */
  int32_t num_arcs; const int32_t *arc_ilabel; const int32_t *arc_olabel; const float *arc_graph; const float *arc_acoustic;
  float final_graph;
} vbh_best_path;
/* pointers stay valid until the next vbh_faster_decode_batch / destroy on this decoder */
int vbh_faster_decoder_result(vbh_faster_decoder *d, int32_t utt, vbh_best_path *out);

#ifdef __cplusplus
}
#endif
#endif /* VOICEBRIDGE_HIP_H_ */
