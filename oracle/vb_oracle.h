/*
 * vb_oracle.h -- CPU restatement (plain C99) of the VoiceBridge/Kaldi hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke test in
 * __graft_entry__.py and the cpu_baseline leg of bench.py may load it.  The product
 * path (voicebridge_amd/, include/voicebridge_hip.h) never links or calls it.
 *
 * Parity status: PINNED.  Every function below is checked (tests/test_oracle_*.py)
 * against golden vectors produced in the build container by the reference's own C++
 * (oracle/_ref harness compiled from /root/reference, see oracle/Makefile and
 * oracle/gen_golden.py) and against the reference's HTK fixtures
 * (kaldi-master/src/feat/test_data/test.wav.fea_htk.*, copied as data under tests/golden/).
 *
 * Each function cites the reference file:line whose arithmetic it restates.  Paths are
 * relative to /root/reference/kaldi-master/src/ unless they start with VB/.
 */
#ifndef VB_ORACLE_H_
#define VB_ORACLE_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- options: feat/feature-window.h:33-101, feat/mel-computations.h:43-74,
 *      feat/feature-mfcc.h:38-78 (same field meaning and defaults) ---- */
enum { VBO_WIN_HANNING = 0, VBO_WIN_HAMMING = 1, VBO_WIN_POVEY = 2,
       VBO_WIN_RECTANGULAR = 3, VBO_WIN_BLACKMAN = 4 };

typedef struct vbo_mfcc_opts {
  float samp_freq;        /* 16000 */
  float frame_shift_ms;   /* 10 */
  float frame_length_ms;  /* 25 */
  float dither;           /* reference default 1.0; parity requires 0 (libc rand) */
  float preemph_coeff;    /* 0.97 */
  int32_t remove_dc_offset;       /* 1 */
  int32_t window_type;            /* VBO_WIN_POVEY */
  int32_t round_to_power_of_two;  /* 1 */
  float blackman_coeff;   /* 0.42 */
  int32_t snip_edges;     /* 1 */
  /* mel */
  int32_t num_bins;       /* 23 */
  float low_freq;         /* 20 */
  float high_freq;        /* 0 (<=0: offset from nyquist) */
  float vtln_low;         /* 100 */
  float vtln_high;        /* -500 */
  int32_t htk_mode;       /* 0 */
  /* mfcc */
  int32_t num_ceps;       /* 13 */
  int32_t use_energy;     /* 1 */
  float energy_floor;     /* 0 */
  int32_t raw_energy;     /* 1 */
  float cepstral_lifter;  /* 22 */
  int32_t htk_compat;     /* 0 */
} vbo_mfcc_opts;

void vbo_mfcc_opts_default(vbo_mfcc_opts *o);

/* feat/feature-window.cc:30-87 */
int32_t vbo_window_shift(const vbo_mfcc_opts *o);
int32_t vbo_window_size(const vbo_mfcc_opts *o);
int32_t vbo_padded_window_size(const vbo_mfcc_opts *o);
int64_t vbo_first_sample_of_frame(int32_t frame, const vbo_mfcc_opts *o);
int32_t vbo_num_frames(int64_t num_samples, const vbo_mfcc_opts *o);

/* Tables (sizes: window[WindowSize], lifter[num_ceps], dct[num_ceps*num_bins],
 * mel: first[num_bins], len[num_bins], weights[num_bins * (padded/2)] zero padded rows).
 * feat/feature-window.cc:109-131, feat/mel-computations.cc:33-144,255-261,
 * matrix/matrix-functions.cc:592-608 */
void vbo_window_function(const vbo_mfcc_opts *o, float *window);
void vbo_lifter_coeffs(float Q, int32_t n, float *coeffs);
void vbo_dct_matrix(int32_t K, int32_t N, float *M /* K x N row-major */);
int vbo_mel_banks(const vbo_mfcc_opts *o, float vtln_warp, int32_t *first,
                  int32_t *len, float *weights /* num_bins x (padded/2) */);

/* Whole utterance MFCC: feat/feature-common-inl.h:75-98 + feature-window.cc:133-220 +
 * feature-mfcc.cc:28-80.  wave is float in int16 range.  out is [T x num_ceps].
 * Optional stage dumps (may be NULL): win [T x padded] (after ProcessWindow, zero
 * padded), pspec [T x (padded/2+1)], mel [T x num_bins] (log mel energies). */
int vbo_mfcc(const vbo_mfcc_opts *o, const float *wave, int64_t n, float vtln_warp,
             float *out, float *win, float *pspec, float *mel);

/* feat/feature-functions.cc:54-111,160-171 */
void vbo_delta_scales(int32_t order, int32_t window, float *scales /* (order+1) x (2*order*window+1) */,
                      int32_t *dims);
void vbo_add_deltas(int32_t order, int32_t window, const float *in, int32_t T,
                    int32_t D, float *out /* T x D*(order+1) */);
/* feat/feature-functions.cc:205-226 */
void vbo_splice(const float *in, int32_t T, int32_t D, int32_t left, int32_t right,
                float *out /* T x D*(left+right+1) */);
/* transform/cmvn.cc:64-113; stats is [2 x (D+1)] double row-major; returns -1 on error */
int vbo_apply_cmvn(const double *stats, int32_t norm_vars, float *feats, int32_t T,
                   int32_t D);
/* transform/cmvn.cc:30-62 (AccCmvnStats over all frames, weight 1) */
void vbo_acc_cmvn_stats(const float *feats, int32_t T, int32_t D, double *stats);
/* VB/src/featbin/transform-feats.cpp:89-107: xform [R x C], C==D (linear) or D+1 (affine) */
int vbo_transform_feats(const float *xform, int32_t R, int32_t C, const float *in,
                        int32_t T, int32_t D, float *out /* T x R */);

/* matrix/compressed-matrix.cc:60-187,347-521 (kSpeechFeature format, "CM"):
 * layout: header{min f32, range f32, rows i32, cols i32} (16 bytes as on disk),
 * cols x {p0,p25,p75,p100 u16}, then column-major u8[rows] per column. */
size_t vbo_compress_size(int32_t rows, int32_t cols);
int vbo_compress(const float *mat, int32_t rows, int32_t cols, uint8_t *blob);
int vbo_uncompress(const uint8_t *blob, float *mat /* rows x cols */);

/* ---- diagonal GMM: gmm/diag-gmm.cc:114-152 (gconsts), :528-543 (loglikes),
 *      gmm/decodable-am-diag-gmm.cc:28-72, matrix/kaldi-vector.cc:757-775 ---- */
int vbo_gmm_gconsts(int32_t G, int32_t D, const float *weights, const float *means_invvars,
                    const float *inv_vars, float *gconsts);
/* all-pdf log-likelihoods, out [T x P]; pdf p owns Gaussians offs[p]..offs[p+1]-1 */
void vbo_gmm_loglikes(int32_t P, const int32_t *offs, int32_t D, const float *gconsts,
                      const float *means_invvars, const float *inv_vars,
                      const float *feats, int32_t T, float *out);
/* per-Gaussian loglikes for one frame (diag-gmm.cc:528-543) */
void vbo_gmm_component_loglikes(int32_t G, int32_t D, const float *gconsts,
                                const float *means_invvars, const float *inv_vars,
                                const float *x, float *out);
float vbo_log_sum_exp(const float *v, int32_t n, float prune);

/* ---- decoding graph: OFST/include/fst/const-fst.h:102-110, arc.h:27-36 ---- */
typedef struct vbo_arc { int32_t ilabel, olabel; float weight; int32_t nextstate; } vbo_arc;
typedef struct vbo_fst {
  int32_t num_states, start;
  const float *final;        /* [S] final cost, +inf if not final */
  const int64_t *arc_offset; /* [S+1] */
  const vbo_arc *arcs;       /* [A] */
} vbo_fst;

/* decoder/lattice-faster-decoder.h:41-92 */
typedef struct vbo_lat_cfg {
  float beam; int32_t max_active; int32_t min_active; float lattice_beam;
  int32_t prune_interval; float beam_delta; float hash_ratio; float prune_scale;
  int32_t exact_cutoff; /* 0 = reference visiting order (online next_cutoff tightening,
                           HashList order); 1 = order-independent variant: next_cutoff is
                           the exact frame minimum + adaptive_beam before any token is
                           admitted (round 1-3's GPU search);
                           2 = the bound is the reference's INITIAL next_cutoff (from the best token's
                           arcs, lattice-faster-decoder.cc:712-724) for every arc: a superset of anything the
                           reference admits in any visiting order;
                           3 = ORDERED admission in a canonical order (the GPU search from round 4 on): the
                           reference's rule -- an arc is admitted iff tot <= min(initial next_cutoff,
                           tot' + adaptive_beam of every arc visited earlier), :747-760 -- with "earlier" =
                           "leaves a token whose state falls into a lower bucket", bucket(state) =
                           (state * 2654435761 mod 2^32) >> (32 - adm_bucket_bits): a prefix minimum over
                           buckets, the same whatever order the tokens of a frame are stored or visited in.
                           Tokens of one bucket do not tighten the bound for each other */
  int32_t adm_bucket_bits; /* exact_cutoff == 3: log2 of the number of buckets (1 .. 20; default 6: one bucket per
                           lane of a wavefront) */
} vbo_lat_cfg;
void vbo_lat_cfg_default(vbo_lat_cfg *c);

/* Output of one lattice decode.  All arrays malloc'ed by the callee, released by
 * vbo_lat_result_free.  Tokens are those alive after FinalizeDecoding
 * (lattice-faster-decoder.cc:600-615), sorted per frame by HCLG state id. */
typedef struct vbo_lat_result {
  int32_t ok;                 /* Decode() return value */
  int32_t reached_final;
  int32_t num_frames;         /* NumFramesDecoded */
  float final_relative_cost, final_best_cost;
  /* surviving tokens */
  int32_t num_toks;
  int32_t *tok_frame;         /* frame_plus_one index 0..num_frames */
  int32_t *tok_state;
  float *tok_cost, *tok_extra;
  float *tok_final;           /* final cost recorded for last-frame tokens (inf if none) */
  /* surviving forward links (raw lattice arcs), sorted */
  int32_t num_links;
  int32_t *link_src, *link_dst;   /* token indices */
  int32_t *link_ilabel, *link_olabel;
  float *link_graph, *link_acoustic; /* acoustic = stored ac_cost - cost_offset[frame] for emitting */
  /* best path (GetBestPath + GetLinearSymbolSequence) */
  int32_t num_ali; int32_t *ali;
  int32_t num_words; int32_t *words;
  float best_graph, best_acoustic;
  /* diagnostics */
  int64_t n_tok_created, n_links_created;
  int32_t n_frames_maxactive_bound, n_frames_minactive_bound;
  float *cost_offsets;        /* [num_frames] */
} vbo_lat_result;
void vbo_lat_result_free(vbo_lat_result *r);

/* decoder/lattice-faster-decoder.cc:56-96 (Decode), with loglikes[t*P + pdf] already
 * acoustic-scaled?  NO: loglikes are raw pdf log-likelihoods [T x P]; the decodable
 * (gmm/decodable-am-diag-gmm.h:142-145) returns acwt * loglikes[t][tid2pdf[tid]]. */
int vbo_decode_lattice_faster(const vbo_fst *fst, const vbo_lat_cfg *cfg,
                              const float *loglikes, int32_t T, int32_t P,
                              const int32_t *tid2pdf /* [num_tids+1], index 0 unused */,
                              int32_t num_tids, float acwt, vbo_lat_result *res);

/* decoder/faster-decoder.{h,cc} */
typedef struct vbo_faster_cfg {
  float beam; int32_t max_active; int32_t min_active; float beam_delta; float hash_ratio;
  /* 0: replay the reference's HashList order (bit-exact with it); 1: the order-independent
   * semantics a parallel search implements: next_cutoff takes its final value before anything is
   * admitted; per state the cheapest incoming arc wins, ties go to the smaller arc index; the
   * epsilon closure is taken to its fixed point before tokens are made; ties between final tokens
   * go to the smaller state id; 3: the same, but arcs are admitted by the reference's own rule (faster-decoder.cc:266-273: strictly
   * below the running next_weight_cutoff, which every admitted arc tightens) applied in a canonical order -- arcs leaving tokens of
   * a lower one of 2^adm_bucket_bits state-hash buckets first, starting from the bound of the best token's arcs (:241-255) -- the
   * lattice search's mode 3 for this decoder (the GPU's VBH_ADMISSION_ORDERED) */
  int32_t exact_cutoff;
  int32_t adm_bucket_bits;
} vbo_faster_cfg;
void vbo_faster_cfg_default(vbo_faster_cfg *c);
typedef struct vbo_faster_result {
  int32_t ok, reached_final, num_frames;
  int32_t num_ali; int32_t *ali;
  int32_t num_words; int32_t *words;
  float best_graph, best_acoustic;
} vbo_faster_result;
void vbo_faster_result_free(vbo_faster_result *r);
int vbo_decode_faster(const vbo_fst *fst, const vbo_faster_cfg *cfg, const float *loglikes,
                      int32_t T, int32_t P, const int32_t *tid2pdf, int32_t num_tids,
                      float acwt, vbo_faster_result *res);

#ifdef __cplusplus
}
#endif
#endif  /* VB_ORACLE_H_ */
