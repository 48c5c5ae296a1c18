// voicebridge_steps.h -- the reference's step API (seam B1 of SURVEY.md §8b) for the hot path,
// re-stated over the bin functions of voicebridge_api.h: directory conventions in, files out.
//
// Mirrors (paths relative to /root/reference/VoiceBridge/VoiceBridge/kaldi-win/):
//   scr/kaldi_scr.h:71-77    VOICEBRIDGE_API int MakeMfcc(fs::path datadir, fs::path mfcc_config, int nj, bool compress,
//                                                          bool write_utt2num_frames)       scr/steps/make_mfcc.cpp:24-327
//   scr/kaldi_scr.h:87-94    VOICEBRIDGE_API int ComputeCmvnStats(fs::path datadir, bool fake, bool two_channel,
//                                                                  std::string fake_dims)    scr/steps/compute_cmvn_stats.cpp:23-199
//   scr/kaldi_scr.h:85       int SplitData(fs::path datadir, int numsplit, bool per_utt)     scr/utils/split_data.cpp:20-160
//   scr/kaldi_scr.h:142-165  VOICEBRIDGE_API int Decode(graph_dir, data_dir, decode_dir, model, trans_dir, wer filters, iter,
//                                                        nj, acwt, stage, max_active, beam, lattice_beam, skip_scoring, ...)
//                                                                                             scr/steps/decode_gmm.cpp:60-390
// Same names, argument order and meaning, return 0 / -1; std::string stands in for boost::filesystem::path.
// The nj jobs of a step are batches for a GPU here, not processes.  By default they run on up to four host threads that share the
// calling thread's GPU (vb_set_device; VB_STEP_JOB_THREADS=1 or a one-entry device list: one after the other on the calling thread);
// vb_set_step_devices (or VB_STEP_DEVICES=0,1,...) names the GPUs of the node to use: the jobs
// are then dealt round-robin to them, one host thread per entry -- the reference's nj concurrent job threads
// (scr/steps/decode_gmm.cpp:300-324), with a device behind each.  A device may be named more than once ("0,0,0,0"): that many
// job threads share it, one job's host work (archives, determinization, fMLLR updates) overlapping the others' kernels, their
// decoders drawing on ONE pool of search workspaces.  The job threads persist (and keep their model / graph / workspace
// caches) until the device list changes.  The per-job file layout (JOB-numbered archives, logs,
// split directories) is the reference's, so whatever consumes those directories keeps working.
//
//   scr/kaldi_scr.h:167-183  int ScoreKaldiWER(data, lang_or_graph, dir, wer filters, nj, stage, decode_mbr, stats, beam,
//                                                word_ins_penalty, min_lmwt, max_lmwt, iter)  scr/steps/scoring/score_kaldi_wer.cpp:36-254
// Decode writes what the reference's does: determinized lattices lat.JOB, words w.JOB, transcriptionJOB.trn (when
// the graph directory has words.txt), and -- unless skip_scoring -- scores them: for every word insertion penalty
// and LM weight the lattices are rescaled, penalised and their best path compared with data/text
// (wer_LMWT_WIP files in the reference's "%WER errs words ins del sub %SER serrs sents" form,
// scoring_kaldi/penalty_WIP/LMWT.txt hypotheses, scoring_kaldi/best_wer, wer_details/{lmwt,wip}).
// Scoring works on the lattices in memory (one read, all LMWT x WIP combinations spread over host
// threads) instead of the reference's 33 passes over temporary archives; the files it leaves are the same.
//
// Not part of this build, refused with an error rather than skipped: fake / two-channel / fake-dims CMVN, MBR decoding (decode_mbr).
// `segments` files are honoured (extract-segments' rules) and, with stats = true, the per-utterance alignment statistics of scoring
// (align-text, per_utt, per_spk, ops, bootstrap confidence interval) are produced (voicebridge_scoring.cc).
// Not declared at all (outside the hot path): Train*, AlignSi, MkGraph -- the search AlignSi runs, gmm-align-compiled, is
// the bin function GmmAlignCompiled (voicebridge_api.h).
#ifndef VOICEBRIDGE_STEPS_H_
#define VOICEBRIDGE_STEPS_H_

#include <string>
#include <unordered_map>

typedef std::unordered_map<std::string, std::string> UMAPSS;

int MakeMfcc(const std::string &datadir, const std::string &mfcc_config, int nj = 4, bool compress = true,
             bool write_utt2num_frames = false);
int ComputeCmvnStats(const std::string &datadir, bool fake = false, bool two_channel = false, const std::string &fake_dims = "");
int SplitData(const std::string &datadir, int numsplit, bool per_utt = false);
int Decode(const std::string &graph_dir, const std::string &data_dir, const std::string &decode_dir, const std::string &model,
           const std::string &trans_dir, UMAPSS &wer_ref_filter, UMAPSS &wer_hyp_filter, const std::string &iter = "", int nj = 4,
           float acwt = 0.083333f, int stage = 0, int max_active = 7000, double beam = 13.0, double lattice_beam = 6.0,
           bool skip_scoring = false, bool decode_mbr = false, bool stats = true, const std::string &word_ins_penalty = "0.0,0.5,1.0",
           int min_lmwt = 7, int max_lmwt = 17);
// scr/kaldi_scr.h:282-305  VOICEBRIDGE_API int DecodeFmllr(graphdir, data, dir, alignment_model, adapt_model, final_model, nj, acwt,
//                          stage, max_active, beam, lattice_beam, silence_weight, si_dir, fmllr_update_type, skip_scoring, decode_mbr,
//                          stats, word_ins_penalty, min_lmwt, max_lmwt)                     scr/steps/decode_fmllr.cpp:91-760
// Speaker-adapted decoding: a speaker-independent pass (Decode with beam 10 / max-active 2000 into <dir>.si) -> fMLLR from its lattice
// posteriors (pre_trans.JOB) -> main search on the transformed features (raw lattices lat.tmp.JOB) -> a second fMLLR estimate from the
// pruned-determinized lattices, composed with the first (trans.JOB) -> acoustic rescoring with the final model on the final features ->
// determinization (lat.JOB) -> best path (w.JOB, transcriptionJOB.trn) -> scoring.  On the GPU: the feature pipeline, both searches,
// (the first one in the search kernel's light layout), the fMLLR sufficient statistics and the row-by-row update from them
// (vbh_fmllr_accumulate, vbh_fmllr_update_full) and the scores the rescoring adds (vbh_gmm_loglikes_pairs); on the host: the lattice
// forward-backward and silence weighting (on the determinized lattices: milliseconds), determinization, scoring.  The small temporaries
// between the programs are in-memory tables (posteriors as objects) unless VB_STEP_FILE_TEMPS=1.
int DecodeFmllr(const std::string &graphdir, const std::string &data, const std::string &dir, const std::string &alignment_model = "",
                const std::string &adapt_model = "", const std::string &final_model = "", int nj = 4, float acwt = 0.083333f, int stage = 0,
                int max_active = 7000, double beam = 13.0, double lattice_beam = 6.0, double silence_weight = 0.01, const std::string &si_dir = "",
                const std::string &fmllr_update_type = "full", bool skip_scoring = false, bool decode_mbr = false, bool stats = true,
                const std::string &word_ins_penalty = "0.0,0.5,1.0", int min_lmwt = 7, int max_lmwt = 17);
int ScoreKaldiWER(const std::string &data, const std::string &lang_or_graph, const std::string &dir, UMAPSS &wer_ref_filter,
                  UMAPSS &wer_hyp_filter, int nj, int stage = 0, bool decode_mbr = false, bool stats = true, double beam = 6,
                  const std::string &word_ins_penalty = "0.0,0.5,1.0", int min_lmwt = 7, int max_lmwt = 17, const std::string &iter = "final");

// plain-C shims (ctypes, cgo ...); messages of the last failure: vb_steps_last_error()
extern "C" {
int vb_MakeMfcc(const char *datadir, const char *mfcc_config, int nj, int compress, int write_utt2num_frames);
int vb_ComputeCmvnStatsDir(const char *datadir);
int vb_SplitData(const char *datadir, int numsplit, int per_utt);
int vb_Decode(const char *graph_dir, const char *data_dir, const char *decode_dir, const char *model, int nj, float acwt,
              int max_active, double beam, double lattice_beam, int skip_scoring);
// the full form: trans_dir ("" = none), word insertion penalties "a,b,c", LM weight range
int vb_DecodeEx(const char *graph_dir, const char *data_dir, const char *decode_dir, const char *model, const char *trans_dir, int nj,
                float acwt, int max_active, double beam, double lattice_beam, int skip_scoring, const char *word_ins_penalty,
                int min_lmwt, int max_lmwt);
int vb_DecodeFmllr(const char *graphdir, const char *data, const char *dir, int nj, float acwt, int max_active, double beam, double lattice_beam,
                   double silence_weight, const char *si_dir, int skip_scoring, const char *word_ins_penalty, int min_lmwt, int max_lmwt);
int vb_ScoreKaldiWER(const char *data, const char *lang_or_graph, const char *dir, int nj, const char *word_ins_penalty, int min_lmwt,
                     int max_lmwt);
const char *vb_steps_last_error(void);
int vb_set_release_device_cache(int on);               // 1: every outermost step frees model, graph and search workspace when it returns (default 0: they stay for the next step; vb_release_device_cache() frees them)
int vb_last_step_jobs(int *out_device, int *out_worker, int cap);   // jobs of the last fan-out: device and worker thread (-1: the caller's) of each; returns their number
int vb_set_step_devices(const int *devices, int n);    // GPUs the jobs of the steps are spread over (n = 0: back to the default, four job threads on the calling thread's device)
}

#endif
