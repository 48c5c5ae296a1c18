// voicebridge_api.h -- the reference's argv-style "bin function" interface for the hot path,
// re-declared for this build and implemented on top of the C ABI (include/voicebridge_hip.h).
//
// Mirrors (paths relative to /root/reference/VoiceBridge/VoiceBridge/kaldi-win/):
//   src/kaldi_src.h:18   int ComputeMFCCFeats(int argc, char *argv[], fs::ofstream &file_log);
//   src/kaldi_src.h:28   int ApplyCmvn(int argc, char *argv[], fs::ofstream &file_log);
//   src/kaldi_src.h:29   int AddDeltas(int argc, char *argv[]);
//   src/kaldi_src.h:31   int SpliceFeats(int argc, char *argv[]);
//   src/kaldi_src.h:32   int TransformFeats(int argc, char *argv[], fs::ofstream &file_log);
//   src/kaldi_src.h:22   int ComputeCmvnStats(int argc, char *argv[], fs::ofstream &file_log);
//   src/kaldi_src.h:47   int GmmLatgenFaster(int argc, char *argv[], fs::ofstream &file_log);
// Same names, same Kaldi command-line grammar in argv (argv[0] = program name, --opt=value,
// then r/w-specifiers), same return convention: 0 = at least one utterance done, 1 = none done,
// -1 = error (never throws across the boundary).  The only signature difference is the log
// stream: std::ofstream& instead of boost::filesystem::ofstream& (which derives from it, so a
// caller holding an fs::ofstream passes it unchanged).  Log text goes to file_log when it is
// open, else to stderr.
//
// Differences that a caller must know (all reported as errors or warnings, never silently ignored):
//   * compute-mfcc-feats: --dither is honoured (default 1.0 as the reference; voicebridge_api.cc:246-252) from a counter-based
//     generator, so dithered features equal the reference's statistically and bit for bit only with --dither=0 (the reference
//     draws from libc rand()); --dither-seed (extension) selects the stream.  --vtln-map / --utt2spk, --min-duration, --segments
//     are supported.  Refused: --output-format=htk, --allow-downsample; a wav.scp entry that is a pipe ("cmd |") is skipped with
//     a warning, as an unreadable file is.
//   * gmm-latgen-faster: --determinize-lattice=true (the default) writes CompactLattices determinized by the reference's
//     two-pass (phone level, then word level) pruned algorithm, restated in host/lattice.cc; --minimize=true is supported
//     (voicebridge_api.cc:761; bit-exact against the reference programs, tests/test_host_lattice.py).  --word-symbol-table is
//     accepted and ignored (it only echoes words to stderr in the reference).
//   * lattice, posterior and int-vector tables are read and written in binary and in text mode ("ark,t:"); models are read in both
//     modes and written in binary only.
//   * apply-cmvn: --norm-means=false / --skip-dims / --reverse are refused; compute-cmvn-stats: --weights is refused.
#ifndef VOICEBRIDGE_API_H_
#define VOICEBRIDGE_API_H_

#include <fstream>
#include <string>

int ComputeMFCCFeats(int argc, char *argv[], std::ofstream &file_log);
int ComputeCmvnStats(int argc, char *argv[], std::ofstream &file_log);
int ApplyCmvn(int argc, char *argv[], std::ofstream &file_log);
int AddDeltas(int argc, char *argv[]);
int SpliceFeats(int argc, char *argv[]);
int TransformFeats(int argc, char *argv[], std::ofstream &file_log);
int GmmLatgenFaster(int argc, char *argv[], std::ofstream &file_log);
int CopyFeats(int argc, char *argv[], std::ofstream &file_log);          // VB/src/kaldi_src.h:20
// not wrapped by the reference DLL; upstream program K/gmmbin/gmm-decode-faster.cc with the same conventions
int GmmDecodeFaster(int argc, char *argv[], std::ofstream &file_log);
// gmm-align-compiled (VB/src/gmmbin/gmm-align-compiled.cpp:24-160 over AlignUtteranceWrapper, K/decoder/decoder-wrappers.cc:424-516):
// <model> <graphs-rspecifier> <feature-rspecifier> <alignments-wspecifier> [scores-wspecifier]; --beam, --retry-beam,
// --transition-scale, --acoustic-scale, --self-loop-scale, --write-per-frame-acoustic-loglikes as the reference's; --careful=true is
// refused; --decode-batch=N utterances per launch of the batched FasterDecoder (extension).  voicebridge_alignbin.cc
int GmmAlignCompiled(int argc, char *argv[], std::ofstream &file_log);

// ---- lattices after the search; host code in voicebridge_latbin.cc
//   src/kaldi_src.h:94   int LatticeDeterminizePruned(int argc, char *argv[], fs::ofstream &file_log);
//   src/kaldi_src.h:92   int LatticeScale(...);   :89 LatticeAddPenalty(...);   :88 LatticeBestPath(...);
//   src/kaldi_src.h:64   int ComputeWer(int argc, char *argv[], fs::ofstream &file_log);
int LatticeDeterminizePruned(int argc, char *argv[], std::ofstream &file_log);
int LatticeScale(int argc, char *argv[], std::ofstream &file_log);
int LatticeAddPenalty(int argc, char *argv[], std::ofstream &file_log);
int LatticeBestPath(int argc, char *argv[], std::ofstream &file_log);
int ComputeWer(int argc, char *argv[], std::ofstream &file_log);           // the three result lines go to the log stream
// upstream program K/latbin/lattice-determinize-phone-pruned.cc (what gmm-latgen-faster does per utterance)
int LatticeDeterminizePhonePruned(int argc, char *argv[], std::ofstream &file_log);

// ---- between the two searches of DecodeFmllr; host code in voicebridge_fmllrbin.cc
//   src/kaldi_src.h:93 LatticeToPost, :84 WeightSilencePost, :55 GmmPostToGpost, :56 GmmEstFmllrGpost, :53 GmmEstFmllr,
//   :34 ComposeTransforms, :57 GmmRescoreLattice  (all: int f(int argc, char *argv[], fs::ofstream &file_log))
int LatticeToPost(int argc, char *argv[], std::ofstream &file_log);
int WeightSilencePost(int argc, char *argv[], std::ofstream &file_log);
int GmmPostToGpost(int argc, char *argv[], std::ofstream &file_log);
int GmmEstFmllrGpost(int argc, char *argv[], std::ofstream &file_log);
int GmmEstFmllr(int argc, char *argv[], std::ofstream &file_log);
int ComposeTransforms(int argc, char *argv[], std::ofstream &file_log);
int GmmRescoreLattice(int argc, char *argv[], std::ofstream &file_log);

// ---- the per-utterance statistics of ScoreKaldiWER (`stats`); host code in voicebridge_scoring.cc
//   src/kaldi_src.h:75   int AlignText(int argc, char *argv[]);                     (K/bin/align-text.cc)
//   src/kaldi_src.h:63   int ComputeWerBootci(int argc, char *argv[], fs::ofstream &file_log);   (K/bin/compute-wer-bootci.cc + output file)
//   scr/kaldi_scr.h:204-224  WerPerUttDetails / WerPerSpkDetails / WerOpsDetails    (scr/utils/scoring/wer_*_details.cpp)
int AlignText(int argc, char *argv[]);
int AlignTextLogged(int argc, char *argv[], std::ofstream &file_log);       // the same with its messages in a log file
int ComputeWerBootci(int argc, char *argv[], std::ofstream &file_log);
int WerPerUttDetails(const std::string &in, const std::string &out, const std::string &special_symbol = "<eps>", const std::string &separator = ";",
                     bool output_hyp = true, bool output_ref = true, bool output_ops = true, bool output_csid = true);
int WerPerSpkDetails(const std::string &in, const std::string &utt2spk, const std::string &out, int WIDTH = 10, int SPK_WIDTH = 15);
int WerOpsDetails(const std::string &in, const std::string &out, const std::string &special_symbol = "<eps>");

// plain-C shims of the above for hosts without C++ streams (ctypes, cgo ...): log_path may be NULL
extern "C" {
int vb_ComputeMFCCFeats(int argc, char *argv[], const char *log_path);
int vb_ComputeCmvnStats(int argc, char *argv[], const char *log_path);
int vb_ApplyCmvn(int argc, char *argv[], const char *log_path);
int vb_AddDeltas(int argc, char *argv[]);
int vb_SpliceFeats(int argc, char *argv[]);
int vb_TransformFeats(int argc, char *argv[], const char *log_path);
int vb_GmmLatgenFaster(int argc, char *argv[], const char *log_path);
int vb_GmmDecodeFaster(int argc, char *argv[], const char *log_path);
int vb_GmmAlignCompiled(int argc, char *argv[], const char *log_path);
int vb_CopyFeats(int argc, char *argv[], const char *log_path);
int vb_LatticeDeterminizePruned(int argc, char *argv[], const char *log_path);
int vb_LatticeDeterminizePhonePruned(int argc, char *argv[], const char *log_path);
int vb_LatticeScale(int argc, char *argv[], const char *log_path);
int vb_LatticeAddPenalty(int argc, char *argv[], const char *log_path);
int vb_LatticeBestPath(int argc, char *argv[], const char *log_path);
int vb_ComputeWer(int argc, char *argv[], const char *log_path);
int vb_LatticeToPost(int argc, char *argv[], const char *log_path);
int vb_WeightSilencePost(int argc, char *argv[], const char *log_path);
int vb_GmmPostToGpost(int argc, char *argv[], const char *log_path);
int vb_GmmEstFmllrGpost(int argc, char *argv[], const char *log_path);
int vb_GmmEstFmllr(int argc, char *argv[], const char *log_path);
int vb_ComposeTransforms(int argc, char *argv[], const char *log_path);
int vb_GmmRescoreLattice(int argc, char *argv[], const char *log_path);
int vb_AlignText(int argc, char *argv[]);
int vb_ComputeWerBootci(int argc, char *argv[], const char *log_path);
int vb_WerPerUttDetails(const char *in, const char *out, const char *special_symbol);
int vb_WerPerSpkDetails(const char *in, const char *utt2spk, const char *out);
int vb_WerOpsDetails(const char *in, const char *out, const char *special_symbol);
void vb_ComputeWerCounts(long long *out8);   // {word_errs, num_words, ins, del, sub, num_sent, sent_errs, absent} of the last ComputeWer
// search workspaces of following gmm-latgen-faster calls cover this max_active too (0: off).  PROCESS-WIDE (the jobs of a step run on
// threads of their own): a caller sets it for the duration of a step and puts the old value back (vb_get_workspace_max_active)
int vb_set_workspace_max_active(int max_active);
int vb_get_workspace_max_active(void);
int vb_set_device(int device_id);   // GPU used by the calling thread's subsequent calls (default 0)
int vb_get_device(void);
// GmmLatgenFaster keeps the model, the graph and the search workspace of its last call on the device (the nj jobs of a step name the
// same files again); this frees them
void vb_release_device_cache(void);
// the same, and the calling thread's GPU context (stream, staging arenas) with it: call on a thread that is about to end
void vb_release_thread_context(void);
}

#endif
