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
// Differences that a caller must know (all reported as errors, never silently ignored):
//   * compute-mfcc-feats: --dither must be 0 (the reference default 1.0 draws from libc rand());
//     --output-format=htk, --vtln-map, --min-duration, wav.scp pipes are not supported.
//   * gmm-latgen-faster: lattices are written with --determinize-lattice=false only (raw state-level
//     lattice, as DecodeFmllr's main pass does, VB/scr/steps/decode_fmllr.cpp:405-485);
//     determinization is the next row of SURVEY.md §8f.  --word-symbol-table is accepted and ignored
//     (it only echoes words to stderr in the reference).
#ifndef VOICEBRIDGE_API_H_
#define VOICEBRIDGE_API_H_

#include <fstream>

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
int vb_CopyFeats(int argc, char *argv[], const char *log_path);
int vb_set_device(int device_id);   // GPU used by the calling thread's subsequent calls (default 0)
}

#endif
