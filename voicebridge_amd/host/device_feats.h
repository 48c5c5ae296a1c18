// device_feats.h -- a feature table on the device, shared by the bin functions that read features (gmm-latgen-faster,
// gmm-est-fmllr, gmm-rescore-lattice).
//
// The reference's Decode / DecodeFmllr steps build, for every program that needs features, a pipeline of feature programs
// over temporary archives: apply-cmvn | add-deltas (or splice-feats | transform-feats final.mat) [| transform-feats trans]
// (VB/scr/steps/decode_gmm.cpp:201-293, decode_fmllr.cpp:275-311, 390-402, 491-500, 568-580).  Here the same chain is the
// fused device pass vbh_featpost, selected by the --fe-* options, in front of whichever program consumes the features:
//   --fe-cmvn=<cmvn rspecifier> [--fe-utt2spk=<rspecifier>] [--fe-norm-vars=true]
//   --fe-type=delta [--fe-delta-order=2 --fe-delta-window=2]  |  --fe-type=lda --fe-lda=<final.mat> [--fe-left-context=4 --fe-right-context=4]
//   [--fe-trans=<per-speaker (or per-utterance) transform rspecifier>]
// so that the only features ever read from disk are the raw ones of feats.scp -- and those once per job: the raw rows of the
// table a thread read last stay on the device (keyed by rspecifier and file signature) for the next program of the job.
#ifndef VOICEBRIDGE_DEVICE_FEATS_H_
#define VOICEBRIDGE_DEVICE_FEATS_H_

#include <map>
#include <memory>
#include <string>
#include <vector>

#include "api_common.h"
#include "kaldi_io.h"
#include "voicebridge_hip.h"

namespace vbapi {

struct DevBuffer {       // RAII device memory through the C ABI
  vbh_ctx *ctx; void *p = nullptr; size_t bytes = 0;
  DevBuffer(vbh_ctx *c, size_t n);
  ~DevBuffer();
  DevBuffer(const DevBuffer &) = delete;
  DevBuffer &operator=(const DevBuffer &) = delete;
};

struct FrontEnd {        // the --fe-* options
  bool fused = false;
  std::map<std::string, vbio::MatrixD> stats;
  std::map<std::string, std::string> utt2spk;
  std::map<std::string, vbio::Matrix> trans;
  vbio::Matrix lda;
  vbh_featpost_cfg cfg{};
  static void Know(Options *po);          // registers the option names
  void Parse(const Options &po);
};

struct RawTable {        // the rows of every utterance of a feature table, in table order, on the device
  std::string rspec; long long sig = -2; vbh_ctx *ctx = nullptr;
  std::vector<std::string> keys;
  std::vector<int64_t> offs{0};           // rows of utterance u: [offs[u], offs[u+1]) (an entry of zero rows is kept, with no rows)
  std::map<std::string, int> index;       // key -> position
  int dim = 0;
  std::unique_ptr<DevBuffer> d_raw;
  int64_t frames() const { return offs.back(); }
  int rows(size_t u) const { return (int)(offs[u + 1] - offs[u]); }
};
// the table of rspecifier on the calling thread's device: read and uploaded, or the thread's cached copy when rspecifier and
// the signature (size, modification time) of the file it names are those of the last call
const RawTable &CachedRawTable(const std::string &rspecifier);
void ReleaseRawTableCache();              // (vb_release_device_cache)

// A model's device form (vbh_gmm) of the calling thread, kept from program to program: gmm-est-fmllr and gmm-rescore-lattice name the
// same two or three model files job after job, and creating and destroying the device model per call is a dozen hipMalloc / hipFree --
// every hipFree waits for ALL work of the device, i.e. for the other jobs' persistent search grids: the second fMLLR estimate of a job
// whose search had ended early took 745 ms for 143 ms of work (round 5).  Keyed by path and file signature; a thread's models are
// never shared (a vbh_gmm holds per-call buffers); released by vb_release_device_cache.
vbh_gmm *CachedDeviceGmm(const std::string &path, const vbio::Model &model);
void ReleaseDeviceGmmCache();

// Features of utterances [u0, u1) of the table behind the front end -- or the table's own rows when no --fe-* options were
// given -- as a device matrix [rows x *dim]; the returned pointer is into *hold (front end) or into the table (plain).
const float *DeviceFeatures(const FrontEnd &fe, const RawTable &t, size_t u0, size_t u1, int *dim, std::unique_ptr<DevBuffer> *hold);

}  // namespace vbapi
#endif
