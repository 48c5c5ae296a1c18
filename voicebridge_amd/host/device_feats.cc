// device_feats.cc -- see device_feats.h
#include "device_feats.h"

#include <sys/stat.h>

#include <cstring>

namespace vbapi {

namespace {
void Check(int rc, const char *what) {
  if (rc != 0) throw ApiError(std::string(what) + ": " + vbh_last_error());
}
long long FileSigOf(const std::string &path) {
  struct stat st;
  if (stat(path.c_str(), &st) != 0) return -1;
  return (long long)st.st_size * 1000003LL + (long long)st.st_mtime * 7919LL + (long long)st.st_mtim.tv_nsec;
}
thread_local RawTable g_table;
}  // namespace

DevBuffer::DevBuffer(vbh_ctx *c, size_t n) : ctx(c), bytes(n) { Check(vbh_dev_alloc(c, n ? n : 4, &p), "vbh_dev_alloc"); }
DevBuffer::~DevBuffer() { if (p) vbh_dev_free(ctx, p); }

void FrontEnd::Know(Options *po) {
  po->Know({"fe-cmvn", "fe-utt2spk", "fe-norm-vars", "fe-type", "fe-delta-order", "fe-delta-window", "fe-left-context", "fe-right-context", "fe-lda",
            "fe-trans"});
}

void FrontEnd::Parse(const Options &po) {
  fused = po.Has("fe-cmvn");
  if (!fused) {
    for (const char *n : {"fe-utt2spk", "fe-norm-vars", "fe-type", "fe-lda", "fe-trans"})
      if (po.Has(n)) throw ApiError(std::string("--") + n + " needs --fe-cmvn (the fused front end starts with apply-cmvn)");
    return;
  }
  stats = vbio::ReadMatrixDTable(po.Str("fe-cmvn", ""));
  if (po.Has("fe-utt2spk")) utt2spk = vbio::ReadStringMap(po.Str("fe-utt2spk", ""));
  cfg = vbh_featpost_cfg{};
  cfg.apply_cmvn = 1;
  cfg.norm_vars = po.Bool("fe-norm-vars", false);
  const std::string type = po.Str("fe-type", "delta");
  if (type == "delta") { cfg.mode = 1; cfg.delta_order = (int)po.Num("fe-delta-order", 2); cfg.delta_window = (int)po.Num("fe-delta-window", 2); }
  else if (type == "lda") {
    cfg.mode = 2; cfg.splice_left = (int)po.Num("fe-left-context", 4); cfg.splice_right = (int)po.Num("fe-right-context", 4);
    if (!po.Has("fe-lda")) throw ApiError("--fe-type=lda needs --fe-lda=<final.mat>");
    std::ifstream f(po.Str("fe-lda", ""), std::ios::binary);
    if (!f) throw ApiError("cannot open " + po.Str("fe-lda", ""));
    const bool binary = vbio::ReadBinaryHeader(f);
    vbio::ReadMatrix(f, binary, &lda);
    cfg.lda_rows = lda.rows; cfg.lda_cols = lda.cols;
  } else throw ApiError("--fe-type must be delta or lda");
  if (po.Has("fe-trans")) trans = vbio::ReadMatrixTable(po.Str("fe-trans", ""));
}

const RawTable &CachedRawTable(const std::string &rspecifier) {
  vbh_ctx *ctx = SharedCtx();
  const vbio::Rspec rs = vbio::ParseRspecifier(rspecifier);
  const long long sig = FileSigOf(rs.path);
  RawTable &t = g_table;
  if (t.ctx == ctx && t.rspec == rspecifier && t.sig == sig && sig >= 0 && t.d_raw) return t;
  t = RawTable();
  // (four host threads read and decode the entries: the job has nothing else to do until its features are on the device)
  std::vector<std::string> keys; std::vector<vbio::Matrix> mats;
  vbio::ReadMatricesInOrder(rspecifier, 4, &keys, &mats);
  for (size_t i = 0; i < keys.size(); i++) {
    const vbio::Matrix &m = mats[i];
    if (m.rows > 0) {
      if (t.dim == 0) t.dim = m.cols;
      if (m.cols != t.dim) throw ApiError("feature table " + rspecifier + ": " + keys[i] + " has dimension " + std::to_string(m.cols) + ", the entries before it " + std::to_string(t.dim));
    }
    t.index[keys[i]] = (int)t.keys.size();
    t.keys.push_back(keys[i]);
    t.offs.push_back(t.offs.back() + (m.rows > 0 ? m.rows : 0));
  }
  const size_t total = (size_t)t.offs.back() * (size_t)t.dim;
  std::unique_ptr<float[]> data_p(new float[std::max<size_t>(total, 1)]);
  struct { float *p; size_t n; float *data() { return p; } size_t size() const { return n; } bool empty() const { return n == 0; } } data{data_p.get(), total};
  for (size_t i = 0; i < mats.size(); i++)
    if (mats[i].rows > 0) memcpy(data.p + (size_t)t.offs[i] * (size_t)t.dim, mats[i].data.data(), sizeof(float) * mats[i].data.size());
  t.d_raw.reset(new DevBuffer(ctx, sizeof(float) * data.size()));
  if (!data.empty()) Check(vbh_copy_h2d(ctx, t.d_raw->p, data.data(), sizeof(float) * data.size()), "vbh_copy_h2d");
  Check(vbh_ctx_sync(ctx), "vbh_ctx_sync");      // `data` goes out of scope
  t.ctx = ctx; t.rspec = rspecifier; t.sig = sig;
  return t;
}

void ReleaseRawTableCache() { g_table = RawTable(); }

namespace {
struct GmmCache {
  vbh_ctx *ctx = nullptr;
  std::map<std::string, vbh_gmm *> models;
  void Clear() { for (auto &kv : models) vbh_gmm_destroy(kv.second); models.clear(); ctx = nullptr; }
};
thread_local GmmCache g_gmms;
long long ModelSig(const std::string &path) {
  struct stat st;
  if (stat(path.c_str(), &st) != 0) return -1;
  return (long long)st.st_size * 1000003LL + (long long)st.st_mtime * 7919LL + (long long)st.st_mtim.tv_nsec;
}
}  // namespace
vbh_gmm *CachedDeviceGmm(const std::string &path, const vbio::Model &model) {
  vbh_ctx *ctx = SharedCtx();
  if (g_gmms.ctx != ctx) g_gmms.Clear();      // (the thread's context was replaced: its device changed)
  g_gmms.ctx = ctx;
  const std::string key = std::to_string(ModelSig(path)) + "|" + path;
  auto it = g_gmms.models.find(key);
  if (it != g_gmms.models.end()) return it->second;
  if (g_gmms.models.size() >= 8) g_gmms.Clear(), g_gmms.ctx = ctx;      // (a step names two or three models)
  vbh_gmm *g = nullptr;
  if (vbh_gmm_create(ctx, model.dim, model.num_pdfs, model.pdf_offsets.data(), model.weights.data(), model.means_invvars.data(), model.inv_vars.data(), &g) != 0)
    throw ApiError(std::string("vbh_gmm_create: ") + vbh_last_error());
  g_gmms.models[key] = g;
  return g;
}
void ReleaseDeviceGmmCache() { g_gmms.Clear(); }

const float *DeviceFeatures(const FrontEnd &fe, const RawTable &t, size_t u0, size_t u1, int *dim, std::unique_ptr<DevBuffer> *hold) {
  const float *d_in = static_cast<const float *>(t.d_raw->p) + t.offs[u0] * (int64_t)t.dim;
  if (!fe.fused) { *dim = t.dim; return d_in; }
  vbh_ctx *ctx = t.ctx;
  vbh_featpost_cfg cfg = fe.cfg;
  cfg.in_dim = t.dim;
  // CMVN statistics and transforms of the range's speakers, packed in the order the speakers appear
  std::map<std::string, int> idx; std::vector<int32_t> u2s; std::vector<double> stats; std::vector<float> xf;
  std::vector<int64_t> offs;
  for (size_t u = u0; u < u1; u++) {
    const std::string &k = t.keys[u];
    offs.push_back(t.offs[u] - t.offs[u0]);
    std::string spk = k;
    if (!fe.utt2spk.empty()) { auto it = fe.utt2spk.find(k); if (it != fe.utt2spk.end()) spk = it->second; }
    if (!idx.count(spk)) {
      auto it = fe.stats.find(spk);
      if (it == fe.stats.end()) throw ApiError("No normalization statistics available for key " + k);
      const vbio::MatrixD &ms = it->second;
      if (ms.cols != t.dim + 1 || ms.rows < 1 || ms.rows > 2) throw ApiError("Dim mismatch: cmvn " + std::to_string(ms.rows) + "x" + std::to_string(ms.cols));
      if (ms.rows == 1 && cfg.norm_vars) throw ApiError("You requested variance normalization but no variance stats are supplied.");
      stats.insert(stats.end(), ms.data.begin(), ms.data.end());
      if (ms.rows == 1) stats.insert(stats.end(), (size_t)(t.dim + 1), 0.0);
      if (!fe.trans.empty()) {
        auto tt = fe.trans.find(spk);
        if (tt == fe.trans.end()) tt = fe.trans.find(k);          // per-utterance transforms
        if (tt == fe.trans.end()) throw ApiError("No fMLLR transform for speaker " + spk);
        if (cfg.spk_rows == 0) { cfg.spk_rows = tt->second.rows; cfg.spk_cols = tt->second.cols; }
        if (tt->second.rows != cfg.spk_rows || tt->second.cols != cfg.spk_cols) throw ApiError("fMLLR transforms of different sizes");
        xf.insert(xf.end(), tt->second.data.begin(), tt->second.data.end());
      }
      const int n = (int)idx.size(); idx[spk] = n;
    }
    u2s.push_back(idx[spk]);
  }
  offs.push_back(t.offs[u1] - t.offs[u0]);
  const int od = vbh_featpost_out_dim(&cfg);
  const int64_t T = t.offs[u1] - t.offs[u0];
  // the output lives in scratch slot 0 of the thread's context: the programs of a job run one after the other on that context (stream
  // order), and a buffer allocated and freed per call would make every call wait for the other jobs' searches (hipFree waits for the device)
  (void)hold;
  void *d_out = nullptr;
  Check(vbh_ctx_scratch(ctx, 0, sizeof(float) * (size_t)std::max<int64_t>(T, 1) * od, &d_out), "vbh_ctx_scratch");
  if (T > 0)
    Check(vbh_featpost(ctx, &cfg, d_in, offs.data(), (int32_t)(u1 - u0), u2s.data(), (int)idx.size(), stats.data(),
                       fe.lda.data.empty() ? nullptr : fe.lda.data.data(), xf.empty() ? nullptr : xf.data(), (float *)d_out), "vbh_featpost");
  *dim = od;
  return static_cast<const float *>(d_out);
}

}  // namespace vbapi
