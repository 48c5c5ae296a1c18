// ref_decodable_test.cc -- BASELINE config 2 ("GPU scores, CPU decoder") inside the REFERENCE: the reference's own
// LatticeFasterDecoder (compiled from /root/reference by `make -C oracle ref`) decodes from scores that the product's
// GPU library computed, through include/voicebridge_decodable.h deriving from kaldi::DecodableInterface.  Test
// infrastructure (oracle/): built into oracle/_ref/vb_ref_decodable, run by tests/test_gpu_decodable.py.
//
// usage: vb_ref_decodable <dir>
//   inputs  <dir>/{feats,offs,weights,miv,iv,fst_final,fst_off,fst_arcs_i,fst_arcs_w,tid2pdf}.npy, opts.txt
//   outputs <dir>/out_{words,ali,best_weight,meta}.npy
#include <cstdio>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "decoder/lattice-faster-decoder.h"
#include "fstext/fstext-lib.h"
#include "itf/decodable-itf.h"
#include "lat/kaldi-lattice.h"

#include "voicebridge_decodable.h"

using namespace kaldi;

struct Npy { std::string dtype; std::vector<int64> shape; std::vector<char> data;
  int64 size() const { int64 n = 1; for (size_t i = 0; i < shape.size(); i++) n *= shape[i]; return n; }
  template <class T> T *ptr() { return reinterpret_cast<T *>(data.data()); } };
static Npy ReadNpy(const std::string &path) {
  std::ifstream f(path.c_str(), std::ios::binary);
  if (!f) { fprintf(stderr, "cannot open %s\n", path.c_str()); exit(2); }
  char magic[8]; f.read(magic, 8);
  unsigned short hl; f.read(reinterpret_cast<char *>(&hl), 2);
  std::string hdr(hl, ' '); f.read(&hdr[0], hl);
  Npy a;
  size_t p = hdr.find("'descr': '"); a.dtype = hdr.substr(p + 10, 3);
  p = hdr.find("'shape': ("); size_t q = hdr.find(")", p);
  std::stringstream ss(hdr.substr(p + 10, q - p - 10)); std::string tok;
  while (std::getline(ss, tok, ',')) if (tok.find_first_of("0123456789") != std::string::npos) a.shape.push_back(atoll(tok.c_str()));
  a.data.resize(a.size() * (a.dtype[2] - '0'));
  f.read(a.data.data(), a.data.size());
  return a;
}
static void WriteNpy(const std::string &path, const std::string &dtype, int64 n, const void *data, size_t elem) {
  std::ostringstream h;
  h << "{'descr': '" << dtype << "', 'fortran_order': False, 'shape': (" << n << ",), }";
  std::string hs = h.str();
  while ((10 + hs.size() + 1) % 64 != 0) hs += ' ';
  hs += '\n';
  std::ofstream f(path.c_str(), std::ios::binary);
  f.write("\x93NUMPY\x01\x00", 8);
  unsigned short hl = hs.size(); f.write(reinterpret_cast<char *>(&hl), 2);
  f.write(hs.data(), hs.size());
  f.write(reinterpret_cast<const char *>(data), n * elem);
}
static std::map<std::string, double> ReadOpts(const std::string &dir) {
  std::map<std::string, double> o; std::ifstream f((dir + "/opts.txt").c_str()); std::string line;
  while (std::getline(f, line)) { size_t e = line.find('='); if (e != std::string::npos) o[line.substr(0, e)] = atof(line.substr(e + 1).c_str()); }
  return o;
}
#define VB(call) do { if ((call) != 0) { fprintf(stderr, "%s failed: %s\n", #call, vbh_last_error()); return 1; } } while (0)

int main(int argc, char **argv) {
  if (argc != 2) { fprintf(stderr, "usage: %s <dir>\n", argv[0]); return 2; }
  const std::string dir = argv[1];
  try {
    std::map<std::string, double> o = ReadOpts(dir);
    Npy feats = ReadNpy(dir + "/feats.npy"), offs = ReadNpy(dir + "/offs.npy"), w = ReadNpy(dir + "/weights.npy"), miv = ReadNpy(dir + "/miv.npy"),
        iv = ReadNpy(dir + "/iv.npy"), t2p = ReadNpy(dir + "/tid2pdf.npy");
    Npy fin = ReadNpy(dir + "/fst_final.npy"), off = ReadNpy(dir + "/fst_off.npy"), ai = ReadNpy(dir + "/fst_arcs_i.npy"), aw = ReadNpy(dir + "/fst_arcs_w.npy");
    const int32 T = feats.shape[0], D = feats.shape[1], P = offs.size() - 1, num_tids = t2p.size() - 1;
    // ---- the product: all-pdf scores on the GPU, through the C ABI
    vbh_ctx *ctx = NULL; vbh_gmm *gmm = NULL;
    VB(vbh_ctx_create(0, &ctx));
    VB(vbh_gmm_create(ctx, D, P, offs.ptr<int32>(), w.ptr<float>(), miv.ptr<float>(), iv.ptr<float>(), &gmm));
    void *d_feats = NULL;
    VB(vbh_dev_alloc(ctx, sizeof(float) * (size_t)T * D, &d_feats));
    VB(vbh_copy_h2d(ctx, d_feats, feats.ptr<float>(), sizeof(float) * (size_t)T * D));
    VB(vbh_ctx_sync(ctx));
    const int64_t frame_offsets[2] = {0, T};
    vbh::GpuAcousticScores scores;
    scores.Compute(ctx, gmm, static_cast<const float *>(d_feats), frame_offsets, 1);
    // ---- the reference: its own graph class and decoder, reading the scores through its own interface
    fst::VectorFst<fst::StdArc> v;
    const int32 S = fin.size();
    for (int32 s = 0; s < S; s++) v.AddState();
    v.SetStart(0);
    for (int32 s = 0; s < S; s++) {
      const float f = fin.ptr<float>()[s];
      if (f != std::numeric_limits<float>::infinity()) v.SetFinal(s, fst::TropicalWeight(f));
      for (int64 a = off.ptr<int64>()[s]; a < off.ptr<int64>()[s + 1]; a++) {
        const int32 *r = ai.ptr<int32>() + 3 * a;
        v.AddArc(s, fst::StdArc(r[0], r[1], fst::TropicalWeight(aw.ptr<float>()[a]), r[2]));
      }
    }
    fst::ConstFst<fst::StdArc> g(v);
    LatticeFasterDecoderConfig cfg;
    cfg.beam = o.count("beam") ? o["beam"] : 13.0; cfg.max_active = o.count("max_active") ? (int32)o["max_active"] : 7000;
    cfg.min_active = o.count("min_active") ? (int32)o["min_active"] : 200; cfg.lattice_beam = o.count("lattice_beam") ? o["lattice_beam"] : 6.0;
    const float acwt = o.count("acwt") ? o["acwt"] : 0.083333;
    LatticeFasterDecoder decoder(g, cfg);
    vbh::DecodableGpuScores<DecodableInterface> decodable(scores.Rows(0), scores.NumFrames(0), scores.NumPdfs(), t2p.ptr<int32>(), num_tids, acwt);
    const bool ok = decoder.Decode(&decodable);
    std::vector<int32> ali, words; std::vector<float> bw(2, std::numeric_limits<float>::infinity());
    if (ok) {
      Lattice best; decoder.GetBestPath(&best);
      LatticeWeight wt; fst::GetLinearSymbolSequence(best, &ali, &words, &wt);
      bw[0] = wt.Value1(); bw[1] = wt.Value2();
    }
    std::vector<int32> meta; meta.push_back(ok); meta.push_back(decoder.ReachedFinal()); meta.push_back(decoder.NumFramesDecoded());
    WriteNpy(dir + "/out_words.npy", "<i4", words.size(), words.data(), 4); WriteNpy(dir + "/out_ali.npy", "<i4", ali.size(), ali.data(), 4);
    WriteNpy(dir + "/out_best_weight.npy", "<f4", 2, bw.data(), 4); WriteNpy(dir + "/out_meta.npy", "<i4", 3, meta.data(), 4);
    // the interface's error behaviour: an index the model does not have is the reference's exception
    bool threw = false;
    try { decodable.LogLikelihood(0, num_tids + 1); } catch (const std::runtime_error &) { threw = true; }
    if (!threw) { fprintf(stderr, "out-of-range transition-id did not throw\n"); return 1; }
    vbh_dev_free(ctx, d_feats); vbh_gmm_destroy(gmm); vbh_ctx_destroy(ctx);
  } catch (const std::exception &e) { fprintf(stderr, "error: %s\n", e.what()); return 1; }
  return 0;
}
