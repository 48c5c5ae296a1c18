// ref_harness.cc -- drives the REFERENCE's own C++ classes (compiled from /root/reference by
// oracle/Makefile `make ref`) on inputs supplied as .npy files and dumps their outputs as
// .npy files.  Test infrastructure only: it exists to (1) produce the golden vectors under
// tests/golden/ (oracle/gen_golden.py) and (2) validate oracle/vb_oracle_*.c.
// Nothing of the reference is copied here; this file only *calls* its public classes
// (and reads LatticeFasterDecoder's private token lists through an access define, as
// SURVEY.md §7 step 1 prescribes, because the reference offers no accessor for them).
//
// usage: vb_ref_harness <command> <dir>
//   inputs  <dir>/<name>.npy, options <dir>/opts.txt (key=value lines)
//   outputs <dir>/out_<name>.npy
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#define private public
#define protected public
#include "decoder/lattice-faster-decoder.h"
#include "decoder/faster-decoder.h"
#undef private
#undef protected

#include "base/kaldi-common.h"
#include "feat/feature-mfcc.h"
#include "feat/feature-functions.h"
#include "feat/wave-reader.h"
#include "fstext/fstext-lib.h"
#include "gmm/am-diag-gmm.h"
#include "gmm/decodable-am-diag-gmm.h"
#include "hmm/transition-model.h"
#include "hmm/hmm-test-utils.h"
#include "lat/kaldi-lattice.h"
#include "lat/lattice-functions.h"
#include "matrix/compressed-matrix.h"
#include "matrix/kaldi-matrix.h"
#include "transform/cmvn.h"
#include "util/common-utils.h"

using namespace kaldi;

// ------------------------------------------------------------------ tiny .npy reader/writer
struct Npy {
  std::string dtype;            // "<f4", "<f8", "<i4", "<i8", "|u1"
  std::vector<int64> shape;
  std::vector<char> data;
  int64 size() const { int64 n = 1; for (size_t i = 0; i < shape.size(); i++) n *= shape[i]; return n; }
  template <class T> T *ptr() { return reinterpret_cast<T *>(data.data()); }
};
static size_t DtypeSize(const std::string &d) { return d[2] - '0'; }

static Npy ReadNpy(const std::string &path) {
  std::ifstream f(path.c_str(), std::ios::binary);
  if (!f) { fprintf(stderr, "cannot open %s\n", path.c_str()); exit(2); }
  char magic[8]; f.read(magic, 8);
  unsigned short hl; f.read(reinterpret_cast<char *>(&hl), 2);
  std::string hdr(hl, ' '); f.read(&hdr[0], hl);
  Npy a;
  size_t p = hdr.find("'descr': '"); a.dtype = hdr.substr(p + 10, 3);
  if (hdr.find("'fortran_order': False") == std::string::npos) { fprintf(stderr, "fortran order\n"); exit(2); }
  p = hdr.find("'shape': ("); size_t q = hdr.find(")", p);
  std::string sh = hdr.substr(p + 10, q - p - 10);
  std::stringstream ss(sh); std::string tok;
  while (std::getline(ss, tok, ',')) { if (tok.find_first_of("0123456789") != std::string::npos) a.shape.push_back(atoll(tok.c_str())); }
  a.data.resize(a.size() * DtypeSize(a.dtype));
  f.read(a.data.data(), a.data.size());
  return a;
}
static void WriteNpy(const std::string &path, const std::string &dtype, const std::vector<int64> &shape, const void *data) {
  std::ostringstream h;
  h << "{'descr': '" << dtype << "', 'fortran_order': False, 'shape': (";
  for (size_t i = 0; i < shape.size(); i++) h << shape[i] << (shape.size() == 1 || i + 1 < shape.size() ? "," : "");
  h << "), }";
  std::string hs = h.str();
  while ((10 + hs.size() + 1) % 64 != 0) hs += ' ';
  hs += '\n';
  std::ofstream f(path.c_str(), std::ios::binary);
  f.write("\x93NUMPY\x01\x00", 8);
  unsigned short hl = hs.size(); f.write(reinterpret_cast<char *>(&hl), 2);
  f.write(hs.data(), hs.size());
  int64 n = 1; for (size_t i = 0; i < shape.size(); i++) n *= shape[i];
  f.write(reinterpret_cast<const char *>(data), n * DtypeSize(dtype));
}
static void WriteMat(const std::string &path, const MatrixBase<BaseFloat> &m) {
  std::vector<float> v(static_cast<size_t>(m.NumRows()) * m.NumCols());
  for (int32 r = 0; r < m.NumRows(); r++) for (int32 c = 0; c < m.NumCols(); c++) v[static_cast<size_t>(r) * m.NumCols() + c] = m(r, c);
  std::vector<int64> sh; sh.push_back(m.NumRows()); sh.push_back(m.NumCols());
  WriteNpy(path, "<f4", sh, v.data());
}
static void WriteVecF(const std::string &path, const std::vector<float> &v) { std::vector<int64> sh(1, v.size()); WriteNpy(path, "<f4", sh, v.data()); }
static void WriteVecI(const std::string &path, const std::vector<int32> &v) { std::vector<int64> sh(1, v.size()); WriteNpy(path, "<i4", sh, v.data()); }
static Matrix<BaseFloat> ReadMat(const std::string &path) {
  Npy a = ReadNpy(path);
  KALDI_ASSERT(a.dtype == "<f4" && a.shape.size() == 2);
  Matrix<BaseFloat> m(a.shape[0], a.shape[1]);
  const float *p = a.ptr<float>();
  for (int32 r = 0; r < m.NumRows(); r++) for (int32 c = 0; c < m.NumCols(); c++) m(r, c) = p[static_cast<size_t>(r) * m.NumCols() + c];
  return m;
}

typedef std::map<std::string, std::string> Opts;
static Opts ReadOpts(const std::string &dir) {
  Opts o; std::ifstream f((dir + "/opts.txt").c_str()); std::string line;
  while (std::getline(f, line)) { size_t p = line.find('='); if (p != std::string::npos) o[line.substr(0, p)] = line.substr(p + 1); }
  return o;
}
static double OptD(const Opts &o, const char *k, double d) { Opts::const_iterator i = o.find(k); return i == o.end() ? d : atof(i->second.c_str()); }
static std::string OptS(const Opts &o, const char *k, const char *d) { Opts::const_iterator i = o.find(k); return i == o.end() ? d : i->second; }

static MfccOptions MakeMfccOptions(const Opts &o) {
  MfccOptions m;
  FrameExtractionOptions &f = m.frame_opts;
  f.samp_freq = OptD(o, "samp_freq", f.samp_freq); f.frame_shift_ms = OptD(o, "frame_shift_ms", f.frame_shift_ms);
  f.frame_length_ms = OptD(o, "frame_length_ms", f.frame_length_ms); f.dither = OptD(o, "dither", 0.0);
  f.preemph_coeff = OptD(o, "preemph_coeff", f.preemph_coeff); f.remove_dc_offset = OptD(o, "remove_dc_offset", 1) != 0;
  f.window_type = OptS(o, "window_type", "povey"); f.round_to_power_of_two = OptD(o, "round_to_power_of_two", 1) != 0;
  f.blackman_coeff = OptD(o, "blackman_coeff", f.blackman_coeff); f.snip_edges = OptD(o, "snip_edges", 1) != 0;
  m.mel_opts.num_bins = OptD(o, "num_bins", 23); m.mel_opts.low_freq = OptD(o, "low_freq", 20);
  m.mel_opts.high_freq = OptD(o, "high_freq", 0); m.mel_opts.vtln_low = OptD(o, "vtln_low", 100);
  m.mel_opts.vtln_high = OptD(o, "vtln_high", -500); m.mel_opts.htk_mode = OptD(o, "htk_mode", 0) != 0;
  m.num_ceps = OptD(o, "num_ceps", 13); m.use_energy = OptD(o, "use_energy", 1) != 0;
  m.energy_floor = OptD(o, "energy_floor", 0); m.raw_energy = OptD(o, "raw_energy", 1) != 0;
  m.cepstral_lifter = OptD(o, "cepstral_lifter", 22); m.htk_compat = OptD(o, "htk_compat", 0) != 0;
  return m;
}

// ------------------------------------------------------------------ mfcc
// in: wave.npy [N] f4 (int16 range); out: out_feats [T x C]; stage dumps out_win [T x padded],
// out_pspec [T x padded/2+1], out_mel [T x nbins] (log mel energies), out_numframes
static int CmdMfcc(const std::string &dir) {
  Opts o = ReadOpts(dir);
  MfccOptions mo = MakeMfccOptions(o);
  BaseFloat vtln = OptD(o, "vtln_warp", 1.0);
  Npy w = ReadNpy(dir + "/wave.npy");
  Vector<BaseFloat> wave(w.size());
  for (int64 i = 0; i < w.size(); i++) wave(i) = w.ptr<float>()[i];
  Mfcc mfcc(mo);
  Matrix<BaseFloat> feats;
  mfcc.ComputeFeatures(wave, mo.frame_opts.samp_freq, vtln, &feats);
  WriteMat(dir + "/out_feats.npy", feats);
  std::vector<int32> nf(1, NumFrames(wave.Dim(), mo.frame_opts));
  WriteVecI(dir + "/out_numframes.npy", nf);
  // stage dumps through the reference's own stage functions
  int32 T = feats.NumRows(), padded = mo.frame_opts.PaddedWindowSize();
  if (T > 0 && (padded & (padded - 1)) == 0) {
    FeatureWindowFunction wf(mo.frame_opts);
    SplitRadixRealFft<BaseFloat> srfft(padded);
    MelBanks banks(mo.mel_opts, mo.frame_opts, vtln);
    Matrix<BaseFloat> win(T, padded), ps(T, padded / 2 + 1), mel(T, mo.mel_opts.num_bins);
    Vector<BaseFloat> window, mel_e(mo.mel_opts.num_bins);
    for (int32 t = 0; t < T; t++) {
      BaseFloat e;
      ExtractWindow(0, wave, t, mo.frame_opts, wf, &window, &e);
      win.Row(t).CopyFromVec(window);
      srfft.Compute(window.Data(), true);
      ComputePowerSpectrum(&window);
      SubVector<BaseFloat> p(window, 0, padded / 2 + 1);
      ps.Row(t).CopyFromVec(p);
      banks.Compute(p, &mel_e);
      mel_e.ApplyFloor(std::numeric_limits<BaseFloat>::epsilon());
      mel_e.ApplyLog();
      mel.Row(t).CopyFromVec(mel_e);
    }
    WriteMat(dir + "/out_win.npy", win); WriteMat(dir + "/out_pspec.npy", ps); WriteMat(dir + "/out_mel.npy", mel);
  }
  return 0;
}

// ------------------------------------------------------------------ feature post-processing
// in: feats.npy [T x D]; cmvn_stats.npy [2 x (D+1)] f8; xform.npy [R x C]
// out: out_deltas, out_splice, out_cmvn, out_cmvn_var, out_xform, out_stats, out_cm (bytes), out_cm_back
static int CmdPost(const std::string &dir) {
  Opts o = ReadOpts(dir);
  Matrix<BaseFloat> feats = ReadMat(dir + "/feats.npy");
  DeltaFeaturesOptions dopts(OptD(o, "delta_order", 2), OptD(o, "delta_window", 2));
  Matrix<BaseFloat> d; ComputeDeltas(dopts, feats, &d); WriteMat(dir + "/out_deltas.npy", d);
  Matrix<BaseFloat> s; SpliceFrames(feats, OptD(o, "splice_left", 4), OptD(o, "splice_right", 4), &s); WriteMat(dir + "/out_splice.npy", s);
  {
    Matrix<double> stats;
    InitCmvnStats(feats.NumCols(), &stats);
    AccCmvnStats(feats, NULL, &stats);
    std::vector<int64> sh; sh.push_back(2); sh.push_back(feats.NumCols() + 1);
    std::vector<double> v(2 * (feats.NumCols() + 1));
    for (int32 r = 0; r < 2; r++) for (int32 c = 0; c <= feats.NumCols(); c++) v[r * (feats.NumCols() + 1) + c] = stats(r, c);
    WriteNpy(dir + "/out_stats.npy", "<f8", sh, v.data());
  }
  {
    Npy st = ReadNpy(dir + "/cmvn_stats.npy");
    Matrix<double> stats(2, feats.NumCols() + 1);
    for (int32 r = 0; r < 2; r++) for (int32 c = 0; c <= feats.NumCols(); c++) stats(r, c) = st.ptr<double>()[r * (feats.NumCols() + 1) + c];
    Matrix<BaseFloat> a(feats); ApplyCmvn(stats, false, &a); WriteMat(dir + "/out_cmvn.npy", a);
    Matrix<BaseFloat> b(feats); ApplyCmvn(stats, true, &b); WriteMat(dir + "/out_cmvn_var.npy", b);
  }
  {
    // the body of TransformFeats (VB/src/featbin/transform-feats.cpp:89-107): sgemm (+ offset)
    Matrix<BaseFloat> trans = ReadMat(dir + "/xform.npy");
    int32 R = trans.NumRows(), C = trans.NumCols(), D = d.NumCols();
    Matrix<BaseFloat> out(d.NumRows(), R);
    if (C == D) out.AddMatMat(1.0, d, kNoTrans, trans, kTrans, 0.0);
    else {
      KALDI_ASSERT(C == D + 1);
      SubMatrix<BaseFloat> lin(trans, 0, R, 0, D);
      out.AddMatMat(1.0, d, kNoTrans, lin, kTrans, 0.0);
      Vector<BaseFloat> off(R); off.CopyColFromMat(trans, D);
      out.AddVecToRows(1.0, off);
    }
    WriteMat(dir + "/out_xform.npy", out);
  }
  {
    CompressedMatrix cm(feats);  // kAutomaticMethod -> kSpeechFeature when rows > 8
    std::ostringstream os; cm.Write(os, true);
    std::string blob = os.str();  // "CM " + header(16) + per-col headers + bytes
    std::vector<int64> sh(1, blob.size());
    WriteNpy(dir + "/out_cm.npy", "|u1", sh, blob.data());
    Matrix<BaseFloat> back(cm.NumRows(), cm.NumCols()); cm.CopyToMat(&back);
    WriteMat(dir + "/out_cm_back.npy", back);
  }
  return 0;
}

// ------------------------------------------------------------------ gmm
// in: offs.npy [P+1] i4, weights.npy [G], means.npy [G x D], inv_vars.npy [G x D], feats.npy [T x D]
// out: out_gconsts [G], out_miv [G x D], out_loglikes [T x P]; also out_model.mdl-less: am only
static void BuildAm(const std::string &dir, AmDiagGmm *am, std::vector<float> *gconsts, Matrix<BaseFloat> *miv_all) {
  Npy offs = ReadNpy(dir + "/offs.npy"); Npy w = ReadNpy(dir + "/weights.npy");
  Matrix<BaseFloat> means = ReadMat(dir + "/means.npy"), iv = ReadMat(dir + "/inv_vars.npy");
  int32 P = offs.size() - 1, D = means.NumCols();
  miv_all->Resize(means.NumRows(), D);
  for (int32 p = 0; p < P; p++) {
    int32 g0 = offs.ptr<int32>()[p], ng = offs.ptr<int32>()[p + 1] - g0;
    DiagGmm gmm(ng, D);
    Vector<BaseFloat> wv(ng);
    for (int32 g = 0; g < ng; g++) wv(g) = w.ptr<float>()[g0 + g];
    gmm.SetWeights(wv);
    gmm.SetInvVarsAndMeans(SubMatrix<BaseFloat>(iv, g0, ng, 0, D), SubMatrix<BaseFloat>(means, g0, ng, 0, D));
    gmm.ComputeGconsts();
    for (int32 g = 0; g < ng; g++) {
      gconsts->push_back(gmm.gconsts()(g));
      for (int32 d = 0; d < D; d++) (*miv_all)(g0 + g, d) = gmm.means_invvars()(g, d);
    }
    am->AddPdf(gmm);
  }
}
static int CmdGmm(const std::string &dir) {
  AmDiagGmm am; std::vector<float> gconsts; Matrix<BaseFloat> miv;
  BuildAm(dir, &am, &gconsts, &miv);
  Matrix<BaseFloat> feats = ReadMat(dir + "/feats.npy");
  DecodableAmDiagGmmUnmapped dec(am, feats);
  Matrix<BaseFloat> ll(feats.NumRows(), am.NumPdfs());
  for (int32 t = 0; t < feats.NumRows(); t++)
    for (int32 p = 0; p < am.NumPdfs(); p++) ll(t, p) = dec.LogLikelihood(t, p + 1);
  WriteVecF(dir + "/out_gconsts.npy", gconsts);
  WriteMat(dir + "/out_miv.npy", miv);
  WriteMat(dir + "/out_loglikes.npy", ll);
  return 0;
}

// ------------------------------------------------------------------ decoders
// in: fst_final.npy [S] f4 (inf = not final), fst_off.npy [S+1] i8, fst_arcs_i.npy [A x 3] i4
//     (ilabel, olabel, nextstate), fst_arcs_w.npy [A] f4, fst_start via opts, loglikes.npy [T x P],
//     tid2pdf.npy [ntid+1] i4.  opts: beam, max_active, min_active, lattice_beam, acwt, ...
class MatrixDecodable : public DecodableInterface {
 public:
  MatrixDecodable(const Matrix<BaseFloat> &ll, const std::vector<int32> &tid2pdf, BaseFloat scale)
      : ll_(ll), tid2pdf_(tid2pdf), scale_(scale) {}
  virtual BaseFloat LogLikelihood(int32 frame, int32 tid) { return scale_ * ll_(frame, tid2pdf_[tid]); }
  virtual bool IsLastFrame(int32 frame) const { return frame == ll_.NumRows() - 1; }
  virtual int32 NumFramesReady() const { return ll_.NumRows(); }
  virtual int32 NumIndices() const { return tid2pdf_.size() - 1; }
 private:
  const Matrix<BaseFloat> &ll_; const std::vector<int32> &tid2pdf_; BaseFloat scale_;
};

static fst::ConstFst<fst::StdArc> *LoadFst(const std::string &dir, const Opts &o) {
  Npy fin = ReadNpy(dir + "/fst_final.npy"), off = ReadNpy(dir + "/fst_off.npy"),
      ai = ReadNpy(dir + "/fst_arcs_i.npy"), aw = ReadNpy(dir + "/fst_arcs_w.npy");
  fst::VectorFst<fst::StdArc> v;
  int32 S = fin.size();
  for (int32 s = 0; s < S; s++) v.AddState();
  v.SetStart(static_cast<int32>(OptD(o, "start", 0)));
  for (int32 s = 0; s < S; s++) {
    float f = fin.ptr<float>()[s];
    if (f != std::numeric_limits<float>::infinity()) v.SetFinal(s, fst::TropicalWeight(f));
    for (int64 a = off.ptr<int64>()[s]; a < off.ptr<int64>()[s + 1]; a++) {
      const int32 *r = ai.ptr<int32>() + 3 * a;
      v.AddArc(s, fst::StdArc(r[0], r[1], fst::TropicalWeight(aw.ptr<float>()[a]), r[2]));
    }
  }
  return new fst::ConstFst<fst::StdArc>(v);
}

static int CmdDecode(const std::string &dir) {
  Opts o = ReadOpts(dir);
  fst::ConstFst<fst::StdArc> *g = LoadFst(dir, o);
  Matrix<BaseFloat> ll = ReadMat(dir + "/loglikes.npy");
  Npy t2p = ReadNpy(dir + "/tid2pdf.npy");
  std::vector<int32> tid2pdf(t2p.ptr<int32>(), t2p.ptr<int32>() + t2p.size());
  BaseFloat acwt = OptD(o, "acwt", 0.083333);
  {
    LatticeFasterDecoderConfig cfg;
    cfg.beam = OptD(o, "beam", 13.0); cfg.max_active = OptD(o, "max_active", 7000);
    cfg.min_active = OptD(o, "min_active", 200); cfg.lattice_beam = OptD(o, "lattice_beam", 6.0);
    cfg.prune_interval = OptD(o, "prune_interval", 25); cfg.beam_delta = OptD(o, "beam_delta", 0.5);
    cfg.hash_ratio = OptD(o, "hash_ratio", 2.0);
    LatticeFasterDecoder dec(*g, cfg);
    MatrixDecodable decodable(ll, tid2pdf, acwt);
    bool ok = dec.Decode(&decodable);
    std::vector<int32> meta;  // ok, reached_final, num_frames
    meta.push_back(ok); meta.push_back(dec.ReachedFinal()); meta.push_back(dec.NumFramesDecoded());
    WriteVecI(dir + "/out_lat_meta.npy", meta);
    std::vector<float> fm; fm.push_back(dec.FinalRelativeCost()); fm.push_back(dec.final_best_cost_);
    WriteVecF(dir + "/out_lat_fmeta.npy", fm);
    std::vector<float> co(dec.cost_offsets_.begin(), dec.cost_offsets_.end());
    WriteVecF(dir + "/out_lat_cost_offsets.npy", co);
    // surviving tokens need their HCLG state: the reference keeps the state only in the
    // hash key, so recover it by walking links: dst state of a link is arc.nextstate, which
    // the link does not store either.  Instead reconstruct states by replaying: token ->
    // state is recoverable from the raw lattice only through ilabel/olabel; so we obtain it
    // from a parallel map filled by intercepting the per-frame hash before it is cleared.
    // Simplest faithful way: re-run decoding frame by frame with AdvanceDecoding(1) and
    // snapshot toks_.GetList() (state, Token*) after every frame.
    LatticeFasterDecoder dec2(*g, cfg);
    MatrixDecodable decodable2(ll, tid2pdf, acwt);
    dec2.InitDecoding();
    unordered_map<LatticeFasterDecoder::Token *, int32> state_of;
    std::vector<int32> ntok_frame;  // tokens in the hash after each frame (before pruning)
    for (const LatticeFasterDecoder::Elem *e = dec2.toks_.GetList(); e != NULL; e = e->tail) state_of[e->val] = e->key;
    ntok_frame.push_back(state_of.size());
    while (dec2.NumFramesDecoded() < ll.NumRows()) {
      dec2.AdvanceDecoding(&decodable2, 1);
      int32 c = 0;
      for (const LatticeFasterDecoder::Elem *e = dec2.toks_.GetList(); e != NULL; e = e->tail, c++) state_of[e->val] = e->key;
      ntok_frame.push_back(c);
    }
    dec2.FinalizeDecoding();
    WriteVecI(dir + "/out_lat_ntok_frame.npy", ntok_frame);
    // NOTE: Token* keys of pruned tokens may be reused by the allocator for later tokens;
    // the map is updated on every frame so the entry for a *live* token is always current
    // (a live token's address cannot be reused while it is alive).
    std::vector<int32> tf, ts; std::vector<float> tc, te;
    std::vector<int32> ls, ld, li, lo; std::vector<float> lg, la;
    unordered_map<LatticeFasterDecoder::Token *, int32> idx;
    // index tokens sorted per frame by state
    for (size_t f = 0; f < dec2.active_toks_.size(); f++) {
      std::vector<std::pair<int32, LatticeFasterDecoder::Token *> > v;
      for (LatticeFasterDecoder::Token *t = dec2.active_toks_[f].toks; t != NULL; t = t->next) v.push_back(std::make_pair(state_of[t], t));
      std::sort(v.begin(), v.end());
      for (size_t i = 0; i < v.size(); i++) {
        idx[v[i].second] = tf.size();
        tf.push_back(f); ts.push_back(v[i].first); tc.push_back(v[i].second->tot_cost); te.push_back(v[i].second->extra_cost);
      }
    }
    for (size_t f = 0; f < dec2.active_toks_.size(); f++)
      for (LatticeFasterDecoder::Token *t = dec2.active_toks_[f].toks; t != NULL; t = t->next)
        for (LatticeFasterDecoder::ForwardLink *l = t->links; l != NULL; l = l->next) {
          BaseFloat cost_offset = (l->ilabel != 0) ? dec2.cost_offsets_[f] : 0.0;
          ls.push_back(idx[t]); ld.push_back(idx[l->next_tok]); li.push_back(l->ilabel); lo.push_back(l->olabel);
          lg.push_back(l->graph_cost); la.push_back(l->acoustic_cost - cost_offset);
        }
    WriteVecI(dir + "/out_lat_tok_frame.npy", tf); WriteVecI(dir + "/out_lat_tok_state.npy", ts);
    WriteVecF(dir + "/out_lat_tok_cost.npy", tc); WriteVecF(dir + "/out_lat_tok_extra.npy", te);
    WriteVecI(dir + "/out_lat_link_src.npy", ls); WriteVecI(dir + "/out_lat_link_dst.npy", ld);
    WriteVecI(dir + "/out_lat_link_ilabel.npy", li); WriteVecI(dir + "/out_lat_link_olabel.npy", lo);
    WriteVecF(dir + "/out_lat_link_graph.npy", lg); WriteVecF(dir + "/out_lat_link_acoustic.npy", la);
    // best path exactly as DecodeUtteranceLatticeFaster does (decoder-wrappers.cc:233-258)
    std::vector<int32> ali, words; std::vector<float> bw(2, std::numeric_limits<float>::infinity());
    if (ok) {
      Lattice best; dec.GetBestPath(&best);
      LatticeWeight w; fst::GetLinearSymbolSequence(best, &ali, &words, &w);
      bw[0] = w.Value1(); bw[1] = w.Value2();
      // n-best over the determinized lattice is covered by the §8f "next" row, not here.
      Lattice raw; dec.GetRawLattice(&raw);
      std::vector<int32> rs(2); rs[0] = raw.NumStates(); int32 na = 0;
      for (int32 s = 0; s < raw.NumStates(); s++) na += raw.NumArcs(s);
      rs[1] = na;
      WriteVecI(dir + "/out_lat_raw_size.npy", rs);
    }
    WriteVecI(dir + "/out_lat_ali.npy", ali); WriteVecI(dir + "/out_lat_words.npy", words); WriteVecF(dir + "/out_lat_best_weight.npy", bw);
  }
  {
    FasterDecoderOptions fo;
    fo.beam = OptD(o, "beam", 13.0); fo.max_active = OptD(o, "max_active", 7000);
    fo.min_active = OptD(o, "faster_min_active", 20); fo.beam_delta = OptD(o, "beam_delta", 0.5);
    fo.hash_ratio = OptD(o, "hash_ratio", 2.0);
    FasterDecoder dec(*g, fo);
    MatrixDecodable decodable(ll, tid2pdf, acwt);
    if (OptD(o, "faster_dump_frames", 0) != 0) {
      // Decode() unrolled (faster-decoder.cc:47-54) so that the per-frame state can be dumped: tokens alive after every frame's
      // ProcessNonemitting, the cutoff ProcessEmitting returned
      std::vector<int32> ntok; std::vector<float> cut;
      dec.InitDecoding();
      while (!decodable.IsLastFrame(dec.num_frames_decoded_ - 1)) {
        { const int32 df = OptD(o, "faster_dump_tokens_at", -1);
          if (df >= 0 && dec.num_frames_decoded_ == df) {
            size_t cnt; BaseFloat ab; FasterDecoder::Elem *be = NULL;
            double wc0 = dec.GetCutoff(const_cast<FasterDecoder::Elem *>(dec.toks_.GetList()), &cnt, &ab, &be);
            int32 below = 0, eq = 0;
            for (const FasterDecoder::Elem *e = dec.toks_.GetList(); e != NULL; e = e->tail) { below += e->val->cost_ < wc0; eq += e->val->cost_ == wc0; }
            fprintf(stderr, "G frame %d: GetCutoff -> %.9f adaptive_beam %.9f, %d tokens, %d strictly below, %d equal; min_active %d max_active %d beam %.3f\n",
                    dec.num_frames_decoded_, wc0, ab, (int)cnt, below, eq, dec.config_.min_active, dec.config_.max_active, dec.config_.beam);
          } }
        double weight_cutoff = dec.ProcessEmitting(&decodable);
        dec.ProcessNonemitting(weight_cutoff);
        int32 c = 0; double best = std::numeric_limits<double>::infinity();
        for (const FasterDecoder::Elem *e = dec.toks_.GetList(); e != NULL; e = e->tail) { c++; if (e->val->cost_ < best) best = e->val->cost_; }
        ntok.push_back(c); cut.push_back(static_cast<float>(weight_cutoff - best)); cut.push_back(static_cast<float>(best));
        { const int32 df = OptD(o, "faster_dump_tokens_at", -1);
          if (df >= 0 && (dec.num_frames_decoded_ == df || dec.num_frames_decoded_ == df + 1))
            for (const FasterDecoder::Elem *e = dec.toks_.GetList(); e != NULL; e = e->tail) fprintf(stderr, "T %d %d %.6f\n", dec.num_frames_decoded_, e->key, e->val->cost_); }
      }
      WriteVecI(dir + "/out_fast_ntok.npy", ntok); WriteVecF(dir + "/out_fast_cutoff_minus_best.npy", cut);
    } else
    dec.Decode(&decodable);
    std::vector<int32> ali, words; std::vector<float> bw(2, std::numeric_limits<float>::infinity());
    std::vector<int32> meta; meta.push_back(0); meta.push_back(dec.ReachedFinal()); meta.push_back(dec.NumFramesDecoded());
    fst::VectorFst<LatticeArc> best;
    if (dec.GetBestPath(&best)) {
      meta[0] = 1;
      LatticeWeight w; fst::GetLinearSymbolSequence(best, &ali, &words, &w);
      bw[0] = w.Value1(); bw[1] = w.Value2();
    }
    WriteVecI(dir + "/out_fast_meta.npy", meta);
    WriteVecI(dir + "/out_fast_ali.npy", ali); WriteVecI(dir + "/out_fast_words.npy", words); WriteVecF(dir + "/out_fast_best_weight.npy", bw);
  }
  delete g;
  return 0;
}

// ------------------------------------------------------------------ model / graph files
// Writes a reference-format model (<dir>/out_final.mdl, binary), graph (<dir>/out_HCLG.fst),
// feature archive (<dir>/out_feats.ark, compressed) so the build's own readers can be
// tested against files produced by the reference's writers.
static int CmdFiles(const std::string &dir) {
  Opts o = ReadOpts(dir);
  AmDiagGmm am; std::vector<float> gconsts; Matrix<BaseFloat> miv;
  BuildAm(dir, &am, &gconsts, &miv);
  // transition model over a random topology whose pdf count matches the acoustic model
  std::vector<int32> phones; int32 nphones = OptD(o, "num_phones", 4);
  for (int32 i = 1; i <= nphones; i++) phones.push_back(i);
  srand(static_cast<unsigned>(OptD(o, "seed", 7)));
  ContextDependency *ctx = NULL;
  TransitionModel *tm = GenRandTransitionModel(&ctx);
  {
    Output ko(dir + "/out_final.mdl", true);
    tm->Write(ko.Stream(), true);
    // acoustic model sized to the transition model's pdfs: reuse pdf 0.. cyclically
    AmDiagGmm am2;
    for (int32 p = 0; p < tm->NumPdfs(); p++) am2.AddPdf(am.GetPdf(p % am.NumPdfs()));
    am2.Write(ko.Stream(), true);
    std::vector<int32> t2p(tm->NumTransitionIds() + 1, -1);
    for (int32 t = 1; t <= tm->NumTransitionIds(); t++) t2p[t] = tm->TransitionIdToPdf(t);
    WriteVecI(dir + "/out_tid2pdf.npy", t2p);
    std::vector<int32> np(2); np[0] = tm->NumPdfs(); np[1] = am2.Dim();
    WriteVecI(dir + "/out_mdl_meta.npy", np);
    // flattened model arrays of am2 for the reader test
    std::vector<int32> offs(1, 0); std::vector<float> gc, wts, mv, ivv;
    for (int32 p = 0; p < am2.NumPdfs(); p++) {
      const DiagGmm &g = am2.GetPdf(p);
      for (int32 k = 0; k < g.NumGauss(); k++) {
        gc.push_back(g.gconsts()(k)); wts.push_back(g.weights()(k));
        for (int32 d = 0; d < g.Dim(); d++) { mv.push_back(g.means_invvars()(k, d)); ivv.push_back(g.inv_vars()(k, d)); }
      }
      offs.push_back(offs.back() + g.NumGauss());
    }
    WriteVecI(dir + "/out_mdl_offs.npy", offs); WriteVecF(dir + "/out_mdl_gconsts.npy", gc);
    WriteVecF(dir + "/out_mdl_weights.npy", wts); WriteVecF(dir + "/out_mdl_miv.npy", mv); WriteVecF(dir + "/out_mdl_iv.npy", ivv);
  }
  delete tm; delete ctx;
  {
    fst::ConstFst<fst::StdArc> *g = LoadFst(dir, o);
    g->Write(dir + "/out_HCLG.fst");
    delete g;
  }
  {
    Matrix<BaseFloat> feats = ReadMat(dir + "/feats.npy");
    BaseFloatMatrixWriter plain("ark,scp:" + dir + "/out_feats_plain.ark," + dir + "/out_feats_plain.scp");
    CompressedMatrixWriter comp("ark,scp:" + dir + "/out_feats.ark," + dir + "/out_feats.scp");
    plain.Write("utt1", feats); plain.Write("utt2", feats);
    comp.Write("utt1", CompressedMatrix(feats)); comp.Write("utt2", CompressedMatrix(feats));
  }
  {
    // PCM-16 wav written by the reference's WaveData writer
    Npy w = ReadNpy(dir + "/wave.npy");
    Matrix<BaseFloat> data(1, w.size());
    for (int64 i = 0; i < w.size(); i++) data(0, i) = w.ptr<float>()[i];
    WaveData wd(16000, data);
    Output ko(dir + "/out_test.wav", true, false);
    wd.Write(ko.Stream());
  }
  return 0;
}

int main(int argc, char **argv) {
  if (argc != 3) { fprintf(stderr, "usage: %s <mfcc|post|gmm|decode|files> <dir>\n", argv[0]); return 2; }
  std::string cmd = argv[1], dir = argv[2];
  try {
    if (cmd == "mfcc") return CmdMfcc(dir);
    if (cmd == "post") return CmdPost(dir);
    if (cmd == "gmm") return CmdGmm(dir);
    if (cmd == "decode") return CmdDecode(dir);
    if (cmd == "files") return CmdFiles(dir);
  } catch (const std::exception &e) { fprintf(stderr, "harness error: %s\n", e.what()); return 1; }
  fprintf(stderr, "unknown command %s\n", cmd.c_str());
  return 2;
}
