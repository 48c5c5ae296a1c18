// kaldi_io.h -- readers/writers for the on-disk formats the hot path touches.
// These are *formats*, re-implemented from their description; nothing of Kaldi/OpenFst is
// linked or copied.  Reference definitions (paths relative to /root/reference):
//   binary header "\0B", tokens, basic ints   kaldi-master/src/base/io-funcs-inl.h:34-52,291-325
//   Matrix "FM "/"DM ", Vector "FV "/"DV "   kaldi-master/src/matrix/kaldi-matrix.cc:1375-1401, kaldi-vector.cc:1221-1227
//   CompressedMatrix "CM"/"CM2"/"CM3"         kaldi-master/src/matrix/compressed-matrix.cc:129-187,347-600
//   archives / script files                    kaldi-master/src/util/kaldi-table-inl.h:1055,1370-1382
//   RIFF PCM-16                                kaldi-master/src/feat/wave-reader.cc:137-310
//   .mdl = TransitionModel + AmDiagGmm         kaldi-master/src/hmm/transition-model.cc:383-409, hmm/hmm-topology.cc:127-160,
//                                              gmm/am-diag-gmm.cc:147-161, gmm/diag-gmm.cc:705-756
//   ConstFst / VectorFst<StdArc>               openfst-win-1.6/src/include/fst/const-fst.h:102-110,195-330, vector-fst.h, fst.h (FstHeader)
//   Lattice (VectorFst<LatticeArc>)            kaldi-master/src/lat/kaldi-lattice.cc:62-90,388-413
#ifndef VOICEBRIDGE_KALDI_IO_H_
#define VOICEBRIDGE_KALDI_IO_H_

#include <cstdint>
#include <deque>
#include <fstream>
#include <istream>
#include <map>
#include <ostream>
#include <stdexcept>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "lattice.h"

namespace vbio {

struct IoError : public std::runtime_error {
  explicit IoError(const std::string &m) : std::runtime_error(m) {}
};

struct Matrix {            // row-major float32, no stride padding
  int32_t rows = 0, cols = 0;
  std::vector<float> data;
  float &at(int r, int c) { return data[(size_t)r * cols + c]; }
  float at(int r, int c) const { return data[(size_t)r * cols + c]; }
};
struct MatrixD { int32_t rows = 0, cols = 0; std::vector<double> data; };

// ---- basic binary pieces
bool ReadBinaryHeader(std::istream &is);                    // consumes "\0B" if present; returns binary flag
void ExpectToken(std::istream &is, bool binary, const char *tok);
std::string ReadToken(std::istream &is, bool binary);
int32_t ReadInt32(std::istream &is, bool binary);
float ReadFloat(std::istream &is, bool binary);
void ReadIntVector(std::istream &is, bool binary, std::vector<int32_t> *v);
void WriteToken(std::ostream &os, bool binary, const char *tok);
void WriteInt32(std::ostream &os, bool binary, int32_t v);

// ---- matrices / vectors (object body, after the optional "\0B")
void ReadMatrix(std::istream &is, bool binary, Matrix *m);          // FM / DM / CM / CM2 / CM3 / text
void ReadMatrixD(std::istream &is, bool binary, MatrixD *m);        // DM / FM / text (CMVN stats)
void ReadVector(std::istream &is, bool binary, std::vector<float> *v);
void WriteMatrix(std::ostream &os, bool binary, const Matrix &m);   // "FM "
void WriteMatrixD(std::ostream &os, bool binary, const MatrixD &m); // "DM "
// CompressedMatrix, speech-feature format ("CM "): bit-exact with the reference's writer
void CompressMatrix(const Matrix &m, std::vector<uint8_t> *blob);   // header(16) + col headers + bytes
void WriteCompressedMatrix(std::ostream &os, const Matrix &m);      // "CM " + blob; falls back to "FM " when rows <= 8? no: see .cc

// ---- tables
struct ScpEntry { std::string key, path; int64_t offset = -1; };
std::vector<ScpEntry> ReadScp(const std::string &path);
struct Rspec { bool is_scp = false, text = false; std::string path; };
struct Wspec { bool has_ark = false, has_scp = false, text = false; std::string ark_path, scp_path; };
Rspec ParseRspecifier(const std::string &s);
Wspec ParseWspecifier(const std::string &s);

// sequential reader over "ark:" or "scp:" of matrices
class MatrixReader {
 public:
  explicit MatrixReader(const std::string &rspecifier);
  bool Next(std::string *key, Matrix *m);   // false at end
 private:
  Rspec spec_;
  std::ifstream ark_;
  std::vector<ScpEntry> scp_;
  size_t pos_ = 0;
};
// all entries of a table, in table order; the entries of an scp table (a file position each) are read and decoded on up to
// `threads` host threads -- a job's feature table is hundreds of compressed matrices, and everything the job does waits for it
void ReadMatricesInOrder(const std::string &rspecifier, int threads, std::vector<std::string> *keys, std::vector<Matrix> *mats);
// all entries of a table of matrices keyed by utterance/speaker (transforms, cmvn stats)
std::map<std::string, Matrix> ReadMatrixTable(const std::string &rspecifier);
std::map<std::string, MatrixD> ReadMatrixDTable(const std::string &rspecifier);
std::map<std::string, std::string> ReadStringMap(const std::string &rspecifier);   // utt2spk

// one matrix as a table writer stores it: text, "FM " or (compress) the speech-feature "CM " forms -- the costly part of writing a
// compressed feature archive (a sort per column), separable from the file so that utterances can be encoded on several threads
std::string SerializeMatrix(const Matrix &m, bool text, bool compress);

class MatrixWriter {   // "ark:", "ark,t:", "ark,scp:a,b", "scp:"? (scp alone is not a valid wspecifier)
 public:
  explicit MatrixWriter(const std::string &wspecifier, bool compress = false);
  void Write(const std::string &key, const Matrix &m);
  // the object exactly as Write would have serialised it (made with SerializeMatrix, possibly on another thread)
  void WriteSerialized(const std::string &key, const std::string &object);
  bool compress() const { return compress_; }
  bool text() const { return spec_.text; }
  void Close();
 private:
  Wspec spec_;
  bool compress_;
  std::ofstream ark_, scp_;
};

class IntVectorWriter {   // Int32VectorWriter, text or binary
 public:
  IntVectorWriter() {}
  explicit IntVectorWriter(const std::string &wspecifier) { Open(wspecifier); }
  void Open(const std::string &wspecifier);
  bool IsOpen() const { return open_; }
  void Write(const std::string &key, const std::vector<int32_t> &v);
 private:
  Wspec spec_;
  bool open_ = false;
  std::ofstream ark_;
};

// ---- wave
struct Wave { float samp_freq = 0; int channels = 0; std::vector<std::vector<float>> data; };
void ReadWave(std::istream &is, Wave *w);
void ReadWaveFile(const std::string &path, int64_t offset, Wave *w);
// wav.scp entries may be "key path" (plain files); pipes ("cmd |") are not supported

// ---- model
struct Model {
  // TransitionModel: id2pdf_id_ (index 0 unused) -- K/hmm/transition-model.cc:144-177
  std::vector<int32_t> tid2pdf;
  int32_t num_pdfs = 0, dim = 0;
  std::vector<int32_t> pdf_offsets;            // [num_pdfs+1]
  std::vector<float> weights, means_invvars, inv_vars, gconsts_in_file;
  // per transition-id (index 0 unused): TransitionIdToPhone / TransitionIdToHmmState / IsSelfLoop
  // (K/hmm/transition-model.h:160-175, .cc:144-177) -- what lattice determinization asks of the model
  vblat::TransInfo trans;
  // transition probabilities (K/hmm/transition-model.h:148-158,.cc:364-381): log_probs_ per transition-id (index 0 unused), the
  // transition-state of every transition-id, and per transition-state (index 0 unused) the log of 1 - its self-loop probability
  std::vector<float> log_probs;
  std::vector<int32_t> tid2tstate;
  std::vector<float> non_self_loop_log_probs;
};
void ReadModel(const std::string &path, Model *m);
// GetScaledTransitionLogProb (K/hmm/hmm-utils.cc:1019-1038): what AddTransitionProbs adds (negated) to the arc of a transition-id
float ScaledTransitionLogProb(const Model &m, int32_t tid, float transition_scale, float self_loop_scale);

// ---- graphs
struct Arc { int32_t ilabel, olabel; float weight; int32_t nextstate; };
struct Graph {
  int32_t start = -1;
  std::vector<float> final_cost;               // +inf = not final
  std::vector<int64_t> arc_offset;             // [S+1]
  std::vector<Arc> arcs;
};
void ReadFst(const std::string &path, Graph *g);   // "const" or "vector", arc type "standard"
void ReadFstStream(std::istream &is, Graph *g, const std::string &what);   // the same from an open stream (leaves it behind the FST)
// sequential reader over a table of graphs ("ark:file", binary VectorFst<StdArc> entries as compile-train-graphs writes them:
// VectorFstTplHolder, K/fstext/kaldi-fst-io-inl.h:170-207)
class FstTableReader {
 public:
  explicit FstTableReader(const std::string &rspecifier);
  bool Next(std::string *key, Graph *g);    // false at end
 private:
  Rspec spec_;
  std::ifstream ark_;
};
// BaseFloatWriter / BaseFloatVectorWriter (K/util/kaldi-holder-inl.h BasicHolder<float>, KaldiObjectHolder<Vector<float>>)
class FloatWriter {
 public:
  FloatWriter() {}
  void Open(const std::string &wspecifier);
  bool IsOpen() const { return open_; }
  void Write(const std::string &key, float v);
 private:
  Wspec spec_; bool open_ = false; std::ofstream ark_;
};
class FloatVectorWriter {
 public:
  FloatVectorWriter() {}
  void Open(const std::string &wspecifier);
  bool IsOpen() const { return open_; }
  void Write(const std::string &key, const std::vector<float> &v);
 private:
  Wspec spec_; bool open_ = false; std::ofstream ark_;
};

// In-memory lattice tables.  An archive path that starts with "mem:" ("ark:mem:NAME") names a table that never touches a file: the
// writers keep every entry's bytes (moved in where the caller gives them up), publish the table under its name when they are destroyed,
// and LatticeReader walks the entries in place.  For the temporary tables a step hands from one program to the next inside one process
// -- DecodeFmllr's raw lattices are 100 MB per job, written once and read twice (VB/scr/steps/decode_fmllr.cpp:390-640 writes lat.tmp.JOB
// and rescored lattices as files); MemUnlink drops a table as unlink() drops a file.
typedef std::vector<std::pair<std::string, std::string>> MemTable;      // (key, serialised object)
bool IsMemPath(const std::string &path);
void MemPublish(const std::string &path, std::shared_ptr<MemTable> t);
std::shared_ptr<MemTable> MemLookup(const std::string &path);           // nullptr: no such table
void MemUnlink(const std::string &path);

// raw lattice: one state per token, as GetRawLattice builds it
struct LatticeArcOut { int32_t ilabel, olabel; float graph, acoustic; int32_t nextstate; };
struct LatticeOut {
  std::vector<std::vector<LatticeArcOut>> arcs;   // per state
  std::vector<float> final_graph, final_acoustic; // +inf graph = not final
};
class LatticeWriter {   // LatticeWriter ("Lattice" holder), binary archives
 public:
  LatticeWriter() {}
  ~LatticeWriter();
  void Open(const std::string &wspecifier);
  bool IsOpen() const { return open_; }
  void Write(const std::string &key, const LatticeOut &lat);
  void Write(const std::string &key, const vblat::Lattice &lat);
  void WriteSerialized(const std::string &key, const std::string &object);   // object: what SerializeLattice made
  void WriteSerialized(const std::string &key, std::string &&object);        // ... given up by the caller (an in-memory table takes it without a copy)
 private:
  Wspec spec_;
  bool open_ = false;
  std::ofstream ark_;
  std::string buf_;
  std::shared_ptr<MemTable> mem_;
};
// the bytes of one archive entry (after the key); callers serialise on their own threads and hand the result to WriteSerialized
void SerializeLattice(const LatticeOut &lat, std::string *b);
void SerializeLattice(const vblat::Lattice &lat, std::string *b);
void SerializeLattice(const vblat::CompactLattice &clat, std::string *b);


// CompactLatticeWriter ("CompactLattice" holder, kaldi-lattice.cc:62-70): binary archives of
// VectorFst<CompactLatticeArc> (arc type "compactlattice44": weight = two floats, int32 string
// length, int32 transition-ids)
class CompactLatticeWriter {
 public:
  CompactLatticeWriter() {}
  ~CompactLatticeWriter();
  void Open(const std::string &wspecifier);
  bool IsOpen() const { return open_; }
  void Write(const std::string &key, const vblat::CompactLattice &clat);
  void WriteSerialized(const std::string &key, const std::string &object);
  void WriteSerialized(const std::string &key, std::string &&object);
 private:
  Wspec spec_;
  bool open_ = false;
  std::ofstream ark_;
  std::string buf_;
  std::shared_ptr<MemTable> mem_;
};

// sequential reader over binary lattice archives ("ark:" or "scp:"); an entry is either a
// Lattice ("lattice4") or a CompactLattice ("compactlattice44")
class LatticeReader {
 public:
  explicit LatticeReader(const std::string &rspecifier);
  ~LatticeReader();
  LatticeReader(const LatticeReader &) = delete;
  LatticeReader &operator=(const LatticeReader &) = delete;
  // fills *lat or *clat according to what the entry holds (*is_compact tells which)
  bool Next(std::string *key, vblat::Lattice *lat, vblat::CompactLattice *clat, bool *is_compact);
  // convenience: the entry as a CompactLattice / Lattice; the other kind is an error
  bool NextCompact(std::string *key, vblat::CompactLattice *clat);
  bool NextLattice(std::string *key, vblat::Lattice *lat);
  // the entry's bytes only (valid while the reader lives): callers parse them on their own threads with ParseLatticeBytes
  bool NextBytes(std::string *key, const char **bytes, size_t *n);
 private:
  Rspec spec_;
  const char *map_ = nullptr;      // "ark:": the whole archive, mapped
  size_t map_len_ = 0;
  std::vector<ScpEntry> scp_;
  size_t pos_ = 0;                 // scp: next line; ark: offset into the mapping
  std::deque<std::string> held_;   // scp: the entries NextBytes handed out
  std::shared_ptr<MemTable> mem_;  // "ark:mem:NAME": the table, walked entry by entry (pos_ = next entry)
};
void ParseLatticeBytes(const char *p, size_t n, vblat::Lattice *lat, vblat::CompactLattice *clat, bool *is_compact);
void ParseLatticeBytes(const char *p, size_t n, vblat::Lattice *lat);     // a compact entry is converted (LatticeHolder::Read)
std::string LatticeObjectToText(const std::string &object);                // a serialised entry in the text form of "ark,t:" archives

// Posterior / GaussPost tables (K/hmm/posterior.cc:32-62,127-215), binary archives
typedef vblat::Posterior Posterior;
typedef std::vector<std::vector<std::pair<int32_t, std::vector<float>>>> GaussPost;    // per frame: (pdf-id, per-Gaussian posteriors)
void WritePosteriorEntry(std::ostream &os, const std::string &key, const Posterior &post, bool text = false);
void SerializePosterior(const Posterior &post, std::string *b, bool text = false);      // the entry's bytes after "key " (text: the "ark,t:" form)
void WriteGaussPostEntry(std::ostream &os, const std::string &key, const GaussPost &gpost);
std::vector<std::pair<std::string, Posterior>> ReadPosteriorTable(const std::string &rspecifier);
// The same tables with an utterance's frames in one array: frame t holds e[off[t] .. off[t + 1]).  What the programs between lattice-to-post
// and gmm-est-fmllr work on -- a job's table has a million frames of one or two pairs each, and a std::vector per frame made reading,
// weighting and converting it a matter of allocations (round 5: 0.6 s of one core per 1024 x 30 s in DecodeFmllr's chain).  A table written
// to "ark:mem:NAME" is published as the object itself (no bytes are made); a file gets Kaldi's binary / text format, byte for byte what
// WritePosteriorEntry writes.  Both readers take both forms.
struct FlatPost {
  std::vector<int32_t> off{0};
  std::vector<std::pair<int32_t, float>> e;
  size_t frames() const { return off.size() - 1; }
  void EndFrame() { off.push_back((int32_t)e.size()); }
};
typedef std::vector<std::pair<std::string, FlatPost>> FlatPostTable;
void FlattenPosterior(const Posterior &post, FlatPost *flat);
std::shared_ptr<const FlatPostTable> ReadPosteriorTableFlat(const std::string &rspecifier);
void WritePosteriorTableFlat(const std::string &wspecifier, std::shared_ptr<FlatPostTable> table);
std::vector<std::pair<std::string, GaussPost>> ReadGaussPostTable(const std::string &rspecifier);
std::ofstream OpenArkForWriting(const std::string &wspecifier, bool *text = nullptr);      // "ark:" (binary); "ark,t:" when the caller takes the flag
// spk2utt: speaker -> utterances, in file order
std::vector<std::pair<std::string, std::vector<std::string>>> ReadSpk2Utt(const std::string &rspecifier);
void WriteFloat(std::ostream &os, bool binary, float v);

// Int32Vector tables in text or binary form (words, alignments): key -> ints
std::vector<std::pair<std::string, std::vector<int32_t>>> ReadIntVectorTable(const std::string &rspecifier);
// BaseFloat tables (one float per key, text "key 0.95" or binary): VTLN warp factors per speaker / utterance
std::map<std::string, float> ReadFloatTable(const std::string &rspecifier);
// token tables ("key tok tok ..." text lines): reference transcripts for compute-wer
std::vector<std::pair<std::string, std::vector<std::string>>> ReadTokenTable(const std::string &rspecifier);

}  // namespace vbio
#endif
