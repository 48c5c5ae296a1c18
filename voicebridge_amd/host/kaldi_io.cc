// kaldi_io.cc -- see kaldi_io.h.  Formats re-implemented from their on-disk description.
#include "kaldi_io.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <sstream>
#include <thread>

namespace vbio {

static void Fail(const std::string &m) { throw IoError(m); }

// ------------------------------------------------------------------ basic pieces
bool ReadBinaryHeader(std::istream &is) {
  if (is.peek() == '\0') {
    is.get();
    if (is.peek() != 'B') Fail("bad binary header");
    is.get();
    return true;
  }
  return false;
}

std::string ReadToken(std::istream &is, bool binary) {
  std::string tok;
  if (!binary) is >> std::ws;
  is >> tok;
  if (is.fail()) Fail("ReadToken: stream failure");
  if (!isspace(is.peek())) Fail("ReadToken: expected space after token " + tok);
  is.get();
  return tok;
}

void ExpectToken(std::istream &is, bool binary, const char *tok) {
  const std::string t = ReadToken(is, binary);
  if (t != tok) Fail(std::string("expected token ") + tok + ", got " + t);
}

int32_t ReadInt32(std::istream &is, bool binary) {
  int32_t v = 0;
  if (binary) {
    const int sz = is.get();
    if (sz != 4) Fail("ReadInt32: size byte " + std::to_string(sz));
    is.read(reinterpret_cast<char *>(&v), 4);
  } else {
    is >> v;
  }
  if (is.fail()) Fail("ReadInt32: stream failure");
  return v;
}

float ReadFloat(std::istream &is, bool binary) {
  if (binary) {
    const int sz = is.get();
    if (sz == 4) { float f; is.read(reinterpret_cast<char *>(&f), 4); return f; }
    if (sz == 8) { double d; is.read(reinterpret_cast<char *>(&d), 8); return (float)d; }
    Fail("ReadFloat: size byte " + std::to_string(sz));
  }
  float f;
  is >> f;
  if (is.fail()) Fail("ReadFloat: stream failure");
  return f;
}

void ReadIntVector(std::istream &is, bool binary, std::vector<int32_t> *v) {
  if (binary) {
    const int sz = is.get();
    if (sz != 4) Fail("ReadIntVector: element size " + std::to_string(sz));
    int32_t n;
    is.read(reinterpret_cast<char *>(&n), 4);
    if (n < 0) Fail("ReadIntVector: negative size");
    v->resize(n);
    if (n) is.read(reinterpret_cast<char *>(v->data()), 4 * (size_t)n);
  } else {
    std::string t;
    is >> t;
    if (t != "[") Fail("ReadIntVector: expected [");
    v->clear();
    while (true) {
      is >> std::ws;
      if (is.peek() == ']') { is.get(); break; }
      int32_t x; is >> x;
      if (is.fail()) Fail("ReadIntVector: bad element");
      v->push_back(x);
    }
  }
  if (is.fail()) Fail("ReadIntVector: stream failure");
}

void WriteToken(std::ostream &os, bool, const char *tok) { os << tok << ' '; }
void WriteInt32(std::ostream &os, bool binary, int32_t v) {
  if (binary) { os.put(4); os.write(reinterpret_cast<const char *>(&v), 4); }
  else os << v << ' ';
}

// ------------------------------------------------------------------ CompressedMatrix codec
// compressed-matrix.h:212-239 headers; :347-377 float<->uint16; :457-500 float<->uint8
namespace {
struct CmHeader { float min_value, range; int32_t num_rows, num_cols; };   // on disk (format id is the token)

inline uint16_t FloatToU16(const CmHeader &h, float value) {
  float f = (value - h.min_value) / h.range;
  if (f > 1.0f) f = 1.0f;
  if (f < 0.0f) f = 0.0f;
  return (uint16_t)(int)(f * 65535 + 0.499);
}
inline float U16ToFloat(const CmHeader &h, uint16_t v) { return h.min_value + h.range * 1.52590218966964e-05F * v; }
inline uint8_t FloatToChar(float p0, float p25, float p75, float p100, float value) {
  int ans;
  if (value < p25) {
    float f = (value - p0) / (p25 - p0);
    ans = (int)(f * 64 + 0.5);
    ans = std::min(64, std::max(0, ans));
  } else if (value < p75) {
    float f = (value - p25) / (p75 - p25);
    ans = 64 + (int)(f * 128 + 0.5);
    ans = std::min(192, std::max(64, ans));
  } else {
    float f = (value - p75) / (p100 - p75);
    ans = 192 + (int)(f * 63 + 0.5);
    ans = std::min(255, std::max(192, ans));
  }
  return (uint8_t)ans;
}
inline float CharToFloat(float p0, float p25, float p75, float p100, uint8_t v) {
  if (v <= 64) return (float)(p0 + (p25 - p0) * v * (1 / 64.0));
  if (v <= 192) return (float)(p25 + (p75 - p25) * (v - 64) * (1 / 128.0));
  return (float)(p75 + (p100 - p75) * (v - 192) * (1 / 63.0));
}
}  // namespace

void CompressMatrix(const Matrix &m, std::vector<uint8_t> *blob) {
  if (m.rows <= 0 || m.cols <= 0) Fail("CompressMatrix: empty matrix");
  CmHeader h;
  float mn = m.data[0], mx = m.data[0];
  for (float v : m.data) { mn = std::min(mn, v); mx = std::max(mx, v); }
  if (mx == mn) mx = (float)(mn + (1.0 + std::fabs(mn)));
  if (!(mn - mn == 0) || !(mx - mx == 0)) Fail("CompressMatrix: NaN or inf in matrix");
  h.min_value = mn; h.range = mx - mn; h.num_rows = m.rows; h.num_cols = m.cols;
  blob->assign(sizeof(h) + (size_t)m.cols * (8 + (size_t)m.rows), 0);
  memcpy(blob->data(), &h, sizeof(h));
  uint16_t *pc = reinterpret_cast<uint16_t *>(blob->data() + sizeof(h));
  uint8_t *bytes = blob->data() + sizeof(h) + (size_t)m.cols * 8;
  std::vector<float> col(m.rows);
  for (int c = 0; c < m.cols; c++) {
    for (int r = 0; r < m.rows; r++) col[r] = m.at(r, c);
    std::vector<float> s(col);
    std::sort(s.begin(), s.end());
    uint16_t q0, q25, q75, q100;
    const int n = m.rows;
    auto u16 = [&](float v) { return FloatToU16(h, v); };
    if (n >= 5) {
      const int q = n / 4;
      q0 = std::min<uint16_t>(u16(s[0]), 65532);
      q25 = std::min<uint16_t>(std::max<uint16_t>(u16(s[q]), q0 + 1), 65533);
      q75 = std::min<uint16_t>(std::max<uint16_t>(u16(s[3 * q]), q25 + 1), 65534);
      q100 = std::max<uint16_t>(u16(s[n - 1]), q75 + 1);
    } else {
      q0 = std::min<uint16_t>(u16(s[0]), 65532);
      q25 = n > 1 ? std::min<uint16_t>(std::max<uint16_t>(u16(s[1]), q0 + 1), 65533) : (uint16_t)(q0 + 1);
      q75 = n > 2 ? std::min<uint16_t>(std::max<uint16_t>(u16(s[2]), q25 + 1), 65534) : (uint16_t)(q25 + 1);
      q100 = n > 3 ? std::max<uint16_t>(u16(s[3]), q75 + 1) : (uint16_t)(q75 + 1);
    }
    pc[c * 4 + 0] = q0; pc[c * 4 + 1] = q25; pc[c * 4 + 2] = q75; pc[c * 4 + 3] = q100;
    const float p0 = U16ToFloat(h, q0), p25 = U16ToFloat(h, q25), p75 = U16ToFloat(h, q75), p100 = U16ToFloat(h, q100);
    for (int r = 0; r < n; r++) bytes[(size_t)c * n + r] = FloatToChar(p0, p25, p75, p100, col[r]);
  }
}

void WriteCompressedMatrix(std::ostream &os, const Matrix &m) {
  // CompressedMatrix(mat) with kAutomaticMethod: speech-feature format iff rows > 8, else
  // two-byte (compressed-matrix.cc:60-63)
  os.put('\0'); os.put('B');
  if (m.rows == 0) {  // empty: written as an empty "CM " header? the reference writes "CM " + zero header
    os << "CM ";
    CmHeader h{0, 0, 0, 0};
    os.write(reinterpret_cast<const char *>(&h), sizeof(h));
    return;
  }
  if (m.rows > 8) {
    std::vector<uint8_t> blob;
    CompressMatrix(m, &blob);
    os << "CM ";
    os.write(reinterpret_cast<const char *>(blob.data()), blob.size());
  } else {
    CmHeader h;
    float mn = m.data[0], mx = m.data[0];
    for (float v : m.data) { mn = std::min(mn, v); mx = std::max(mx, v); }
    if (mx == mn) mx = (float)(mn + (1.0 + std::fabs(mn)));
    h.min_value = mn; h.range = mx - mn; h.num_rows = m.rows; h.num_cols = m.cols;
    os << "CM2 ";
    os.write(reinterpret_cast<const char *>(&h), sizeof(h));
    for (float v : m.data) { uint16_t q = FloatToU16(h, v); os.write(reinterpret_cast<const char *>(&q), 2); }
  }
}

static void ReadCompressedBody(std::istream &is, const std::string &tok, Matrix *m) {
  CmHeader h;
  is.read(reinterpret_cast<char *>(&h), sizeof(h));
  if (is.fail()) Fail("compressed matrix: header read failed");
  m->rows = h.num_rows; m->cols = h.num_cols;
  m->data.assign((size_t)h.num_rows * h.num_cols, 0.0f);
  if (h.num_cols == 0) return;
  if (tok == "CM") {
    std::vector<uint16_t> pc((size_t)h.num_cols * 4);
    is.read(reinterpret_cast<char *>(pc.data()), pc.size() * 2);
    std::vector<uint8_t> bytes((size_t)h.num_cols * h.num_rows);
    is.read(reinterpret_cast<char *>(bytes.data()), bytes.size());
    for (int c = 0; c < h.num_cols; c++) {
      const float p0 = U16ToFloat(h, pc[c * 4]), p25 = U16ToFloat(h, pc[c * 4 + 1]), p75 = U16ToFloat(h, pc[c * 4 + 2]),
                  p100 = U16ToFloat(h, pc[c * 4 + 3]);
      for (int r = 0; r < h.num_rows; r++) m->at(r, c) = CharToFloat(p0, p25, p75, p100, bytes[(size_t)c * h.num_rows + r]);
    }
  } else if (tok == "CM2") {
    std::vector<uint16_t> d(m->data.size());
    is.read(reinterpret_cast<char *>(d.data()), d.size() * 2);
    const float inc = h.range * (1.0 / 65535.0);
    for (size_t i = 0; i < d.size(); i++) m->data[i] = h.min_value + d[i] * inc;
  } else {  // CM3
    std::vector<uint8_t> d(m->data.size());
    is.read(reinterpret_cast<char *>(d.data()), d.size());
    const float inc = h.range * (1.0 / 255.0);
    for (size_t i = 0; i < d.size(); i++) m->data[i] = h.min_value + d[i] * inc;
  }
  if (is.fail()) Fail("compressed matrix: data read failed");
}

// ------------------------------------------------------------------ matrices / vectors
static void ReadTextMatrixD(std::istream &is, int32_t *rows, int32_t *cols, std::vector<double> *data) {
  std::string t;
  is >> t;
  if (t != "[") Fail("text matrix: expected [");
  data->clear();
  *rows = 0; *cols = 0;
  int cur = 0;
  std::string line;
  while (true) {
    is >> std::ws;
    const int c = is.peek();
    if (c == ']') {
      is.get();
      if (cur) { if (*rows && cur != *cols) Fail("text matrix: ragged rows"); *cols = cur; (*rows)++; }
      break;
    }
    if (c == EOF) Fail("text matrix: unexpected end");
    double x; is >> x;
    if (is.fail()) Fail("text matrix: bad number");
    data->push_back(x); cur++;
    // a newline ends the row
    while (is.peek() == ' ' || is.peek() == '\t') is.get();
    if (is.peek() == '\n') {
      is.get();
      if (*rows && cur != *cols) Fail("text matrix: ragged rows");
      *cols = cur; (*rows)++; cur = 0;
    }
  }
}

void ReadMatrix(std::istream &is, bool binary, Matrix *m) {
  if (!binary) {
    std::vector<double> d;
    ReadTextMatrixD(is, &m->rows, &m->cols, &d);
    m->data.assign(d.begin(), d.end());
    return;
  }
  const std::string tok = ReadToken(is, true);
  if (tok == "CM" || tok == "CM2" || tok == "CM3") { ReadCompressedBody(is, tok, m); return; }
  if (tok != "FM" && tok != "DM") Fail("ReadMatrix: unexpected token " + tok);
  m->rows = ReadInt32(is, true); m->cols = ReadInt32(is, true);
  const size_t n = (size_t)m->rows * m->cols;
  m->data.resize(n);
  if (tok == "FM") is.read(reinterpret_cast<char *>(m->data.data()), n * 4);
  else {
    std::vector<double> d(n);
    is.read(reinterpret_cast<char *>(d.data()), n * 8);
    for (size_t i = 0; i < n; i++) m->data[i] = (float)d[i];
  }
  if (is.fail()) Fail("ReadMatrix: data read failed");
}

void ReadMatrixD(std::istream &is, bool binary, MatrixD *m) {
  if (!binary) { ReadTextMatrixD(is, &m->rows, &m->cols, &m->data); return; }
  const std::string tok = ReadToken(is, true);
  if (tok != "FM" && tok != "DM") Fail("ReadMatrixD: unexpected token " + tok);
  m->rows = ReadInt32(is, true); m->cols = ReadInt32(is, true);
  const size_t n = (size_t)m->rows * m->cols;
  m->data.resize(n);
  if (tok == "DM") is.read(reinterpret_cast<char *>(m->data.data()), n * 8);
  else {
    std::vector<float> d(n);
    is.read(reinterpret_cast<char *>(d.data()), n * 4);
    for (size_t i = 0; i < n; i++) m->data[i] = d[i];
  }
  if (is.fail()) Fail("ReadMatrixD: data read failed");
}

void ReadVector(std::istream &is, bool binary, std::vector<float> *v) {
  if (!binary) {
    std::string t; is >> t;
    if (t != "[") Fail("text vector: expected [");
    v->clear();
    while (true) {
      is >> std::ws;
      if (is.peek() == ']') { is.get(); break; }
      float x; is >> x;
      if (is.fail()) Fail("text vector: bad number");
      v->push_back(x);
    }
    return;
  }
  const std::string tok = ReadToken(is, true);
  if (tok != "FV" && tok != "DV") Fail("ReadVector: unexpected token " + tok);
  const int32_t n = ReadInt32(is, true);
  v->resize(n);
  if (tok == "FV") is.read(reinterpret_cast<char *>(v->data()), (size_t)n * 4);
  else {
    std::vector<double> d(n);
    is.read(reinterpret_cast<char *>(d.data()), (size_t)n * 8);
    for (int i = 0; i < n; i++) (*v)[i] = (float)d[i];
  }
  if (is.fail()) Fail("ReadVector: data read failed");
}

void WriteMatrix(std::ostream &os, bool binary, const Matrix &m) {
  if (binary) {
    os.put('\0'); os.put('B');
    os << "FM ";
    WriteInt32(os, true, m.rows); WriteInt32(os, true, m.cols);
    os.write(reinterpret_cast<const char *>(m.data.data()), m.data.size() * 4);
  } else {
    if (m.cols == 0) { os << " [ ]\n"; return; }
    os << " [";
    for (int r = 0; r < m.rows; r++) {
      os << "\n  ";
      for (int c = 0; c < m.cols; c++) os << m.at(r, c) << " ";
    }
    os << "]\n";
  }
}

void WriteMatrixD(std::ostream &os, bool binary, const MatrixD &m) {
  if (binary) {
    os.put('\0'); os.put('B');
    os << "DM ";
    WriteInt32(os, true, m.rows); WriteInt32(os, true, m.cols);
    os.write(reinterpret_cast<const char *>(m.data.data()), m.data.size() * 8);
  } else {
    os.precision(17);
    os << " [";
    for (int r = 0; r < m.rows; r++) {
      os << "\n  ";
      for (int c = 0; c < m.cols; c++) os << m.data[(size_t)r * m.cols + c] << " ";
    }
    os << "]\n";
  }
}

// ------------------------------------------------------------------ tables
static std::vector<std::string> Split(const std::string &s, char sep) {
  std::vector<std::string> out;
  std::string cur;
  for (char c : s) { if (c == sep) { out.push_back(cur); cur.clear(); } else cur += c; }
  out.push_back(cur);
  return out;
}

Rspec ParseRspecifier(const std::string &s) {
  const size_t p = s.find(':');
  if (p == std::string::npos) Fail("bad rspecifier: " + s);
  Rspec r;
  bool ok = false;
  for (const std::string &o : Split(s.substr(0, p), ',')) {
    if (o == "ark") ok = true;
    else if (o == "scp") { r.is_scp = true; ok = true; }
    else if (o == "t") r.text = true;
    // s, cs, p, o, ns, ncs, np, b: ordering/permissive hints with no effect on the content
  }
  if (!ok) Fail("bad rspecifier: " + s);
  r.path = s.substr(p + 1);
  return r;
}

Wspec ParseWspecifier(const std::string &s) {
  const size_t p = s.find(':');
  if (p == std::string::npos) Fail("bad wspecifier: " + s);
  Wspec w;
  std::vector<std::string> kinds;
  for (const std::string &o : Split(s.substr(0, p), ',')) {
    if (o == "ark" || o == "scp") kinds.push_back(o);
    else if (o == "t") w.text = true;
  }
  const std::string rest = s.substr(p + 1);
  if (kinds.size() == 1 && kinds[0] == "ark") { w.has_ark = true; w.ark_path = rest; }
  else if (kinds.size() == 2 && kinds[0] == "ark" && kinds[1] == "scp") {
    const size_t c = rest.find(',');
    if (c == std::string::npos) Fail("bad wspecifier: " + s);
    w.has_ark = true; w.has_scp = true; w.ark_path = rest.substr(0, c); w.scp_path = rest.substr(c + 1);
  } else Fail("unsupported wspecifier: " + s);
  return w;
}

std::vector<ScpEntry> ReadScp(const std::string &path) {
  std::ifstream f(path);
  if (!f) Fail("cannot open script file " + path);
  std::vector<ScpEntry> out;
  std::string line;
  while (std::getline(f, line)) {
    if (line.empty()) continue;
    const size_t sp = line.find_first_of(" \t");
    if (sp == std::string::npos) Fail("bad scp line: " + line);
    ScpEntry e;
    e.key = line.substr(0, sp);
    std::string rest = line.substr(line.find_first_not_of(" \t", sp));
    while (!rest.empty() && (rest.back() == '\r' || rest.back() == ' ')) rest.pop_back();
    const size_t colon = rest.rfind(':');
    if (colon != std::string::npos && colon + 1 < rest.size() &&
        rest.find_first_not_of("0123456789", colon + 1) == std::string::npos && colon > 1) {
      e.offset = std::stoll(rest.substr(colon + 1));
      e.path = rest.substr(0, colon);
    } else e.path = rest;
    out.push_back(e);
  }
  return out;
}

static std::string ReadArkKey(std::istream &is) {
  std::string key;
  is >> std::ws;
  if (is.peek() == EOF) return "";
  is >> key;
  if (is.fail()) return "";
  if (is.peek() != ' ' && is.peek() != '\t' && is.peek() != '\n') Fail("archive: expected space after key " + key);
  is.get();
  return key;
}

MatrixReader::MatrixReader(const std::string &rspecifier) : spec_(ParseRspecifier(rspecifier)) {
  if (spec_.is_scp) scp_ = ReadScp(spec_.path);
  else {
    ark_.open(spec_.path, std::ios::binary);
    if (!ark_) Fail("cannot open archive " + spec_.path);
  }
}

bool MatrixReader::Next(std::string *key, Matrix *m) {
  if (spec_.is_scp) {
    if (pos_ >= scp_.size()) return false;
    const ScpEntry &e = scp_[pos_++];
    std::ifstream f(e.path, std::ios::binary);
    if (!f) Fail("cannot open " + e.path);
    if (e.offset >= 0) f.seekg(e.offset);
    const bool binary = ReadBinaryHeader(f);
    ReadMatrix(f, binary, m);
    *key = e.key;
    return true;
  }
  *key = ReadArkKey(ark_);
  if (key->empty()) return false;
  const bool binary = ReadBinaryHeader(ark_);
  ReadMatrix(ark_, binary, m);
  return true;
}

void ReadMatricesInOrder(const std::string &rspecifier, int threads, std::vector<std::string> *keys, std::vector<Matrix> *mats) {
  keys->clear(); mats->clear();
  const Rspec spec = ParseRspecifier(rspecifier);
  if (!spec.is_scp || threads <= 1) {
    MatrixReader r(rspecifier);
    std::string k; Matrix m;
    while (r.Next(&k, &m)) { keys->push_back(k); mats->push_back(std::move(m)); m = Matrix(); }
    return;
  }
  const std::vector<ScpEntry> scp = ReadScp(spec.path);
  const size_t n = scp.size();
  keys->resize(n); mats->resize(n);
  const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)threads, n / 8 + 1));
  std::vector<std::string> err((size_t)nt);
  auto work = [&](int t) {
    try {
      std::ifstream f; std::string open_path;
      for (size_t i = n * (size_t)t / (size_t)nt; i < n * (size_t)(t + 1) / (size_t)nt; i++) {
        const ScpEntry &e = scp[i];
        if (!f.is_open() || open_path != e.path) {       // (the entries of a job usually sit in one archive)
          f.close(); f.clear();
          f.open(e.path, std::ios::binary);
          if (!f) Fail("cannot open " + e.path);
          open_path = e.path;
        }
        f.clear();
        f.seekg(e.offset >= 0 ? e.offset : 0);
        const bool binary = ReadBinaryHeader(f);
        ReadMatrix(f, binary, &(*mats)[i]);
        (*keys)[i] = e.key;
      }
    } catch (const std::exception &ex) { err[(size_t)t] = ex.what(); if (err[(size_t)t].empty()) err[(size_t)t] = "error"; }
  };
  std::vector<std::thread> th;
  for (int t = 1; t < nt; t++) th.emplace_back(work, t);
  work(0);
  for (std::thread &x : th) x.join();
  for (const std::string &e : err) if (!e.empty()) Fail(e);
}

std::map<std::string, Matrix> ReadMatrixTable(const std::string &rspecifier) {
  std::map<std::string, Matrix> out;
  MatrixReader r(rspecifier);
  std::string k; Matrix m;
  while (r.Next(&k, &m)) out[k] = m;
  return out;
}

std::map<std::string, MatrixD> ReadMatrixDTable(const std::string &rspecifier) {
  std::map<std::string, MatrixD> out;
  const Rspec spec = ParseRspecifier(rspecifier);
  auto read_one = [&](std::istream &is, const std::string &key) {
    const bool binary = ReadBinaryHeader(is);
    MatrixD m; ReadMatrixD(is, binary, &m);
    out[key] = m;
  };
  if (spec.is_scp) {
    for (const ScpEntry &e : ReadScp(spec.path)) {
      std::ifstream f(e.path, std::ios::binary);
      if (!f) Fail("cannot open " + e.path);
      if (e.offset >= 0) f.seekg(e.offset);
      read_one(f, e.key);
    }
  } else {
    std::ifstream f(spec.path, std::ios::binary);
    if (!f) Fail("cannot open archive " + spec.path);
    while (true) {
      const std::string key = ReadArkKey(f);
      if (key.empty()) break;
      read_one(f, key);
    }
  }
  return out;
}

std::map<std::string, std::string> ReadStringMap(const std::string &rspecifier) {
  const Rspec spec = ParseRspecifier(rspecifier);
  std::ifstream f(spec.path);
  if (!f) Fail("cannot open " + spec.path);
  std::map<std::string, std::string> out;
  std::string line;
  while (std::getline(f, line)) {
    std::istringstream ss(line);
    std::string k, v;
    if (ss >> k >> v) out[k] = v;
  }
  return out;
}

MatrixWriter::MatrixWriter(const std::string &wspecifier, bool compress) : spec_(ParseWspecifier(wspecifier)), compress_(compress) {
  ark_.open(spec_.ark_path, std::ios::binary);
  if (!ark_) Fail("cannot open " + spec_.ark_path + " for writing");
  if (spec_.has_scp) {
    scp_.open(spec_.scp_path);
    if (!scp_) Fail("cannot open " + spec_.scp_path + " for writing");
  }
}

std::string SerializeMatrix(const Matrix &m, bool text, bool compress) {
  std::ostringstream os;
  if (text) WriteMatrix(os, false, m);
  else if (compress) WriteCompressedMatrix(os, m);
  else WriteMatrix(os, true, m);
  return os.str();
}

void MatrixWriter::Write(const std::string &key, const Matrix &m) { WriteSerialized(key, SerializeMatrix(m, spec_.text, compress_)); }

void MatrixWriter::WriteSerialized(const std::string &key, const std::string &object) {
  ark_ << key << ' ';
  const int64_t off = (int64_t)ark_.tellp();
  ark_.write(object.data(), (std::streamsize)object.size());
  if (spec_.has_scp) scp_ << key << ' ' << spec_.ark_path << ':' << off << '\n';
  if (ark_.fail()) Fail("write failure on " + spec_.ark_path);
}

void MatrixWriter::Close() { ark_.close(); if (spec_.has_scp) scp_.close(); }

void IntVectorWriter::Open(const std::string &wspecifier) {
  spec_ = ParseWspecifier(wspecifier);
  ark_.open(spec_.ark_path, std::ios::binary);
  if (!ark_) Fail("cannot open " + spec_.ark_path + " for writing");
  open_ = true;
}

void IntVectorWriter::Write(const std::string &key, const std::vector<int32_t> &v) {
  ark_ << key << ' ';
  if (spec_.text) {
    for (int32_t x : v) ark_ << x << ' ';
    ark_ << '\n';
  } else {
    ark_.put('\0'); ark_.put('B');
    ark_.put(4);
    const int32_t n = (int32_t)v.size();
    ark_.write(reinterpret_cast<const char *>(&n), 4);
    if (n) ark_.write(reinterpret_cast<const char *>(v.data()), 4 * (size_t)n);
  }
}

// ------------------------------------------------------------------ wave (wave-reader.cc:137-310)
void ReadWave(std::istream &is, Wave *w) {
  auto rd32 = [&]() { uint32_t v = 0; is.read(reinterpret_cast<char *>(&v), 4); return v; };
  auto rd16 = [&]() { uint16_t v = 0; is.read(reinterpret_cast<char *>(&v), 2); return v; };
  char tag[5] = {0};
  is.read(tag, 4);
  if (std::string(tag) != "RIFF") Fail("wave: expected RIFF");
  rd32();
  is.read(tag, 4);
  if (std::string(tag) != "WAVE") Fail("wave: expected WAVE");
  bool have_fmt = false;
  uint16_t fmt = 0, channels = 0, bits = 0, block_align = 0;
  uint32_t rate = 0;
  while (true) {
    is.read(tag, 4);
    if (is.fail()) Fail("wave: no data chunk");
    uint32_t size = rd32();
    const std::string t(tag, 4);
    if (t == "fmt ") {
      fmt = rd16(); channels = rd16(); rate = rd32(); rd32(); block_align = rd16(); bits = rd16();
      uint32_t used = 16;
      if (fmt == 0xFFFE && size >= 40) {   // WAVE_FORMAT_EXTENSIBLE: sub-format GUID starts with the real tag
        rd16(); rd16(); rd32(); fmt = rd16(); used = 26;
        for (int i = 0; i < 14; i++) is.get();
        used = 40;
      }
      for (uint32_t i = used; i < size; i++) is.get();
      have_fmt = true;
    } else if (t == "data") {
      if (!have_fmt) Fail("wave: data before fmt");
      if (fmt != 1) Fail("wave: can read only PCM data, format tag " + std::to_string(fmt));
      if (bits != 16) Fail("wave: unsupported bits per sample " + std::to_string(bits));
      if (channels == 0) Fail("wave: no channels");
      std::vector<char> raw;
      if (size == 0xFFFFFFFFu || size == 0) {   // streamed: read to the end
        raw.assign(std::istreambuf_iterator<char>(is), std::istreambuf_iterator<char>());
      } else {
        raw.resize(size);
        is.read(raw.data(), size);
        raw.resize((size_t)is.gcount());
      }
      const size_t frame = (size_t)channels * 2;
      const size_t n = raw.size() / frame;
      w->samp_freq = (float)rate; w->channels = channels;
      w->data.assign(channels, std::vector<float>(n));
      const int16_t *p = reinterpret_cast<const int16_t *>(raw.data());
      for (size_t i = 0; i < n; i++)
        for (int c = 0; c < channels; c++) w->data[c][i] = (float)p[i * channels + c];
      (void)block_align;
      return;
    } else {
      for (uint32_t i = 0; i < size + (size & 1); i++) is.get();
    }
  }
}

void ReadWaveFile(const std::string &path, int64_t offset, Wave *w) {
  std::ifstream f(path, std::ios::binary);
  if (!f) Fail("cannot open wave file " + path);
  if (offset >= 0) f.seekg(offset);
  ReadWave(f, w);
}

// ------------------------------------------------------------------ model
namespace {
struct HmmState { int32_t fwd, self; std::vector<std::pair<int32_t, float>> trans; };
}

void ReadModel(const std::string &path, Model *m) {
  std::ifstream is(path, std::ios::binary);
  if (!is) Fail("cannot open model " + path);
  const bool binary = ReadBinaryHeader(is);
  // ---- TransitionModel (transition-model.cc:383-409)
  ExpectToken(is, binary, "<TransitionModel>");
  ExpectToken(is, binary, "<Topology>");
  std::vector<int32_t> phones, phone2idx;
  std::vector<std::vector<HmmState>> entries;
  if (!binary) {
    // the text form of a topology is its own little language (hmm-topology.cc:41-127): <TopologyEntry> <ForPhones> p p .. </ForPhones>
    // <State> i [<PdfClass> c | <ForwardPdfClass> c <SelfLoopPdfClass> c] {<Transition> dst prob} </State> ... </TopologyEntry>
    for (;;) {
      std::string tok = ReadToken(is, false);
      if (tok == "</Topology>") break;
      if (tok != "<TopologyEntry>") Fail("model: expected </Topology> or <TopologyEntry>, got " + tok);
      ExpectToken(is, false, "<ForPhones>");
      std::vector<int32_t> ph;
      for (;;) {
        const std::string w = ReadToken(is, false);
        if (w == "</ForPhones>") break;
        char *end = nullptr;
        const long v = strtol(w.c_str(), &end, 10);
        if (end == w.c_str() || *end || v <= 0) Fail("model: expected a phone, got " + w);
        ph.push_back((int32_t)v);
      }
      std::vector<HmmState> entry;
      tok = ReadToken(is, false);
      while (tok != "</TopologyEntry>") {
        if (tok != "<State>") Fail("model: expected </TopologyEntry> or <State>, got " + tok);
        if (ReadInt32(is, false) != (int32_t)entry.size()) Fail("model: topology states are expected in order from zero");
        HmmState st; st.fwd = -1; st.self = -1;
        tok = ReadToken(is, false);
        if (tok == "<PdfClass>") { st.fwd = st.self = ReadInt32(is, false); tok = ReadToken(is, false); }
        else if (tok == "<ForwardPdfClass>") {
          st.fwd = ReadInt32(is, false);
          ExpectToken(is, false, "<SelfLoopPdfClass>");
          st.self = ReadInt32(is, false);
          tok = ReadToken(is, false);
        }
        while (tok == "<Transition>") {
          const int32_t dst = ReadInt32(is, false);
          st.trans.emplace_back(dst, ReadFloat(is, false));
          tok = ReadToken(is, false);
        }
        if (tok != "</State>") Fail("model: unexpected token in the topology: " + tok);
        entry.push_back(std::move(st));
        tok = ReadToken(is, false);
      }
      for (int32_t p : ph) {
        if ((int32_t)phone2idx.size() <= p) phone2idx.resize(p + 1, -1);
        if (phone2idx[p] != -1) Fail("model: phone " + std::to_string(p) + " appears in several topology entries");
        phone2idx[p] = (int32_t)entries.size();
        phones.push_back(p);
      }
      entries.push_back(std::move(entry));
    }
  }
  int32_t sz = 0;
  bool is_hmm = true;
  if (binary) {
    ReadIntVector(is, binary, &phones);
    ReadIntVector(is, binary, &phone2idx);
    sz = ReadInt32(is, binary);
    if (sz == -1) { is_hmm = false; sz = ReadInt32(is, binary); }
    entries.resize(sz);
  }
  for (int32_t i = 0; i < sz; i++) {
    const int32_t n = ReadInt32(is, binary);
    entries[i].resize(n);
    for (int32_t j = 0; j < n; j++) {
      entries[i][j].fwd = ReadInt32(is, binary);
      entries[i][j].self = is_hmm ? entries[i][j].fwd : ReadInt32(is, binary);
      const int32_t nt = ReadInt32(is, binary);
      entries[i][j].trans.resize(nt);
      for (int32_t k = 0; k < nt; k++) {
        entries[i][j].trans[k].first = ReadInt32(is, binary);
        entries[i][j].trans[k].second = ReadFloat(is, binary);
      }
    }
  }
  if (binary) ExpectToken(is, binary, "</Topology>");
  const std::string tok = ReadToken(is, binary);
  if (tok != "<Triples>" && tok != "<Tuples>") Fail("model: expected <Triples> or <Tuples>, got " + tok);
  const int32_t ntup = ReadInt32(is, binary);
  struct Tuple { int32_t phone, hmm_state, fwd, self; };
  std::vector<Tuple> tuples(ntup);
  for (int32_t i = 0; i < ntup; i++) {
    tuples[i].phone = ReadInt32(is, binary);
    tuples[i].hmm_state = ReadInt32(is, binary);
    tuples[i].fwd = ReadInt32(is, binary);
    tuples[i].self = (tok == "<Tuples>") ? ReadInt32(is, binary) : tuples[i].fwd;
  }
  const std::string endtok = ReadToken(is, binary);
  if (endtok != "</Triples>" && endtok != "</Tuples>") Fail("model: bad tuple terminator " + endtok);
  // ComputeDerived (transition-model.cc:144-177): transition-ids in tuple order, one per
  // transition of the tuple's HMM state; a self-loop uses the self-loop pdf
  m->tid2pdf.assign(1, -1);
  m->trans.phone.assign(1, 0); m->trans.hmm_state.assign(1, 0); m->trans.self_loop.assign(1, 0);
  m->tid2tstate.assign(1, 0);
  int32_t tstate = 0;
  for (const Tuple &t : tuples) {
    tstate++;
    if (t.phone < 0 || t.phone >= (int32_t)phone2idx.size() || phone2idx[t.phone] < 0) Fail("model: tuple phone without topology");
    const std::vector<HmmState> &e = entries[phone2idx[t.phone]];
    if (t.hmm_state < 0 || t.hmm_state >= (int32_t)e.size()) Fail("model: bad hmm state in tuple");
    for (const auto &tr : e[t.hmm_state].trans) {
      m->tid2pdf.push_back(tr.first == t.hmm_state ? t.self : t.fwd);
      m->trans.phone.push_back(t.phone);
      m->trans.hmm_state.push_back(t.hmm_state);
      m->trans.self_loop.push_back(tr.first == t.hmm_state ? 1 : 0);
      m->tid2tstate.push_back(tstate);
    }
  }
  ExpectToken(is, binary, "<LogProbs>");
  std::vector<float> logprobs;
  ReadVector(is, binary, &logprobs);
  ExpectToken(is, binary, "</LogProbs>");
  if (logprobs.size() != m->tid2pdf.size()) Fail("model: <LogProbs> has " + std::to_string(logprobs.size()) + " entries for " + std::to_string(m->tid2pdf.size() - 1) + " transition-ids");
  m->log_probs = logprobs;
  // ComputeDerivedOfProbs (transition-model.cc:364-381): log(1 - self-loop probability) per transition-state, 0 without a self-loop
  m->non_self_loop_log_probs.assign((size_t)tstate + 1, 0.0f);
  for (size_t tid = 1; tid < m->tid2pdf.size(); tid++) {
    if (!m->trans.self_loop[tid]) continue;
    const float self_loop_prob = expf(m->log_probs[tid]);
    float non_self_loop_prob = 1.0 - self_loop_prob;
    if (non_self_loop_prob <= 0.0) non_self_loop_prob = 1.0e-10;
    m->non_self_loop_log_probs[m->tid2tstate[tid]] = logf(non_self_loop_prob);
  }
  ExpectToken(is, binary, "</TransitionModel>");
  // ---- AmDiagGmm (am-diag-gmm.cc:147-161)
  ExpectToken(is, binary, "<DIMENSION>");
  m->dim = ReadInt32(is, binary);
  ExpectToken(is, binary, "<NUMPDFS>");
  m->num_pdfs = ReadInt32(is, binary);
  m->pdf_offsets.assign(1, 0);
  m->weights.clear(); m->means_invvars.clear(); m->inv_vars.clear(); m->gconsts_in_file.clear();
  for (int32_t p = 0; p < m->num_pdfs; p++) {
    std::string t = ReadToken(is, binary);
    if (t != "<DiagGMMBegin>" && t != "<DiagGMM>") Fail("model: expected <DiagGMM>, got " + t);
    t = ReadToken(is, binary);
    std::vector<float> gc, w;
    if (t == "<GCONSTS>") { ReadVector(is, binary, &gc); t = ReadToken(is, binary); }
    if (t != "<WEIGHTS>") Fail("model: expected <WEIGHTS>, got " + t);
    ReadVector(is, binary, &w);
    ExpectToken(is, binary, "<MEANS_INVVARS>");
    Matrix miv; ReadMatrix(is, binary, &miv);
    ExpectToken(is, binary, "<INV_VARS>");
    Matrix iv; ReadMatrix(is, binary, &iv);
    t = ReadToken(is, binary);
    if (t != "<DiagGMMEnd>" && t != "</DiagGMM>") Fail("model: expected </DiagGMM>, got " + t);
    if (miv.cols != m->dim || iv.cols != m->dim || miv.rows != (int32_t)w.size() || iv.rows != miv.rows) Fail("model: inconsistent GMM sizes");
    m->weights.insert(m->weights.end(), w.begin(), w.end());
    m->means_invvars.insert(m->means_invvars.end(), miv.data.begin(), miv.data.end());
    m->inv_vars.insert(m->inv_vars.end(), iv.data.begin(), iv.data.end());
    m->gconsts_in_file.insert(m->gconsts_in_file.end(), gc.begin(), gc.end());
    m->pdf_offsets.push_back(m->pdf_offsets.back() + (int32_t)w.size());
  }
}

float ScaledTransitionLogProb(const Model &m, int32_t tid, float transition_scale, float self_loop_scale) {
  if (transition_scale == self_loop_scale) return m.log_probs[tid] * transition_scale;
  if (m.trans.self_loop[tid]) return self_loop_scale * m.log_probs[tid];
  const float nsl = m.non_self_loop_log_probs[m.tid2tstate[tid]];
  // GetTransitionLogProbIgnoringSelfLoops (:457-461) = log_probs_(tid) - non-self-loop log prob of its transition-state
  return self_loop_scale * nsl + transition_scale * (m.log_probs[tid] - nsl);
}

// ------------------------------------------------------------------ OpenFst binary
namespace {
std::string ReadFstString(std::istream &is) {
  int32_t n = 0;
  is.read(reinterpret_cast<char *>(&n), 4);
  if (n < 0 || n > 1024) Fail("fst: bad string length");
  std::string s(n, ' ');
  if (n) is.read(&s[0], n);
  return s;
}
constexpr int32_t kFstMagic = 2125659606;
}  // namespace

// A decoding graph is tens to hundreds of megabytes of fixed-size records: a const FST in a file is mapped and its arc array copied
// out in one piece (no zero-filled vector first, no stream buffering); everything else goes through the stream reader.
static bool ReadConstFstMapped(const std::string &path, Graph *g) {
  const int fd = open(path.c_str(), O_RDONLY);
  if (fd < 0) return false;
  struct stat sb;
  if (fstat(fd, &sb) != 0 || sb.st_size < 64) { close(fd); return false; }
  const size_t len = (size_t)sb.st_size;
  void *mp = mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  if (mp == MAP_FAILED) return false;
  const char *b = static_cast<const char *>(mp);
  size_t pos = 0;
  bool ok = false;
  auto rd = [&](void *dst, size_t n) -> bool { if (pos + n > len) return false; memcpy(dst, b + pos, n); pos += n; return true; };
  auto rd_str = [&](std::string *o) -> bool { int32_t n; if (!rd(&n, 4) || n < 0 || pos + (size_t)n > len) return false; o->assign(b + pos, (size_t)n); pos += (size_t)n; return true; };
  do {
    int32_t magic, version, flags; uint64_t props; int64_t start, nstates, narcs;
    std::string fsttype, arctype;
    if (!rd(&magic, 4) || magic != kFstMagic || !rd_str(&fsttype) || !rd_str(&arctype)) break;
    if (fsttype != "const" || arctype != "standard") break;
    if (!rd(&version, 4) || !rd(&flags, 4) || !rd(&props, 8) || !rd(&start, 8) || !rd(&nstates, 8) || !rd(&narcs, 8)) break;
    if ((flags & 3) || nstates < 0 || narcs < 0) break;
    const bool aligned = (flags & 4) || version == 1;
    if (aligned) pos = (pos + 15) & ~(size_t)15;
    struct ConstState { float final; uint32_t pos, narcs, niepsilons, noepsilons; };
    if (pos + sizeof(ConstState) * (size_t)nstates > len) break;
    const ConstState *st = reinterpret_cast<const ConstState *>(b + pos);      // (20-byte records at a 4-byte-aligned offset)
    pos += sizeof(ConstState) * (size_t)nstates;
    if (aligned) pos = (pos + 15) & ~(size_t)15;
    if (pos + sizeof(Arc) * (size_t)narcs > len) break;
    const Arc *arcs = reinterpret_cast<const Arc *>(b + pos);
    g->start = (int32_t)start;
    g->arcs.assign(arcs, arcs + narcs);
    g->final_cost.resize(nstates); g->arc_offset.resize(nstates + 1);
    for (int64_t s = 0; s < nstates; s++) { g->final_cost[s] = st[s].final; g->arc_offset[s] = st[s].pos; }
    g->arc_offset[nstates] = narcs;
    ok = true;
  } while (false);
  munmap(mp, len);
  return ok;
}

void ReadFst(const std::string &path, Graph *g) {
  if (ReadConstFstMapped(path, g)) return;      // (anything else -- vector FSTs, unsupported or damaged files -- gets the stream reader and its messages)
  std::ifstream is(path, std::ios::binary);
  if (!is) Fail("cannot open fst " + path);
  ReadFstStream(is, g, path);
}

FstTableReader::FstTableReader(const std::string &rspecifier) : spec_(ParseRspecifier(rspecifier)) {
  if (spec_.is_scp) Fail("graph tables are read from archives (ark:), not from " + rspecifier);
  ark_.open(spec_.path, std::ios::binary);
  if (!ark_) Fail("cannot open " + spec_.path);
}
bool FstTableReader::Next(std::string *key, Graph *g) {
  while (ark_.peek() == '\n' || ark_.peek() == ' ') ark_.get();
  if (ark_.peek() == EOF) return false;
  key->clear();
  int c;
  while ((c = ark_.get()) != EOF && c != ' ') key->push_back((char)c);
  if (c == EOF) Fail("graph table: truncated entry for key " + *key);
  if (isspace(ark_.peek())) Fail("graph table: text-form FSTs are not supported (key " + *key + "): write the table with ark:, not ark,t:");
  ReadFstStream(ark_, g, "graph of " + *key);
  return true;
}

void FloatWriter::Open(const std::string &wspecifier) {
  spec_ = ParseWspecifier(wspecifier);
  ark_.open(spec_.ark_path, std::ios::binary);
  if (!ark_) Fail("cannot open " + spec_.ark_path + " for writing");
  open_ = true;
}
void FloatWriter::Write(const std::string &key, float v) {
  ark_ << key << ' ';
  if (spec_.text) { if (ark_.precision() < 7) ark_.precision(7); ark_ << v << " \n"; }      // (WriteBasicType + BasicHolder: value, space, newline)
  else { ark_.put('\0'); ark_.put('B'); ark_.put(4); ark_.write(reinterpret_cast<const char *>(&v), 4); }
}
void FloatVectorWriter::Open(const std::string &wspecifier) {
  spec_ = ParseWspecifier(wspecifier);
  ark_.open(spec_.ark_path, std::ios::binary);
  if (!ark_) Fail("cannot open " + spec_.ark_path + " for writing");
  open_ = true;
}
void FloatVectorWriter::Write(const std::string &key, const std::vector<float> &v) {
  ark_ << key << ' ';
  if (spec_.text) {
    if (ark_.precision() < 7) ark_.precision(7);
    ark_ << " [ ";
    for (float x : v) ark_ << x << ' ';
    ark_ << "]\n";
  } else {
    ark_.put('\0'); ark_.put('B');
    ark_.write("FV ", 3);
    ark_.put(4);
    const int32_t n = (int32_t)v.size();
    ark_.write(reinterpret_cast<const char *>(&n), 4);
    if (n) ark_.write(reinterpret_cast<const char *>(v.data()), 4 * (size_t)n);
  }
}

void ReadFstStream(std::istream &is, Graph *g, const std::string &path) {
  int32_t magic = 0;
  is.read(reinterpret_cast<char *>(&magic), 4);
  if (magic != kFstMagic) Fail("fst: bad magic number in " + path);
  const std::string fsttype = ReadFstString(is), arctype = ReadFstString(is);
  int32_t version, flags;
  uint64_t props;
  int64_t start, nstates, narcs;
  is.read(reinterpret_cast<char *>(&version), 4);
  is.read(reinterpret_cast<char *>(&flags), 4);
  is.read(reinterpret_cast<char *>(&props), 8);
  is.read(reinterpret_cast<char *>(&start), 8);
  is.read(reinterpret_cast<char *>(&nstates), 8);
  is.read(reinterpret_cast<char *>(&narcs), 8);
  if (is.fail()) Fail("fst: truncated header");
  if (arctype != "standard") Fail("fst: arc type " + arctype + " not supported (need standard)");
  if (flags & 3) Fail("fst: embedded symbol tables are not supported");
  g->start = (int32_t)start;
  if (fsttype == "const") {
    const bool aligned = (flags & 4) || version == 1;
    auto align = [&]() { while (is.tellg() % 16 != 0) is.get(); };
    if (aligned) align();
    struct ConstState { float final; uint32_t pos, narcs, niepsilons, noepsilons; };
    std::vector<ConstState> st(nstates);
    is.read(reinterpret_cast<char *>(st.data()), sizeof(ConstState) * (size_t)nstates);
    if (aligned) align();
    g->arcs.resize(narcs);
    is.read(reinterpret_cast<char *>(g->arcs.data()), sizeof(Arc) * (size_t)narcs);
    if (is.fail()) Fail("fst: truncated const fst");
    g->final_cost.resize(nstates); g->arc_offset.resize(nstates + 1);
    for (int64_t s = 0; s < nstates; s++) { g->final_cost[s] = st[s].final; g->arc_offset[s] = st[s].pos; }
    g->arc_offset[nstates] = narcs;
  } else if (fsttype == "vector") {
    g->final_cost.clear(); g->arc_offset.assign(1, 0); g->arcs.clear();
    for (int64_t s = 0; s < nstates; s++) {
      float fin; int64_t na;
      is.read(reinterpret_cast<char *>(&fin), 4);
      is.read(reinterpret_cast<char *>(&na), 8);
      if (is.fail() || na < 0) Fail("fst: truncated vector fst");
      g->final_cost.push_back(fin);
      const size_t old = g->arcs.size();
      g->arcs.resize(old + na);
      if (na) is.read(reinterpret_cast<char *>(g->arcs.data() + old), sizeof(Arc) * (size_t)na);
      g->arc_offset.push_back((int64_t)g->arcs.size());
    }
    if (is.fail()) Fail("fst: truncated vector fst");
  } else Fail("fst: type " + fsttype + " not supported (need const or vector)");
}

namespace {
std::mutex g_mem_mu;
std::map<std::string, std::shared_ptr<MemTable>> g_mem_tables;
std::map<std::string, std::shared_ptr<const FlatPostTable>> g_mem_posts;      // posterior tables published as objects
}  // namespace
bool IsMemPath(const std::string &path) { return path.compare(0, 4, "mem:") == 0; }
void MemPublish(const std::string &path, std::shared_ptr<MemTable> t) { std::lock_guard<std::mutex> lk(g_mem_mu); g_mem_tables[path] = std::move(t); }
std::shared_ptr<MemTable> MemLookup(const std::string &path) {
  std::lock_guard<std::mutex> lk(g_mem_mu);
  auto it = g_mem_tables.find(path);
  return it == g_mem_tables.end() ? nullptr : it->second;
}
void MemUnlink(const std::string &path) { std::lock_guard<std::mutex> lk(g_mem_mu); g_mem_tables.erase(path); g_mem_posts.erase(path); }

void LatticeWriter::Open(const std::string &wspecifier) {
  spec_ = ParseWspecifier(wspecifier);
  if (IsMemPath(spec_.ark_path)) {
    if (spec_.text || spec_.has_scp) Fail("an in-memory table is binary and has no script file: " + wspecifier);
    mem_ = std::make_shared<MemTable>();
    open_ = true;
    return;
  }
  ark_.open(spec_.ark_path, std::ios::binary);
  if (!ark_) Fail("cannot open " + spec_.ark_path + " for writing");
  open_ = true;
}
LatticeWriter::~LatticeWriter() { if (mem_) MemPublish(spec_.ark_path, std::move(mem_)); }

// Lattices are serialised into one buffer per entry and parsed from a mapping of the whole archive: an entry has millions of
// 4-byte fields, and a stream call per field costs more than everything else the lattice programs do.
namespace {
static_assert(sizeof(LatticeArcOut) == 20 && sizeof(vblat::Arc) == 20, "arc records are written as they lie in memory");
template <class T> void Put(std::string *b, const T &v) { b->append(reinterpret_cast<const char *>(&v), sizeof(T)); }
void PutFstString(std::string *b, const char *s) { Put<int32_t>(b, (int32_t)strlen(s)); b->append(s); }
void PutFstHeader(std::string *b, const char *arctype, int64_t start, int64_t ns, int64_t na) {
  Put(b, kFstMagic);
  PutFstString(b, "vector");
  PutFstString(b, arctype);
  Put<int32_t>(b, 2); Put<int32_t>(b, 0);
  Put<uint64_t>(b, 3);          // kExpanded | kMutable; everything else unknown
  Put(b, start); Put(b, ns); Put(b, na);
}
}  // namespace

// VectorFst<LatticeArc>::Write with default options inside an archive entry
// (kaldi-lattice.cc:388-394; vector-fst.h WriteFst): header, then per state the final weight
// (two floats), int64 arc count, arcs {ilabel, olabel, graph, acoustic, nextstate}
void SerializeLattice(const LatticeOut &lat, std::string *b) {
  const int64_t ns = (int64_t)lat.arcs.size();
  int64_t na = 0;
  for (const auto &a : lat.arcs) na += (int64_t)a.size();
  b->reserve(b->size() + 96 + 16 * (size_t)ns + 20 * (size_t)na);
  PutFstHeader(b, "lattice4", ns ? 0 : -1, ns, na);
  for (int64_t s = 0; s < ns; s++) {
    const float fg = lat.final_graph[s], fa = std::isinf(fg) ? std::numeric_limits<float>::infinity() : lat.final_acoustic[s];
    Put(b, fg); Put(b, fa);
    Put<int64_t>(b, (int64_t)lat.arcs[s].size());
    if (!lat.arcs[s].empty()) b->append(reinterpret_cast<const char *>(lat.arcs[s].data()), 20 * lat.arcs[s].size());
  }
}
void SerializeLattice(const vblat::Lattice &lat, std::string *b) {
  const size_t na = lat.NumArcs();
  b->reserve(b->size() + 96 + 16 * lat.st.size() + 20 * na);
  PutFstHeader(b, "lattice4", lat.start, (int64_t)lat.st.size(), (int64_t)na);
  for (const vblat::State &s : lat.st) {
    Put(b, s.final.v1); Put(b, s.final.v2);
    Put<int64_t>(b, (int64_t)s.arcs.size());
    if (!s.arcs.empty()) b->append(reinterpret_cast<const char *>(s.arcs.data()), 20 * s.arcs.size());
  }
}
void SerializeLattice(const vblat::CompactLattice &clat, std::string *b) {
  // One allocation and plain stores: an entry of a rescored raw lattice is half a megabyte of 4-byte fields, and sixteen threads growing
  // strings field by field meet in the allocator (round 5: 1.0 -> 0.3 ms per entry)
  size_t na = 0, nstr = 0;
  for (const vblat::CState &s : clat.st) { na += s.arcs.size(); nstr += s.final.str.size(); for (const vblat::CArc &a : s.arcs) nstr += a.w.str.size(); }
  const size_t head0 = b->size();
  PutFstHeader(b, "compactlattice44", clat.start, (int64_t)clat.st.size(), (int64_t)na);
  const size_t body0 = b->size();
  // per state: final weight (2 floats + int32 length + string) + int64 arc count; per arc: 2 labels + weight (2 floats + length + string) + next state
  const size_t body = clat.st.size() * (12 + 8) + na * (8 + 12 + 4) + 4 * nstr;
  b->resize(body0 + body);
  char *p = &(*b)[body0];
  auto put = [&](const void *src, size_t n) { memcpy(p, src, n); p += n; };
  auto put_w = [&](const vblat::CWeight &w) {
    put(&w.w.v1, 4); put(&w.w.v2, 4);
    const int32_t len = (int32_t)w.str.size();
    put(&len, 4);
    if (len) put(w.str.data(), 4 * (size_t)len);
  };
  for (const vblat::CState &s : clat.st) {
    put_w(s.final);
    const int64_t n = (int64_t)s.arcs.size();
    put(&n, 8);
    for (const vblat::CArc &a : s.arcs) { put(&a.label, 4); put(&a.label, 4); put_w(a.w); put(&a.next, 4); }
  }
  if ((size_t)(p - &(*b)[0]) != body0 + body) Fail("internal: compact lattice entry size");
  (void)head0;
}

// Text form of a lattice entry ("ark,t:"), as the reference prints it (K/lat/kaldi-lattice.cc:62-90,388-415: a newline after the key,
// OpenFst's FstPrinter -- the start state first, then the others in order; per arc "state next ilabel [olabel] [weight]" with tabs, the
// weight left out when it is One; per final state (or state without arcs) "state [weight]" -- and an empty line at the end).  Weights:
// "graph,acoustic" resp. "graph,acoustic,tid_tid_..." with the stream's default float format (%g, 6 digits; K/fstext/lattice-weight.h).
namespace {
void PutTextFloat(std::string *t, float f) {
  if (f == std::numeric_limits<float>::infinity()) { *t += "Infinity"; return; }
  if (f == -std::numeric_limits<float>::infinity()) { *t += "-Infinity"; return; }
  if (f != f) { *t += "BadNumber"; return; }
  char b[48];
  snprintf(b, sizeof(b), "%g", (double)f);
  *t += b;
}
void PutTextWeight(std::string *t, const vblat::Weight &w) { PutTextFloat(t, w.v1); *t += ','; PutTextFloat(t, w.v2); }
void PutTextCWeight(std::string *t, const vblat::CWeight &w) {
  PutTextWeight(t, w.w);
  *t += ',';
  for (size_t i = 0; i < w.str.size(); i++) { if (i) *t += '_'; *t += std::to_string(w.str[i]); }
}
bool TxtIsOne(const vblat::Weight &w) { return w.v1 == 0.0f && w.v2 == 0.0f; }
bool TxtIsZero(const vblat::Weight &w) { return w.v1 == std::numeric_limits<float>::infinity() && w.v2 == std::numeric_limits<float>::infinity(); }
}  // namespace
std::string LatticeObjectToText(const std::string &object) {
  vblat::Lattice lat; vblat::CompactLattice clat; bool compact = false;
  ParseLatticeBytes(object.data(), object.size(), &lat, &clat, &compact);
  std::string t = "\n";
  const int32_t start = compact ? clat.start : lat.start;
  const int32_t ns = (int32_t)(compact ? clat.st.size() : lat.st.size());
  auto print_state = [&](int32_t s) {
    bool output = false;
    if (!compact) {
      const vblat::State &st = lat.st[s];
      for (const vblat::Arc &a : st.arcs) {
        t += std::to_string(s); t += '\t'; t += std::to_string(a.next); t += '\t'; t += std::to_string(a.ilabel); t += '\t'; t += std::to_string(a.olabel);
        if (!TxtIsOne(a.w)) { t += '\t'; PutTextWeight(&t, a.w); }
        t += '\n'; output = true;
      }
      if (!TxtIsZero(st.final) || !output) {
        t += std::to_string(s);
        if (!TxtIsOne(st.final)) { t += '\t'; PutTextWeight(&t, st.final); }
        t += '\n';
      }
    } else {
      const vblat::CState &st = clat.st[s];
      for (const vblat::CArc &a : st.arcs) {
        t += std::to_string(s); t += '\t'; t += std::to_string(a.next); t += '\t'; t += std::to_string(a.label);
        if (!(TxtIsOne(a.w.w) && a.w.str.empty())) { t += '\t'; PutTextCWeight(&t, a.w); }
        t += '\n'; output = true;
      }
      if (!TxtIsZero(st.final.w) || !output) {
        t += std::to_string(s);
        if (!(TxtIsOne(st.final.w) && st.final.str.empty())) { t += '\t'; PutTextCWeight(&t, st.final); }
        t += '\n';
      }
    }
  };
  if (start >= 0 && start < ns) {
    print_state(start);
    for (int32_t s = 0; s < ns; s++) if (s != start) print_state(s);
  }
  t += '\n';
  return t;
}

void LatticeWriter::WriteSerialized(const std::string &key, std::string &&object) {
  if (mem_) { mem_->emplace_back(key, std::move(object)); return; }
  WriteSerialized(key, static_cast<const std::string &>(object));
}
void LatticeWriter::WriteSerialized(const std::string &key, const std::string &object) {
  if (mem_) { mem_->emplace_back(key, object); return; }
  ark_ << key << ' ';
  if (spec_.text) { const std::string t = LatticeObjectToText(object); ark_.write(t.data(), (std::streamsize)t.size()); }
  else ark_.write(object.data(), (std::streamsize)object.size());
  if (ark_.fail()) Fail("write failure on " + spec_.ark_path);
}
void LatticeWriter::Write(const std::string &key, const LatticeOut &lat) {
  buf_.clear(); SerializeLattice(lat, &buf_); WriteSerialized(key, buf_);
}
void LatticeWriter::Write(const std::string &key, const vblat::Lattice &lat) {
  buf_.clear(); SerializeLattice(lat, &buf_); WriteSerialized(key, buf_);
}

void CompactLatticeWriter::Open(const std::string &wspecifier) {
  spec_ = ParseWspecifier(wspecifier);
  if (IsMemPath(spec_.ark_path)) {
    if (spec_.text || spec_.has_scp) Fail("an in-memory table is binary and has no script file: " + wspecifier);
    mem_ = std::make_shared<MemTable>();
    open_ = true;
    return;
  }
  ark_.open(spec_.ark_path, std::ios::binary);
  if (!ark_) Fail("cannot open " + spec_.ark_path + " for writing");
  open_ = true;
}
CompactLatticeWriter::~CompactLatticeWriter() { if (mem_) MemPublish(spec_.ark_path, std::move(mem_)); }
void CompactLatticeWriter::WriteSerialized(const std::string &key, std::string &&object) {
  if (mem_) { mem_->emplace_back(key, std::move(object)); return; }
  WriteSerialized(key, static_cast<const std::string &>(object));
}
void CompactLatticeWriter::WriteSerialized(const std::string &key, const std::string &object) {
  if (mem_) { mem_->emplace_back(key, object); return; }
  ark_ << key << ' ';
  if (spec_.text) { const std::string t = LatticeObjectToText(object); ark_.write(t.data(), (std::streamsize)t.size()); }
  else ark_.write(object.data(), (std::streamsize)object.size());
  if (ark_.fail()) Fail("write failure on " + spec_.ark_path);
}
void CompactLatticeWriter::Write(const std::string &key, const vblat::CompactLattice &clat) {
  buf_.clear(); SerializeLattice(clat, &buf_); WriteSerialized(key, buf_);
}

namespace {
struct StreamIn {            // an entry addressed through an .scp line
  std::istream &is;
  template <class T> T Get() { T v; is.read(reinterpret_cast<char *>(&v), sizeof(T)); if (is.fail()) Fail("lattice: truncated"); return v; }
  void Read(void *dst, size_t n) { is.read(reinterpret_cast<char *>(dst), (std::streamsize)n); if (is.fail()) Fail("lattice: truncated"); }
  int Peek() { return is.peek(); }
  size_t Left() const { return std::numeric_limits<size_t>::max(); }
};
struct MemIn {               // cursor over an archive mapped into memory
  const char *p, *end;
  template <class T> T Get() { T v; Read(&v, sizeof(T)); return v; }
  void Read(void *dst, size_t n) { if ((size_t)(end - p) < n) Fail("lattice: truncated"); memcpy(dst, p, n); p += n; }
  int Peek() { return p < end ? (unsigned char)*p : EOF; }
  size_t Left() const { return (size_t)(end - p); }
};
// one line of a text-mode entry (without its end-of-line); false at the end of the input
bool GetTextLine(StreamIn &in, std::string *line) { return (bool)std::getline(in.is, *line); }
bool GetTextLine(MemIn &in, std::string *line) {
  if (in.p >= in.end) return false;
  const char *e = static_cast<const char *>(memchr(in.p, '\n', (size_t)(in.end - in.p)));
  if (!e) { line->assign(in.p, in.end); in.p = in.end; return true; }
  line->assign(in.p, e);
  in.p = e + 1;
  return true;
}
bool TextFloat(const char **p, float *f) {
  const char *q = *p;
  if (!strncmp(q, "Infinity", 8)) { *f = std::numeric_limits<float>::infinity(); *p = q + 8; return true; }
  if (!strncmp(q, "-Infinity", 9)) { *f = -std::numeric_limits<float>::infinity(); *p = q + 9; return true; }
  char *e = nullptr;
  *f = strtof(q, &e);
  if (e == q) return false;
  *p = e;
  return true;
}
bool TextWeight(const std::string &s, vblat::Weight *w) {      // "graph,acoustic"
  const char *p = s.c_str();
  if (!TextFloat(&p, &w->v1) || *p != ',') return false;
  p++;
  return TextFloat(&p, &w->v2) && *p == '\0';
}
bool TextCWeight(const std::string &s, vblat::CWeight *w) {    // "graph,acoustic,tid_tid_..." (the string may be empty)
  const char *p = s.c_str();
  if (!TextFloat(&p, &w->w.v1) || *p != ',') return false;
  p++;
  if (!TextFloat(&p, &w->w.v2) || *p != ',') return false;
  p++;
  w->str.clear();
  while (*p) {
    char *e = nullptr;
    const long v = strtol(p, &e, 10);
    if (e == p || v <= 0) return false;
    w->str.push_back((int32_t)v);
    p = e;
    if (*p == '_') p++; else if (*p) return false;
  }
  return true;
}
bool TextInt(const std::string &s, int32_t *v) {
  char *e = nullptr;
  const long x = strtol(s.c_str(), &e, 10);
  if (s.empty() || *e != '\0' || x < 0 || x > 2147483647L) return false;
  *v = (int32_t)x;
  return true;
}
// K/lat/kaldi-lattice.cc:94-262 (LatticeReader::ReadText): the entry is read as a Lattice and as a CompactLattice at the same time until a
// line rules one of them out (4 columns can be either: an arc without weight of the one, an arc with weight of the other)
template <class In>
void ReadLatticeText(In &in, vblat::Lattice *lat, vblat::CompactLattice *clat, bool *is_compact) {
  bool lat_ok = true, clat_ok = true;
  lat->st.clear(); clat->st.clear(); lat->start = -1; clat->start = -1;
  std::string line;
  size_t nline = 0;
  auto grow = [&](int32_t s) {
    if (lat_ok && (size_t)s >= lat->st.size()) lat->st.resize((size_t)s + 1);
    if (clat_ok && (size_t)s >= clat->st.size()) clat->st.resize((size_t)s + 1);
  };
  while (GetTextLine(in, &line)) {
    std::vector<std::string> col;
    size_t i = 0;
    while (i < line.size()) {
      while (i < line.size() && strchr("\t \r\n", line[i])) i++;
      size_t j = i;
      while (j < line.size() && !strchr("\t \r\n", line[j])) j++;
      if (j > i) col.push_back(line.substr(i, j - i));
      i = j;
    }
    if (nline == 0 && col.empty() && line.empty()) { nline = 0; continue; }     // the newline that follows the key
    if (col.empty()) break;                                                       // the empty line that ends the entry
    nline++;
    int32_t s = 0, d = 0, il = 0, ol = 0;
    if (col.size() > 5 || !TextInt(col[0], &s)) Fail("lattice (text): bad line: " + line);
    grow(s);
    if (nline == 1) { lat->start = s; clat->start = s; }
    if (lat_ok) {
      vblat::Weight w = vblat::One();
      switch (col.size()) {
        case 1: lat->st[s].final = vblat::One(); break;
        case 2: if (TextWeight(col[1], &w)) lat->st[s].final = w; else lat_ok = false; break;
        case 4: case 5:
          if (TextInt(col[1], &d) && TextInt(col[2], &il) && TextInt(col[3], &ol) && (col.size() == 4 || (TextWeight(col[4], &w) &&
              !(w.v1 == std::numeric_limits<float>::infinity() && w.v2 == std::numeric_limits<float>::infinity())))) {
            grow(d);
            lat->st[s].arcs.push_back(vblat::Arc{il, ol, w, d});
          } else lat_ok = false;
          break;
        default: lat_ok = false;
      }
    }
    if (clat_ok) {
      vblat::CWeight w; w.w = vblat::One();
      switch (col.size()) {
        case 1: clat->st[s].final = w; break;
        case 2: if (TextCWeight(col[1], &w)) clat->st[s].final = w; else clat_ok = false; break;
        case 3: case 4:
          if (TextInt(col[1], &d) && TextInt(col[2], &il) && (col.size() == 3 || (TextCWeight(col[3], &w) &&
              !(w.w.v1 == std::numeric_limits<float>::infinity() && w.w.v2 == std::numeric_limits<float>::infinity())))) {
            if ((size_t)d >= clat->st.size()) clat->st.resize((size_t)d + 1);
            if (lat_ok && (size_t)d >= lat->st.size()) lat->st.resize((size_t)d + 1);
            clat->st[s].arcs.push_back(vblat::CArc{il, w, d});
          } else clat_ok = false;
          break;
        default: clat_ok = false;
      }
    }
    if (!lat_ok && !clat_ok) Fail("lattice (text): bad line: " + line);
  }
  // (a state that only appears as a destination was added with the default weight, Zero: not final)
  *is_compact = clat_ok && (!lat_ok || clat->NumArcs() > 0 || lat->NumArcs() == 0);
  if (*is_compact) { lat->st.clear(); lat->start = -1; } else { clat->st.clear(); clat->start = -1; }
}

template <class In> std::string GetFstString(In &in) {
  const int32_t n = in.template Get<int32_t>();
  if (n < 0 || n > 4096) Fail("fst: bad string length in header");
  std::string s((size_t)n, '\0');
  if (n) in.Read(&s[0], (size_t)n);
  return s;
}
template <class In>
void ReadLatticeBody(In &in, vblat::Lattice *lat, vblat::CompactLattice *clat, bool *is_compact) {
  const int c = in.Peek();
  if (c == EOF) Fail("lattice: unexpected end of file");
  if (isspace(c)) { ReadLatticeText(in, lat, clat, is_compact); return; }
  if (in.template Get<int32_t>() != kFstMagic) Fail("lattice: bad magic number");
  const std::string fsttype = GetFstString(in), arctype = GetFstString(in);
  in.template Get<int32_t>();                       // version
  const int32_t flags = in.template Get<int32_t>();
  in.template Get<uint64_t>();                      // properties
  const int64_t start = in.template Get<int64_t>(), ns = in.template Get<int64_t>();
  in.template Get<int64_t>();                       // arc count (not relied upon)
  if (fsttype != "vector") Fail("lattice: fst type " + fsttype + " not supported");
  if (flags & 3) Fail("lattice: embedded symbol tables are not supported");
  if (ns < 0 || ns > (1LL << 31) || (size_t)ns > in.Left() / 8) Fail("lattice: bad state count");
  if (arctype == "lattice4") {
    *is_compact = false;
    lat->st.assign((size_t)ns, vblat::State());
    lat->start = (int32_t)start;
    for (int64_t s = 0; s < ns; s++) {
      vblat::State &st = lat->st[s];
      st.final.v1 = in.template Get<float>(); st.final.v2 = in.template Get<float>();
      const int64_t na = in.template Get<int64_t>();
      if (na < 0 || (size_t)na > in.Left() / 20) Fail("lattice: truncated");
      st.arcs.resize((size_t)na);
      if (na) in.Read(st.arcs.data(), 20 * (size_t)na);
      for (const vblat::Arc &a : st.arcs)
        if (a.next < 0 || a.next >= ns || a.ilabel < 0 || a.olabel < 0) Fail("lattice: arc with a label or destination state out of range");
    }
  } else if (arctype == "compactlattice44") {
    *is_compact = true;
    clat->st.assign((size_t)ns, vblat::CState());
    clat->start = (int32_t)start;
    auto get_w = [&](vblat::CWeight *w) {
      w->w.v1 = in.template Get<float>(); w->w.v2 = in.template Get<float>();
      const int32_t n = in.template Get<int32_t>();
      if (n < 0 || (size_t)n > in.Left() / 4) Fail("lattice: bad string size");
      w->str.resize(n);
      if (n) in.Read(w->str.data(), 4 * (size_t)n);
      for (int32_t t : w->str) if (t <= 0) Fail("lattice: transition-id out of range");
    };
    for (int64_t s = 0; s < ns; s++) {
      vblat::CState &st = clat->st[s];
      get_w(&st.final);
      const int64_t na = in.template Get<int64_t>();
      if (na < 0 || (size_t)na > in.Left() / 20) Fail("lattice: truncated");
      st.arcs.resize((size_t)na);
      for (vblat::CArc &a : st.arcs) {
        a.label = in.template Get<int32_t>(); in.template Get<int32_t>(); get_w(&a.w); a.next = in.template Get<int32_t>();
        if (a.next < 0 || a.next >= ns || a.label < 0) Fail("lattice: arc with a label or destination state out of range");
      }
    }
  } else Fail("lattice: arc type " + arctype + " not supported");
  if (start < -1 || start >= ns) Fail("lattice: start state out of range");
}
}  // namespace

namespace {
// the end of the entry that starts at in.p, without building anything
void SkimLatticeBody(MemIn &in) {
  auto skip = [&](size_t n) { if (in.Left() < n) Fail("lattice: truncated"); in.p += n; };
  const int c = in.Peek();
  if (c == EOF) Fail("lattice: unexpected end of file");
  if (isspace(c)) {        // text entry: the newline after the key, lines, an empty line
    std::string line;
    bool first = true;
    while (GetTextLine(in, &line)) {
      const bool blank = line.find_first_not_of("\t \r") == std::string::npos;
      if (blank && !first) break;
      first = false;
    }
    return;
  }
  if (in.Get<int32_t>() != kFstMagic) Fail("lattice: bad magic number");
  const std::string fsttype = GetFstString(in), arctype = GetFstString(in);
  skip(4 + 4 + 8 + 8);
  const int64_t ns = in.Get<int64_t>();
  skip(8);
  if (ns < 0 || (size_t)ns > in.Left() / 8) Fail("lattice: bad state count");
  if (arctype == "lattice4") {
    for (int64_t s = 0; s < ns; s++) {
      skip(8);
      const int64_t na = in.Get<int64_t>();
      if (na < 0 || (size_t)na > in.Left() / 20) Fail("lattice: truncated");
      skip(20 * (size_t)na);
    }
  } else if (arctype == "compactlattice44") {
    auto skip_w = [&]() {
      skip(8);
      const int32_t n = in.Get<int32_t>();
      if (n < 0 || (size_t)n > in.Left() / 4) Fail("lattice: bad string size");
      skip(4 * (size_t)n);
    };
    for (int64_t s = 0; s < ns; s++) {
      skip_w();
      const int64_t na = in.Get<int64_t>();
      if (na < 0 || (size_t)na > in.Left() / 20) Fail("lattice: truncated");
      for (int64_t k = 0; k < na; k++) { skip(8); skip_w(); skip(4); }
    }
  } else Fail("lattice: arc type " + arctype + " not supported");
}
}  // namespace

void ParseLatticeBytes(const char *p, size_t n, vblat::Lattice *lat, vblat::CompactLattice *clat, bool *is_compact) {
  MemIn in{p, p + n};
  ReadLatticeBody(in, lat, clat, is_compact);
}
void ParseLatticeBytes(const char *p, size_t n, vblat::Lattice *lat) {
  vblat::CompactLattice clat; bool compact = false;
  ParseLatticeBytes(p, n, lat, &clat, &compact);
  if (compact) vblat::ConvertLattice(clat, lat, true);
}

LatticeReader::LatticeReader(const std::string &rspecifier) : spec_(ParseRspecifier(rspecifier)) {
  if (spec_.is_scp) { scp_ = ReadScp(spec_.path); return; }
  if (IsMemPath(spec_.path)) {
    mem_ = MemLookup(spec_.path);
    if (!mem_) Fail("no in-memory table " + spec_.path);
    return;
  }
  const int fd = ::open(spec_.path.c_str(), O_RDONLY);
  if (fd < 0) Fail("cannot open archive " + spec_.path);
  struct stat sb;
  if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) { ::close(fd); Fail("archive " + spec_.path + " is not a regular file"); }
  map_len_ = (size_t)sb.st_size;
  if (map_len_) {
    void *m = mmap(nullptr, map_len_, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) { ::close(fd); Fail("cannot map archive " + spec_.path); }
    map_ = static_cast<const char *>(m);
  }
  ::close(fd);
}
LatticeReader::~LatticeReader() { if (map_) munmap(const_cast<char *>(map_), map_len_); }

bool LatticeReader::Next(std::string *key, vblat::Lattice *lat, vblat::CompactLattice *clat, bool *is_compact) {
  if (spec_.is_scp) {
    if (pos_ >= scp_.size()) return false;
    const ScpEntry &e = scp_[pos_++];
    std::ifstream f(e.path, std::ios::binary);
    if (!f) Fail("cannot open " + e.path);
    if (e.offset >= 0) f.seekg(e.offset);
    StreamIn in{f};
    ReadLatticeBody(in, lat, clat, is_compact);
    *key = e.key;
    return true;
  }
  if (mem_) {
    if (pos_ >= mem_->size()) return false;
    const std::pair<std::string, std::string> &e = (*mem_)[pos_++];
    *key = e.first;
    MemIn in{e.second.data(), e.second.data() + e.second.size()};
    ReadLatticeBody(in, lat, clat, is_compact);
    return true;
  }
  const char *p = map_ + pos_, *end = map_ + map_len_;
  while (p < end && isspace((unsigned char)*p)) p++;
  if (p == end) { pos_ = map_len_; return false; }
  const char *k = p;
  while (p < end && !isspace((unsigned char)*p)) p++;
  key->assign(k, p);
  if (p == end) Fail("archive: expected space after key " + *key);
  p++;
  MemIn in{p, end};
  ReadLatticeBody(in, lat, clat, is_compact);
  pos_ = (size_t)(in.p - map_);
  return true;
}
bool LatticeReader::NextBytes(std::string *key, const char **bytes, size_t *n) {
  if (spec_.is_scp) {
    vblat::Lattice lat; vblat::CompactLattice clat; bool compact = false;
    if (!Next(key, &lat, &clat, &compact)) return false;
    held_.emplace_back();
    if (compact) SerializeLattice(clat, &held_.back()); else SerializeLattice(lat, &held_.back());
    *bytes = held_.back().data(); *n = held_.back().size();
    return true;
  }
  if (mem_) {
    if (pos_ >= mem_->size()) return false;
    const std::pair<std::string, std::string> &e = (*mem_)[pos_++];
    *key = e.first; *bytes = e.second.data(); *n = e.second.size();
    return true;
  }
  const char *p = map_ + pos_, *end = map_ + map_len_;
  while (p < end && isspace((unsigned char)*p)) p++;
  if (p == end) { pos_ = map_len_; return false; }
  const char *k = p;
  while (p < end && !isspace((unsigned char)*p)) p++;
  key->assign(k, p);
  if (p == end) Fail("archive: expected space after key " + *key);
  p++;
  MemIn in{p, end};
  SkimLatticeBody(in);
  *bytes = p; *n = (size_t)(in.p - p);
  pos_ = (size_t)(in.p - map_);
  return true;
}
bool LatticeReader::NextCompact(std::string *key, vblat::CompactLattice *clat) {
  vblat::Lattice lat; bool compact = false;
  if (!Next(key, &lat, clat, &compact)) return false;
  if (!compact) Fail("expected a CompactLattice for key " + *key + ", found a state-level Lattice (conversion is not part of this build)");
  return true;
}
bool LatticeReader::NextLattice(std::string *key, vblat::Lattice *lat) {
  vblat::CompactLattice clat; bool compact = false;
  if (!Next(key, lat, &clat, &compact)) return false;
  if (compact) vblat::ConvertLattice(clat, lat, true);     // LatticeHolder::Read does the same (kaldi-lattice.cc:460-480)
  return true;
}

void WriteFloat(std::ostream &os, bool binary, float v) {
  if (binary) { os.put(4); os.write(reinterpret_cast<const char *>(&v), 4); } else os << v << ' ';
}

std::ofstream OpenArkForWriting(const std::string &wspecifier, bool *text) {
  const Wspec spec = ParseWspecifier(wspecifier);
  if (text) *text = spec.text;
  if (!spec.has_ark || spec.has_scp || (spec.text && !text)) Fail("expected a binary ark: wspecifier, got " + wspecifier);
  std::ofstream f(spec.ark_path, std::ios::binary);
  if (!f) Fail("cannot open " + spec.ark_path + " for writing");
  return f;
}

// One buffer per entry, one stream call (a job's table has a few million 5-byte fields)
void SerializePosterior(const Posterior &post, std::string *b, bool text) {
  if (text) {       // posterior.cc:45-55: "[ tid weight tid weight ] [ ... ] \n", numbers as operator<< prints them at the precision 7 Kaldi gives its text streams (InitKaldiOutputStream)
    char num[48];
    for (const auto &fr : post) {
      b->append("[ ");
      for (const auto &pr : fr) { snprintf(num, sizeof(num), "%d %.7g ", pr.first, (double)pr.second); b->append(num); }
      b->append("] ");
    }
    b->push_back('\n');
    return;
  }
  size_t n = 0;
  for (const auto &fr : post) n += fr.size();
  b->reserve(b->size() + 7 + 5 * post.size() + 10 * n);
  auto put_i32 = [&](int32_t v) { b->push_back(4); b->append(reinterpret_cast<const char *>(&v), 4); };
  b->push_back('\0'); b->push_back('B');
  put_i32((int32_t)post.size());
  for (const auto &fr : post) {
    put_i32((int32_t)fr.size());
    for (const auto &pr : fr) { put_i32(pr.first); b->push_back(4); b->append(reinterpret_cast<const char *>(&pr.second), 4); }
  }
}
void WritePosteriorEntry(std::ostream &os, const std::string &key, const Posterior &post, bool text) {
  std::string b;
  SerializePosterior(post, &b, text);
  os << key << ' ';
  os.write(b.data(), (std::streamsize)b.size());
}

void WriteGaussPostEntry(std::ostream &os, const std::string &key, const GaussPost &gpost) {
  os << key << ' ';
  os.put('\0'); os.put('B');
  WriteInt32(os, true, (int32_t)gpost.size());
  for (const auto &fr : gpost) {
    WriteInt32(os, true, (int32_t)fr.size());
    for (const auto &pr : fr) {
      WriteInt32(os, true, pr.first);
      os << "FV ";
      WriteInt32(os, true, (int32_t)pr.second.size());
      if (!pr.second.empty()) os.write(reinterpret_cast<const char *>(pr.second.data()), 4 * pr.second.size());
    }
  }
}

namespace {
std::shared_ptr<const FlatPostTable> MemLookupPost(const std::string &path) {
  std::lock_guard<std::mutex> lk(g_mem_mu);
  auto it = g_mem_posts.find(path);
  return it == g_mem_posts.end() ? nullptr : it->second;
}
// one entry's bytes after "key " as SerializePosterior makes them, from the flat form
void SerializeFlatPost(const FlatPost &post, std::string *b, bool text) {
  const size_t T = post.frames();
  if (text) {
    char num[48];
    for (size_t t = 0; t < T; t++) {
      b->append("[ ");
      for (int32_t i = post.off[t]; i < post.off[t + 1]; i++) { snprintf(num, sizeof(num), "%d %.7g ", post.e[(size_t)i].first, (double)post.e[(size_t)i].second); b->append(num); }
      b->append("] ");
    }
    b->push_back('\n');
    return;
  }
  b->reserve(b->size() + 7 + 5 * T + 10 * post.e.size());
  auto put_i32 = [&](int32_t v) { b->push_back(4); b->append(reinterpret_cast<const char *>(&v), 4); };
  b->push_back('\0'); b->push_back('B');
  put_i32((int32_t)T);
  for (size_t t = 0; t < T; t++) {
    put_i32(post.off[t + 1] - post.off[t]);
    for (int32_t i = post.off[t]; i < post.off[t + 1]; i++) { put_i32(post.e[(size_t)i].first); b->push_back(4); b->append(reinterpret_cast<const char *>(&post.e[(size_t)i].second), 4); }
  }
}
}  // namespace

void FlattenPosterior(const Posterior &post, FlatPost *flat) {
  size_t n = 0;
  for (const auto &fr : post) n += fr.size();
  flat->off.clear(); flat->off.reserve(post.size() + 1); flat->off.push_back(0);
  flat->e.clear(); flat->e.reserve(n);
  for (const auto &fr : post) { flat->e.insert(flat->e.end(), fr.begin(), fr.end()); flat->EndFrame(); }
}

void WritePosteriorTableFlat(const std::string &wspecifier, std::shared_ptr<FlatPostTable> table) {
  const Wspec spec = ParseWspecifier(wspecifier);
  if (!spec.has_ark || spec.has_scp) Fail("expected an ark: wspecifier, got " + wspecifier);
  if (IsMemPath(spec.ark_path)) {
    if (spec.text) Fail("an in-memory table is binary: " + wspecifier);
    std::lock_guard<std::mutex> lk(g_mem_mu);
    g_mem_posts[spec.ark_path] = std::move(table);
    return;
  }
  std::ofstream f(spec.ark_path, std::ios::binary);
  if (!f) Fail("cannot open " + spec.ark_path + " for writing");
  std::string b;
  for (const auto &kv : *table) {
    b.clear();
    b.append(kv.first); b.push_back(' ');
    SerializeFlatPost(kv.second, &b, spec.text);
    f.write(b.data(), (std::streamsize)b.size());
  }
  if (f.fail()) Fail("write failure on " + spec.ark_path);
}

std::shared_ptr<const FlatPostTable> ReadPosteriorTableFlat(const std::string &rspecifier) {
  const Rspec spec = ParseRspecifier(rspecifier);
  if (spec.is_scp) Fail("scp: is not supported for posterior tables");
  if (IsMemPath(spec.path)) {
    std::shared_ptr<const FlatPostTable> t = MemLookupPost(spec.path);
    if (!t) Fail("cannot open " + spec.path + " (no such in-memory posterior table)");
    return t;
  }
  std::string buf;
  {
    std::ifstream is(spec.path, std::ios::binary | std::ios::ate);
    if (!is) Fail("cannot open " + spec.path);
    const std::streamoff n = is.tellg();
    is.seekg(0);
    buf.resize((size_t)std::max<std::streamoff>(n, 0));
    if (n > 0) is.read(&buf[0], n);
    if (is.fail()) Fail("read failure on " + spec.path);
  }
  const char *p = buf.data(), *end = p + buf.size();
  auto need = [&](size_t n) { if ((size_t)(end - p) < n) Fail("posterior: truncated table " + spec.path); };
  auto get_i32 = [&]() { need(5); if (*p != 4) Fail("posterior: expected a 4-byte integer in " + spec.path); int32_t v; memcpy(&v, p + 1, 4); p += 5; return v; };
  auto out = std::make_shared<FlatPostTable>();
  for (;;) {
    while (p < end && isspace((unsigned char)*p)) p++;
    if (p == end) break;
    const char *k = p;
    while (p < end && !isspace((unsigned char)*p)) p++;
    const std::string key(k, p);
    need(3);
    p++;
    FlatPost post;
    if (p[0] != '\0' || p[1] != 'B') {       // text: the rest of the line, "[ tid weight ... ] [ ... ]" (posterior.cc:80-125)
      const char *eol = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));
      if (!eol) eol = end;
      std::istringstream ls(std::string(p, eol));
      std::string t;
      while (ls >> t) {
        if (t != "[") Fail("posterior: expected [ in the text entry of " + key + " in " + spec.path);
        for (;;) {
          if (!(ls >> t)) Fail("posterior: unterminated frame in the text entry of " + key);
          if (t == "]") break;
          char *e1 = nullptr;
          const long tid = strtol(t.c_str(), &e1, 10);
          float w;
          if (e1 == t.c_str() || *e1 || !(ls >> w)) Fail("posterior: bad pair in the text entry of " + key);
          post.e.emplace_back((int32_t)tid, w);
        }
        post.EndFrame();
      }
      out->emplace_back(key, std::move(post));
      p = eol;
      continue;
    }
    p += 2;
    const int32_t sz = get_i32();
    if (sz < 0 || (size_t)sz > (size_t)(end - p) / 5) Fail("posterior: bad size");
    post.off.reserve((size_t)sz + 1);
    post.e.reserve((size_t)sz + (size_t)sz / 2);
    for (int32_t t = 0; t < sz; t++) {
      const int32_t n = get_i32();
      if (n < 0 || (size_t)n > (size_t)(end - p) / 10) Fail("posterior: bad size");
      for (int32_t i = 0; i < n; i++) {
        if (p[0] != 4 || p[5] != 4) Fail("posterior: bad field size in " + spec.path);
        std::pair<int32_t, float> pr;
        memcpy(&pr.first, p + 1, 4); memcpy(&pr.second, p + 6, 4);
        post.e.push_back(pr);
        p += 10;
      }
      post.EndFrame();
    }
    out->emplace_back(key, std::move(post));
  }
  return out;
}

std::vector<std::pair<std::string, Posterior>> ReadPosteriorTable(const std::string &rspecifier) {
  const Rspec spec = ParseRspecifier(rspecifier);
  if (spec.is_scp) Fail("scp: is not supported for posterior tables");
  if (IsMemPath(spec.path)) {       // a table published as an object: frame vectors made from the flat form
    std::shared_ptr<const FlatPostTable> t = MemLookupPost(spec.path);
    if (!t) Fail("cannot open " + spec.path + " (no such in-memory posterior table)");
    std::vector<std::pair<std::string, Posterior>> out;
    for (const auto &kv : *t) {
      Posterior post(kv.second.frames());
      for (size_t f = 0; f < post.size(); f++) post[f].assign(kv.second.e.begin() + kv.second.off[f], kv.second.e.begin() + kv.second.off[f + 1]);
      out.emplace_back(kv.first, std::move(post));
    }
    return out;
  }
  std::string buf;
  {
    std::ifstream is(spec.path, std::ios::binary | std::ios::ate);
    if (!is) Fail("cannot open " + spec.path);
    const std::streamoff n = is.tellg();
    is.seekg(0);
    buf.resize((size_t)std::max<std::streamoff>(n, 0));
    if (n > 0) is.read(&buf[0], n);
    if (is.fail()) Fail("read failure on " + spec.path);
  }
  const char *p = buf.data(), *end = p + buf.size();
  auto need = [&](size_t n) { if ((size_t)(end - p) < n) Fail("posterior: truncated table " + spec.path); };
  auto get_i32 = [&]() { need(5); if (*p != 4) Fail("posterior: expected a 4-byte integer in " + spec.path); int32_t v; memcpy(&v, p + 1, 4); p += 5; return v; };
  std::vector<std::pair<std::string, Posterior>> out;
  for (;;) {
    while (p < end && isspace((unsigned char)*p)) p++;
    if (p == end) break;
    const char *k = p;
    while (p < end && !isspace((unsigned char)*p)) p++;
    const std::string key(k, p);
    need(3);
    p++;
    if (p[0] != '\0' || p[1] != 'B') {       // text: the rest of the line, "[ tid weight ... ] [ ... ]" (posterior.cc:80-125)
      const char *eol = static_cast<const char *>(memchr(p, '\n', (size_t)(end - p)));
      if (!eol) eol = end;
      std::istringstream ls(std::string(p, eol));
      Posterior post;
      std::string t;
      while (ls >> t) {
        if (t != "[") Fail("posterior: expected [ in the text entry of " + key + " in " + spec.path);
        post.emplace_back();
        for (;;) {
          if (!(ls >> t)) Fail("posterior: unterminated frame in the text entry of " + key);
          if (t == "]") break;
          char *e1 = nullptr;
          const long tid = strtol(t.c_str(), &e1, 10);
          float w;
          if (e1 == t.c_str() || *e1 || !(ls >> w)) Fail("posterior: bad pair in the text entry of " + key);
          post.back().emplace_back((int32_t)tid, w);
        }
      }
      out.emplace_back(key, std::move(post));
      p = eol;
      continue;
    }
    p += 2;
    const int32_t sz = get_i32();
    if (sz < 0 || (size_t)sz > (size_t)(end - p) / 5) Fail("posterior: bad size");
    Posterior post(sz);
    for (auto &fr : post) {
      const int32_t n = get_i32();
      if (n < 0 || (size_t)n > (size_t)(end - p) / 10) Fail("posterior: bad size");
      fr.resize(n);
      for (auto &pr : fr) {
        need(10);
        if (p[0] != 4 || p[5] != 4) Fail("posterior: bad field size in " + spec.path);
        memcpy(&pr.first, p + 1, 4); memcpy(&pr.second, p + 6, 4);
        p += 10;
      }
    }
    out.emplace_back(key, std::move(post));
  }
  return out;
}

std::vector<std::pair<std::string, GaussPost>> ReadGaussPostTable(const std::string &rspecifier) {
  const Rspec spec = ParseRspecifier(rspecifier);
  if (spec.is_scp) Fail("scp: is not supported for Gaussian-posterior tables");
  std::ifstream is(spec.path, std::ios::binary);
  if (!is) Fail("cannot open " + spec.path);
  std::vector<std::pair<std::string, GaussPost>> out;
  for (;;) {
    const std::string key = ReadArkKey(is);
    if (key.empty()) break;
    if (!ReadBinaryHeader(is)) Fail("text-mode Gaussian posteriors are not supported: " + spec.path);
    const int32_t sz = ReadInt32(is, true);
    if (sz < 0 || sz > 10000000) Fail("gpost: bad size");
    GaussPost gp(sz);
    for (auto &fr : gp) {
      const int32_t n = ReadInt32(is, true);
      if (n < 0 || n > 1000000) Fail("gpost: bad size");
      fr.resize(n);
      for (auto &pr : fr) { pr.first = ReadInt32(is, true); ReadVector(is, true, &pr.second); }
    }
    out.emplace_back(key, std::move(gp));
  }
  return out;
}

std::vector<std::pair<std::string, std::vector<std::string>>> ReadSpk2Utt(const std::string &rspecifier) { return ReadTokenTable(rspecifier); }

std::vector<std::pair<std::string, std::vector<int32_t>>> ReadIntVectorTable(const std::string &rspecifier) {
  const Rspec spec = ParseRspecifier(rspecifier);
  if (spec.is_scp) Fail("scp: is not supported for integer-vector tables");
  std::ifstream is(spec.path, std::ios::binary);
  if (!is) Fail("cannot open " + spec.path);
  std::vector<std::pair<std::string, std::vector<int32_t>>> out;
  for (;;) {
    const std::string key = ReadArkKey(is);
    if (key.empty()) break;
    std::vector<int32_t> v;
    if (is.peek() == '\0') {
      const bool binary = ReadBinaryHeader(is);
      ReadIntVector(is, binary, &v);
    } else {
      std::string line;
      std::getline(is, line);
      std::istringstream ls(line);
      long long x;
      while (ls >> x) v.push_back((int32_t)x);
      if (!ls.eof()) Fail("bad integer in table " + spec.path + " at key " + key);
    }
    out.emplace_back(key, std::move(v));
  }
  return out;
}

std::map<std::string, float> ReadFloatTable(const std::string &rspecifier) {
  const Rspec spec = ParseRspecifier(rspecifier);
  if (spec.is_scp) Fail("scp: is not supported for float tables");
  std::ifstream is(spec.path, std::ios::binary);
  if (!is) Fail("cannot open " + spec.path);
  std::map<std::string, float> out;
  for (;;) {
    const std::string key = ReadArkKey(is);
    if (key.empty()) break;
    float v = 0.0f;
    if (is.peek() == '\0') {                 // "\0B", then the size byte and the value (WriteBasicType)
      ReadBinaryHeader(is);
      const int sz = is.get();
      if (sz != 4) Fail("float table " + spec.path + ": expected a 4-byte float at key " + key);
      is.read(reinterpret_cast<char *>(&v), 4);
      if (!is) Fail("float table " + spec.path + " is truncated at key " + key);
    } else {
      std::string line;
      std::getline(is, line);
      std::istringstream ls(line);
      if (!(ls >> v)) Fail("bad value in float table " + spec.path + " at key " + key);
    }
    out[key] = v;
  }
  return out;
}

std::vector<std::pair<std::string, std::vector<std::string>>> ReadTokenTable(const std::string &rspecifier) {
  const Rspec spec = ParseRspecifier(rspecifier);
  if (spec.is_scp) Fail("scp: is not supported for token tables");
  std::ifstream is(spec.path);
  if (!is) Fail("cannot open " + spec.path);
  std::vector<std::pair<std::string, std::vector<std::string>>> out;
  std::string line;
  while (std::getline(is, line)) {
    std::istringstream ls(line);
    std::string key, tok;
    if (!(ls >> key)) continue;
    std::vector<std::string> v;
    while (ls >> tok) v.push_back(tok);
    out.emplace_back(key, std::move(v));
  }
  return out;
}

}  // namespace vbio
