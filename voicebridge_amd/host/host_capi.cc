// host_capi.cc -- plain-C access to the file-format layer (kaldi_io) for bindings and tests.
#include <cstring>
#include <string>

#include "fmllr.h"
#include "kaldi_io.h"

namespace {
thread_local std::string g_err;
thread_local vbio::Model g_model;
thread_local vbio::Graph g_graph;
template <class F> int Try(F f) {
  try { f(); return 0; } catch (const std::exception &e) { g_err = e.what(); return -1; }
}
}  // namespace

extern "C" {

const char *vbio_last_error() { return g_err.c_str(); }

// copy-feats (VB/src/featbin/copy-feats.cpp): table of matrices -> table, optionally compressed
int vbio_copy_feats(const char *rspec, const char *wspec, int compress) {
  return Try([&] {
    vbio::MatrixReader r(rspec);
    vbio::MatrixWriter w(wspec, compress != 0);
    std::string k; vbio::Matrix m;
    while (r.Next(&k, &m)) w.Write(k, m);
    w.Close();
  });
}

int vbio_read_wave(const char *path, float *out, long long cap, long long *n, float *samp_freq, int *channels) {
  return Try([&] {
    vbio::Wave w;
    vbio::ReadWaveFile(path, -1, &w);
    *n = w.data.empty() ? 0 : (long long)w.data[0].size();
    *samp_freq = w.samp_freq; *channels = w.channels;
    if (out) for (long long i = 0; i < *n && i < cap; i++) out[i] = w.data[0][i];
  });
}

int vbio_model_load(const char *path, int *num_pdfs, int *dim, int *num_tids, int *num_gauss) {
  return Try([&] {
    vbio::ReadModel(path, &g_model);
    *num_pdfs = g_model.num_pdfs; *dim = g_model.dim; *num_tids = (int)g_model.tid2pdf.size() - 1;
    *num_gauss = g_model.pdf_offsets.back();
  });
}
int vbio_model_get(int *tid2pdf, int *pdf_offsets, float *weights, float *miv, float *iv) {
  memcpy(tid2pdf, g_model.tid2pdf.data(), sizeof(int) * g_model.tid2pdf.size());
  memcpy(pdf_offsets, g_model.pdf_offsets.data(), sizeof(int) * g_model.pdf_offsets.size());
  memcpy(weights, g_model.weights.data(), sizeof(float) * g_model.weights.size());
  memcpy(miv, g_model.means_invvars.data(), sizeof(float) * g_model.means_invvars.size());
  memcpy(iv, g_model.inv_vars.data(), sizeof(float) * g_model.inv_vars.size());
  return 0;
}

int vbio_fst_load(const char *path, int *num_states, long long *num_arcs, int *start) {
  return Try([&] {
    vbio::ReadFst(path, &g_graph);
    *num_states = (int)g_graph.final_cost.size(); *num_arcs = (long long)g_graph.arcs.size(); *start = g_graph.start;
  });
}
int vbio_fst_get(float *final_cost, long long *arc_offset, void *arcs16) {
  memcpy(final_cost, g_graph.final_cost.data(), sizeof(float) * g_graph.final_cost.size());
  for (size_t i = 0; i < g_graph.arc_offset.size(); i++) arc_offset[i] = g_graph.arc_offset[i];
  memcpy(arcs16, g_graph.arcs.data(), 16 * g_graph.arcs.size());
  return 0;
}


// FmllrDiagGmmAccs::Update from given statistics (K [dim][dim+1], G [dim][dim+1][dim+1] full symmetric): the estimate without the
// accumulation, for tests that bring their own statistics.  xform holds the starting value on entry.
int vb_fmllr_update(int dim, double beta, const double *K, const double *G, const char *update_type, float min_count, int num_iters,
                    float *xform, float *impr) {
  return Try([&] {
    vbfmllr::FmllrAccs accs(dim);
    accs.AddStats(beta, K, G);
    vbfmllr::FmllrOptions o;
    o.update_type = update_type; o.min_count = min_count; o.num_iters = num_iters;
    float count = 0.0f;
    *impr = accs.Update(o, xform, &count);
  });
}

// drops an in-memory table ("mem:NAME", kaldi_io.h) as unlink() drops a file
void vbio_mem_unlink(const char *path) { vbio::MemUnlink(path); }
}  // extern "C"
