"""Host-side Python mirror of the reference's operator interface for the hot path, on top of
the C ABI (include/voicebridge_hip.h).  torch is used only to own device memory; every
compute call goes through libvoicebridge_hip.so and fails loudly when the library or the GPU
is missing.

Reference objects mirrored (paths relative to /root/reference/kaldi-master/src/):
  Mfcc             feat/feature-mfcc.h:143      ComputeFeatures(wave, ...)
  ComputeDeltas / ApplyCmvn / SpliceFrames / transform-feats -> FeaturePipeline
  AmDiagGmm + DecodableAmDiagGmmScaled          gmm/am-diag-gmm.h, gmm/decodable-am-diag-gmm.h
  LatticeFasterDecoder                          decoder/lattice-faster-decoder.h:130-155
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check

WIN = {"hanning": 0, "hamming": 1, "povey": 2, "rectangular": 3, "blackman": 4}


def _i64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def _i32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32)) if a is not None else None


def _f32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


class Context:
    """One per GPU (vbh_ctx)."""

    def __init__(self, device=0, priority=0, cus=0, first_cu=0):
        self.lib = _lib.load()
        if self.lib.vbh_device_count() <= 0:
            raise _lib.VbhError("no HIP device visible: voicebridge_amd has no CPU path")
        h = C.c_void_p()
        if cus > 0:       # confined to `cus` compute units (vbh_ctx_create_on_cus)
            check(self.lib.vbh_ctx_create_on_cus(int(device), int(first_cu), int(cus), C.byref(h)), "vbh_ctx_create_on_cus")
        else:
            check(self.lib.vbh_ctx_create_with_priority(int(device), int(priority), C.byref(h)), "vbh_ctx_create")
        self.h = h
        self.device = torch.device("cuda", int(device))

    def sync(self):
        check(self.lib.vbh_ctx_sync(self.h), "vbh_ctx_sync")

    def wait_for(self, other):
        """work enqueued here from now on starts after everything enqueued on `other` so far"""
        check(self.lib.vbh_ctx_wait_for(self.h, other.h), "vbh_ctx_wait_for")

    def mark(self, slot):
        self.lib.vbh_ctx_mark.argtypes = [C.c_void_p, C.c_int]
        check(self.lib.vbh_ctx_mark(self.h, int(slot)), "vbh_ctx_mark")

    def wait_mark(self, other, slot):
        """work enqueued here from now on starts after `other` has reached its mark `slot`"""
        self.lib.vbh_ctx_wait_mark.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        check(self.lib.vbh_ctx_wait_mark(self.h, other.h, int(slot)), "vbh_ctx_wait_mark")

    def torch_sync(self):
        """make work enqueued by torch (allocations, copies) visible to the ctx stream"""
        torch.cuda.synchronize(self.device)

    def timer_start(self, slot=0):
        check(self.lib.vbh_timer_start(self.h, slot), "vbh_timer_start")

    def timer_stop(self, slot=0):
        check(self.lib.vbh_timer_stop(self.h, slot), "vbh_timer_stop")

    def timer_ms(self, slot=0):
        ms = C.c_float()
        check(self.lib.vbh_timer_elapsed_ms(self.h, slot, C.byref(ms)), "vbh_timer_elapsed_ms")
        return ms.value

    def close(self):
        if self.h:
            self.lib.vbh_ctx_destroy(self.h)
            self.h = None


def mfcc_opts(**kw):
    o = _lib.MfccOpts()
    _lib.load().vbh_mfcc_opts_default(C.byref(o))
    for k, v in kw.items():
        if k == "window_type" and isinstance(v, str):
            v = WIN[v]
        if not hasattr(o, k):
            raise KeyError(k)
        setattr(o, k, v)
    return o


class Mfcc:
    """compute-mfcc-feats for a batch of utterances (vbh_mfcc_*)."""

    def __init__(self, ctx, vtln_warp=1.0, **opts):
        self.ctx, self.lib = ctx, ctx.lib
        self.opts = mfcc_opts(**opts)
        h = C.c_void_p()
        check(self.lib.vbh_mfcc_create(ctx.h, C.byref(self.opts), C.c_float(vtln_warp), C.byref(h)), "vbh_mfcc_create")
        self.h = h

    def num_frames(self, num_samples):
        return self.lib.vbh_mfcc_num_frames(C.byref(self.opts), int(num_samples))

    def set_dither_stream(self, seed=None, first_utt_index=0):
        """noise stream of the dither (vbh_mfcc_set_dither_stream): a seed and the job-wide index of the next call's first utterance"""
        if seed is not None:
            self._seed = int(seed)
        self.lib.vbh_mfcc_set_dither_stream.argtypes = [C.c_void_p, C.c_uint64, C.c_int64]
        check(self.lib.vbh_mfcc_set_dither_stream(self.h, getattr(self, "_seed", 0x4F6CDD1D2545F491), int(first_utt_index)), "vbh_mfcc_set_dither_stream")

    def pack(self, waves):
        """list of 1-D int16/float32 arrays -> (device tensor, int64 offsets[n+1], is_int16)"""
        is_i16 = all(w.dtype == np.int16 for w in waves)
        dt = np.int16 if is_i16 else np.float32
        offs = np.zeros(len(waves) + 1, np.int64)
        offs[1:] = np.cumsum([len(w) for w in waves])
        flat = np.concatenate([np.asarray(w, dt) for w in waves]) if waves else np.zeros(0, dt)
        return torch.from_numpy(flat).to(self.ctx.device), offs, is_i16

    def compute_packed(self, d_wave, wave_offsets, is_int16, out=None):
        n = len(wave_offsets) - 1
        total = sum(self.num_frames(wave_offsets[u + 1] - wave_offsets[u]) for u in range(n))
        fresh = out is None
        if fresh:
            out = torch.empty((total, self.opts.num_ceps), dtype=torch.float32, device=self.ctx.device)
            self.ctx.torch_sync()
        foff = np.zeros(n + 1, np.int64)
        wo = np.ascontiguousarray(wave_offsets, np.int64)
        check(self.lib.vbh_mfcc_compute(self.h, d_wave.data_ptr(), int(is_int16), _i64p(wo), n, out.data_ptr(), _i64p(foff)),
              "vbh_mfcc_compute")
        if fresh:
            self.ctx.sync()
        return out, foff

    def compute(self, waves):
        d_wave, offs, is_i16 = self.pack(waves)
        self.ctx.torch_sync()
        out, foff = self.compute_packed(d_wave, offs, is_i16)
        self.ctx.sync()
        return out, foff

    def close(self):
        if self.h:
            self.lib.vbh_mfcc_destroy(self.h)
            self.h = None


class FeaturePipeline:
    """apply-cmvn | add-deltas (or splice-feats | transform-feats final.mat) | transform-feats trans
    (VB/scr/steps/decode_gmm.cpp:201-293) as one fused device pass (vbh_featpost)."""

    def __init__(self, ctx, in_dim=13, cmvn=True, norm_vars=False, mode="delta", delta_order=2, delta_window=2,
                 splice_left=4, splice_right=4, lda=None, spk_xform_shape=None):
        self.ctx, self.lib = ctx, ctx.lib
        c = _lib.FeatPostCfg()
        c.in_dim, c.apply_cmvn, c.norm_vars = in_dim, int(cmvn), int(norm_vars)
        c.mode = {"none": 0, "delta": 1, "splice": 2}[mode]
        c.delta_order, c.delta_window, c.splice_left, c.splice_right = delta_order, delta_window, splice_left, splice_right
        self.lda = None
        if lda is not None:
            self.lda = np.ascontiguousarray(lda, np.float32)
            c.lda_rows, c.lda_cols = self.lda.shape
        if spk_xform_shape is not None:
            c.spk_rows, c.spk_cols = spk_xform_shape
        self.cfg = c
        self.out_dim = self.lib.vbh_featpost_out_dim(C.byref(c))

    def apply(self, d_in, frame_offsets, utt2spk=None, cmvn_stats=None, spk_xforms=None, out=None):
        n = len(frame_offsets) - 1
        fo = np.ascontiguousarray(frame_offsets, np.int64)
        u2s = np.ascontiguousarray(utt2spk, np.int32) if utt2spk is not None else np.zeros(n, np.int32)
        n_spk = int(u2s.max()) + 1 if n else 1
        st = np.ascontiguousarray(cmvn_stats, np.float64) if cmvn_stats is not None else None
        if st is not None:
            n_spk = st.shape[0]
        sx = np.ascontiguousarray(spk_xforms, np.float32) if spk_xforms is not None else None
        if sx is not None:
            n_spk = sx.shape[0]
        fresh = out is None
        if fresh:
            out = torch.empty((int(fo[-1]), self.out_dim), dtype=torch.float32, device=self.ctx.device)
            self.ctx.torch_sync()
        check(self.lib.vbh_featpost(self.ctx.h, C.byref(self.cfg), d_in.data_ptr(), _i64p(fo), n, _i32p(u2s), n_spk,
                                    st.ctypes.data_as(C.POINTER(C.c_double)) if st is not None else None,
                                    _f32p(self.lda), _f32p(sx), out.data_ptr()), "vbh_featpost")
        if fresh:
            self.ctx.sync()     # a tensor handed out is complete; with out= the call only enqueues (ctx.sync() when needed)
        return out

    def cmvn_stats(self, d_feats, frame_offsets, utt2spk, n_spk):
        fo = np.ascontiguousarray(frame_offsets, np.int64)
        u2s = np.ascontiguousarray(utt2spk, np.int32)
        dim = d_feats.shape[1]
        out = np.zeros((n_spk, 2, dim + 1), np.float64)
        check(self.lib.vbh_cmvn_stats(self.ctx.h, d_feats.data_ptr(), dim, _i64p(fo), len(fo) - 1, _i32p(u2s), n_spk,
                                      out.ctypes.data_as(C.POINTER(C.c_double))), "vbh_cmvn_stats")
        return out


class DiagGmm:
    """AmDiagGmm scored for every pdf of every frame (vbh_gmm_*)."""

    LSE_MODES = {"fast": 0, "strict": 1, "twopass": 2}      # VBH_LSE_* of include/voicebridge_hip.h

    def __init__(self, ctx, pdf_offsets, weights, means_invvars, inv_vars, lse=None):
        self.ctx, self.lib = ctx, ctx.lib
        offs = np.ascontiguousarray(pdf_offsets, np.int32)
        w = np.ascontiguousarray(weights, np.float32)
        miv = np.ascontiguousarray(means_invvars, np.float32)
        iv = np.ascontiguousarray(inv_vars, np.float32)
        self.num_pdfs, self.dim, self.num_gauss = len(offs) - 1, miv.shape[1], miv.shape[0]
        h = C.c_void_p()
        check(self.lib.vbh_gmm_create(ctx.h, self.dim, self.num_pdfs, _i32p(offs), _f32p(w), _f32p(miv), _f32p(iv), C.byref(h)),
              "vbh_gmm_create")
        self.h = h
        if lse is not None:
            self.set_lse_mode(lse)

    def set_lse_mode(self, mode):
        """'fast' (default), 'strict' (the reference's LogSumExp arithmetic: f64 sum, expf, log) or 'twopass' (its passes in f32)"""
        check(self.lib.vbh_gmm_set_lse_mode(self.h, self.LSE_MODES[mode]), "vbh_gmm_set_lse_mode")

    def gconsts(self):
        out = np.zeros(self.num_gauss, np.float32)
        check(self.lib.vbh_gmm_get_gconsts(self.h, _f32p(out)), "vbh_gmm_get_gconsts")
        return out

    @property
    def padded_pdfs(self):
        """num_pdfs rounded up to a multiple of 32: the row length at which the kernel's stores cover whole sectors (vbh_gmm_padded_pdfs)"""
        return int(self.lib.vbh_gmm_padded_pdfs(self.h))

    def loglikes(self, d_feats, out=None, padded=False):
        """[T x num_pdfs] scores; `out` may have longer rows (out.stride(0) >= num_pdfs, e.g. [T x padded_pdfs]: the decoders take
        such a matrix as it is, its extra columns are never read); padded=True allocates that form"""
        T = d_feats.shape[0]
        fresh = out is None
        if fresh:
            out = torch.empty((T, self.padded_pdfs if padded else self.num_pdfs), dtype=torch.float32, device=self.ctx.device)
            self.ctx.torch_sync()
        if T > 0:
            assert out.stride(1) == 1 and out.shape[0] >= T
            check(self.lib.vbh_gmm_loglikes_ld(self.h, d_feats.data_ptr(), T, out.data_ptr(), int(out.stride(0)) if T > 1 or out.shape[0] > 1 else int(out.shape[1])),
                  "vbh_gmm_loglikes_ld")
        if fresh:
            self.ctx.sync()
        return out

    def work(self, T):
        b, f = C.c_double(), C.c_double()
        self.lib.vbh_gmm_work(self.h, int(T), C.byref(b), C.byref(f))
        return b.value, f.value

    def fmllr_stats(self, d_feats, frame_offsets, utt2spk, n_spk, pair_offsets, pair_pdf, pair_weight, ali_gmm=None):
        """fMLLR sufficient statistics of a batch (vbh_fmllr_accumulate): returns (beta [S], K [S, D, D+1],
        G [S, D, D+1, D+1]) as float64 arrays and the accumulation kernel's HIP-event time in ms."""
        foff = np.ascontiguousarray(frame_offsets, np.int64)
        u2s = np.ascontiguousarray(utt2spk, np.int32)
        poff = np.ascontiguousarray(pair_offsets, np.int64)
        ppdf = np.ascontiguousarray(pair_pdf, np.int32)
        pw = np.ascontiguousarray(pair_weight, np.float32)
        D = self.dim
        beta = np.zeros(n_spk, np.float64); K = np.zeros((n_spk, D, D + 1), np.float64); G = np.zeros((n_spk, D, D + 1, D + 1), np.float64)
        ms = C.c_float()
        fn = self.lib.vbh_fmllr_accumulate
        fn.argtypes = [C.c_void_p] * 4 + [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32] + [C.c_void_p] * 6 + [C.POINTER(C.c_float)]
        check(fn(self.ctx.h, (ali_gmm or self).h, self.h, d_feats.data_ptr(), len(foff) - 1, foff.ctypes.data, u2s.ctypes.data, int(n_spk),
                 poff.ctypes.data, ppdf.ctypes.data, pw.ctypes.data, beta.ctypes.data, K.ctypes.data, G.ctypes.data, C.byref(ms)),
              "vbh_fmllr_accumulate")
        return beta, K, G, ms.value

    def fmllr_update_full(self, n_spk, xform=None, min_count=500.0, num_iters=40):
        """the row-by-row fMLLR update on the device (vbh_fmllr_update_full) from the statistics the context's last fmllr_stats call left
        there: returns (transforms [S, D, D+1] float32, objective improvements [S] float32, kernel ms); xform: the initial transforms
        (default: unit)"""
        D = self.dim
        x = np.tile(np.hstack([np.eye(D, dtype=np.float32), np.zeros((D, 1), np.float32)]), (n_spk, 1, 1)) if xform is None else np.array(xform, np.float32)
        x = np.ascontiguousarray(x.reshape(n_spk, D, D + 1))
        impr = np.zeros(n_spk, np.float32)
        ms = C.c_float()
        fn = self.lib.vbh_fmllr_update_full
        fn.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
        check(fn(self.ctx.h, D, int(n_spk), float(min_count), int(num_iters), x.ctypes.data, impr.ctypes.data, C.byref(ms)), "vbh_fmllr_update_full")
        return x, impr, ms.value

    def close(self):
        if self.h:
            self.lib.vbh_gmm_destroy(self.h)
            self.h = None


class Graph:
    """HCLG as ConstFst arrays (vbh_graph_*)."""

    def __init__(self, ctx, final, arc_offset, arcs, start=0):
        self.ctx, self.lib = ctx, ctx.lib
        final = np.ascontiguousarray(final, np.float32)
        off = np.ascontiguousarray(arc_offset, np.int64)
        arcs = np.ascontiguousarray(arcs)
        assert arcs.dtype.itemsize == 16
        self.num_states, self.num_arcs, self.start = len(final), len(arcs), start
        h = C.c_void_p()
        check(self.lib.vbh_graph_create(ctx.h, len(final), int(start), _f32p(final), _i64p(off),
                                        arcs.ctypes.data_as(C.POINTER(_lib.Arc)), C.byref(h)), "vbh_graph_create")
        self.h = h

    def close(self):
        if self.h:
            self.lib.vbh_graph_destroy(self.h)
            self.h = None


def _arr(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype).copy()


class LatticeDecoder:
    """LatticeFasterDecoder over a batch of utterances (vbh_decoder_*, vbh_decode_batch)."""

    def __init__(self, ctx, graph, max_batch, max_frames, beam=13.0, max_active=7000, min_active=200,
                 lattice_beam=6.0, prune_interval=25, beam_delta=0.5, acoustic_scale=0.083333, share_slots_with=None, admission="ordered"):
        """share_slots_with: another LatticeDecoder of the same device whose pool of search workspaces this one uses too;
        admission: "ordered" (the reference's online tightening of next_cutoff in a canonical order, vbh_decode_cfg::admission) or
        "exact" (tot <= the frame's final next_cutoff)"""
        self.ctx, self.lib, self.graph = ctx, ctx.lib, graph
        c = _lib.DecodeCfg(beam, max_active, min_active, lattice_beam, prune_interval, beam_delta, acoustic_scale, _lib.ADMISSION[admission])
        self.cfg = c
        h = C.c_void_p()
        self.lib.vbh_decoder_create_shared.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(_lib.DecodeCfg), C.c_int32, C.c_int32, C.c_void_p,
                                                       C.POINTER(C.c_void_p)]
        check(self.lib.vbh_decoder_create_shared(ctx.h, graph.h, C.byref(c), int(max_batch), int(max_frames),
                                                 share_slots_with.h if share_slots_with is not None else None, C.byref(h)),
              "vbh_decoder_create")
        self.h = h
        self.n_utt = 0

    @property
    def layout(self):
        """1: one 1024-thread workgroup per CU; 2: two 512-thread workgroups per CU (light searches) -- vbh_decoder_layout"""
        self.lib.vbh_decoder_layout.argtypes = [C.c_void_p]
        return int(self.lib.vbh_decoder_layout(self.h))

    def decode(self, d_loglikes, frame_offsets, tid2pdf):
        fo = np.ascontiguousarray(frame_offsets, np.int64)
        t2p = np.ascontiguousarray(tid2pdf, np.int32)
        self.n_utt = len(fo) - 1
        check(self.lib.vbh_decode_batch(self.h, d_loglikes.data_ptr(), int(d_loglikes.shape[1]), _i64p(fo), self.n_utt,
                                        _i32p(t2p), len(t2p) - 1), "vbh_decode_batch")

    def decode_begin(self, d_loglikes, frame_offsets, tid2pdf):
        """enqueue the search and return; d_loglikes must stay alive until decode_end()"""
        fo = np.ascontiguousarray(frame_offsets, np.int64)
        t2p = np.ascontiguousarray(tid2pdf, np.int32)
        self.n_utt = len(fo) - 1
        self._pending = d_loglikes
        check(self.lib.vbh_decode_batch_begin(self.h, d_loglikes.data_ptr(), int(d_loglikes.shape[1]), _i64p(fo), self.n_utt,
                                              _i32p(t2p), len(t2p) - 1), "vbh_decode_batch_begin")

    def decode_end(self):
        check(self.lib.vbh_decode_batch_end(self.h), "vbh_decode_batch_end")
        self._pending = None

    def result(self, u, arrays=True):
        r = _lib.Lattice()
        check(self.lib.vbh_decoder_result(self.h, int(u), C.byref(r)), "vbh_decoder_result")
        out = dict(ok=r.ok, reached_final=r.reached_final, num_frames=r.num_frames,
                   final_relative_cost=r.final_relative_cost,
                   ali=_arr(r.ali, r.num_ali, np.int32), words=_arr(r.words, r.num_words, np.int32),
                   best_graph=r.best_graph, best_acoustic=r.best_acoustic, num_toks=r.num_toks, num_links=r.num_links,
                   n_tokens_expanded=r.n_tokens_expanded, n_arcs_expanded=r.n_arcs_expanded,
                   n_tokens_created=r.n_tokens_created, n_links_created=r.n_links_created,
                   n_frames_maxactive_bound=r.n_frames_maxactive_bound,
                   n_frames_minactive_bound=r.n_frames_minactive_bound)
        if arrays:
            out.update(
                tok_frame=_arr(r.tok_frame, r.num_toks, np.int32), tok_state=_arr(r.tok_state, r.num_toks, np.int32),
                tok_cost=_arr(r.tok_cost, r.num_toks, np.float32), tok_extra=_arr(r.tok_extra, r.num_toks, np.float32),
                tok_final=_arr(r.tok_final, r.num_toks, np.float32),
                link_src=_arr(r.link_src, r.num_links, np.int32), link_dst=_arr(r.link_dst, r.num_links, np.int32),
                link_ilabel=_arr(r.link_ilabel, r.num_links, np.int32), link_olabel=_arr(r.link_olabel, r.num_links, np.int32),
                link_graph=_arr(r.link_graph, r.num_links, np.float32), link_acoustic=_arr(r.link_acoustic, r.num_links, np.float32))
        return out

    def phase_cycles(self, u):
        out = (C.c_int64 * 8)()
        check(self.lib.vbh_decoder_phase_cycles(self.h, int(u), out), "vbh_decoder_phase_cycles")
        return list(out)

    def diag(self, u):
        out = (C.c_int64 * 16)()
        check(self.lib.vbh_decoder_diag(self.h, int(u), out), "vbh_decoder_diag")
        return list(out)

    def timeline(self, u):
        """(start, end) of the utterance's search in 100 MHz device ticks, and the workgroup that ran it"""
        out = (C.c_int64 * 3)()
        check(self.lib.vbh_decoder_utt_timeline(self.h, int(u), out), "vbh_decoder_utt_timeline")
        return int(out[0]), int(out[1]), int(out[2])

    def times(self):
        """ms of the last decode(): search kernel, pool copy, host lattice finishing, whole call"""
        out = (C.c_float * 4)()
        check(self.lib.vbh_decoder_times(self.h, out), "vbh_decoder_times")
        return dict(kernel_ms=out[0], copy_ms=out[1], host_ms=out[2], total_ms=out[3])

    def batch_stats(self):
        """of the last decode(): utterances searched again in a big slot, output pool enlargements, utterances given up (ok = 0)"""
        out = (C.c_int32 * 3)()
        self.lib.vbh_decoder_batch_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        check(self.lib.vbh_decoder_batch_stats(self.h, out), "vbh_decoder_batch_stats")
        return dict(retried_in_big_slot=out[0], pool_grown=out[1], failed=out[2])

    def close(self):
        if self.h:
            self.lib.vbh_decoder_destroy(self.h)
            self.h = None


class FasterDecoder:
    """FasterDecoder (1-best Viterbi beam search) over a Graph (vbh_faster_*)."""

    def __init__(self, ctx, graph, max_batch, max_frames, beam=16.0, max_active=2147483647, min_active=20, beam_delta=0.5,
                 acoustic_scale=0.083333, admission="ordered"):
        self.ctx, self.lib, self.graph = ctx, ctx.lib, graph
        cfg = _lib.FasterCfg(beam, int(max_active), int(min_active), beam_delta, acoustic_scale, _lib.ADMISSION[admission])
        h = C.c_void_p()
        check(self.lib.vbh_faster_decoder_create(ctx.h, graph.h, C.byref(cfg), int(max_batch), int(max_frames), C.byref(h)),
              "vbh_faster_decoder_create")
        self.h = h
        self.n_utt = 0

    def decode(self, d_loglikes, frame_offsets, tid2pdf):
        fo = np.ascontiguousarray(frame_offsets, np.int64)
        t2p = np.ascontiguousarray(tid2pdf, np.int32)
        self.n_utt = len(fo) - 1
        check(self.lib.vbh_faster_decode_batch(self.h, d_loglikes.data_ptr(), int(d_loglikes.shape[1]), _i64p(fo), self.n_utt,
                                               _i32p(t2p), len(t2p) - 1), "vbh_faster_decode_batch")

    def result(self, u):
        r = _lib.BestPath()
        check(self.lib.vbh_faster_decoder_result(self.h, int(u), C.byref(r)), "vbh_faster_decoder_result")
        return dict(ok=r.ok, reached_final=r.reached_final, num_frames=r.num_frames,
                    ali=_arr(r.ali, r.num_ali, np.int32), words=_arr(r.words, r.num_words, np.int32),
                    best_graph=r.best_graph, best_acoustic=r.best_acoustic, n_tokens=r.n_tokens)

    def close(self):
        if self.h:
            self.lib.vbh_faster_decoder_destroy(self.h)
            self.h = None
