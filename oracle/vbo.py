"""ctypes binding of oracle/libvb_oracle.so (the plain-C CPU restatement).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never from voicebridge_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

WIN = {"hanning": 0, "hamming": 1, "povey": 2, "rectangular": 3, "blackman": 4}


class MfccOpts(C.Structure):
    # must mirror vbo_mfcc_opts in vb_oracle.h (and vbh_mfcc_opts in include/voicebridge_hip.h)
    _fields_ = [
        ("samp_freq", C.c_float), ("frame_shift_ms", C.c_float), ("frame_length_ms", C.c_float),
        ("dither", C.c_float), ("preemph_coeff", C.c_float), ("remove_dc_offset", C.c_int32),
        ("window_type", C.c_int32), ("round_to_power_of_two", C.c_int32),
        ("blackman_coeff", C.c_float), ("snip_edges", C.c_int32),
        ("num_bins", C.c_int32), ("low_freq", C.c_float), ("high_freq", C.c_float),
        ("vtln_low", C.c_float), ("vtln_high", C.c_float), ("htk_mode", C.c_int32),
        ("num_ceps", C.c_int32), ("use_energy", C.c_int32), ("energy_floor", C.c_float),
        ("raw_energy", C.c_int32), ("cepstral_lifter", C.c_float), ("htk_compat", C.c_int32),
    ]


class Arc(C.Structure):
    _fields_ = [("ilabel", C.c_int32), ("olabel", C.c_int32), ("weight", C.c_float),
                ("nextstate", C.c_int32)]


ARC_DTYPE = np.dtype([("ilabel", "<i4"), ("olabel", "<i4"), ("weight", "<f4"), ("nextstate", "<i4")])


class Fst(C.Structure):
    _fields_ = [("num_states", C.c_int32), ("start", C.c_int32), ("final", C.POINTER(C.c_float)),
                ("arc_offset", C.POINTER(C.c_int64)), ("arcs", C.POINTER(Arc))]


class LatCfg(C.Structure):
    _fields_ = [("beam", C.c_float), ("max_active", C.c_int32), ("min_active", C.c_int32),
                ("lattice_beam", C.c_float), ("prune_interval", C.c_int32),
                ("beam_delta", C.c_float), ("hash_ratio", C.c_float), ("prune_scale", C.c_float),
                ("exact_cutoff", C.c_int32), ("adm_bucket_bits", C.c_int32)]


class LatResult(C.Structure):
    _fields_ = [
        ("ok", C.c_int32), ("reached_final", C.c_int32), ("num_frames", C.c_int32),
        ("final_relative_cost", C.c_float), ("final_best_cost", C.c_float),
        ("num_toks", C.c_int32), ("tok_frame", C.POINTER(C.c_int32)), ("tok_state", C.POINTER(C.c_int32)),
        ("tok_cost", C.POINTER(C.c_float)), ("tok_extra", C.POINTER(C.c_float)),
        ("tok_final", C.POINTER(C.c_float)),
        ("num_links", C.c_int32), ("link_src", C.POINTER(C.c_int32)), ("link_dst", C.POINTER(C.c_int32)),
        ("link_ilabel", C.POINTER(C.c_int32)), ("link_olabel", C.POINTER(C.c_int32)),
        ("link_graph", C.POINTER(C.c_float)), ("link_acoustic", C.POINTER(C.c_float)),
        ("num_ali", C.c_int32), ("ali", C.POINTER(C.c_int32)),
        ("num_words", C.c_int32), ("words", C.POINTER(C.c_int32)),
        ("best_graph", C.c_float), ("best_acoustic", C.c_float),
        ("n_tok_created", C.c_int64), ("n_links_created", C.c_int64),
        ("n_frames_maxactive_bound", C.c_int32), ("n_frames_minactive_bound", C.c_int32),
        ("cost_offsets", C.POINTER(C.c_float)),
    ]


class FasterCfg(C.Structure):
    _fields_ = [("beam", C.c_float), ("max_active", C.c_int32), ("min_active", C.c_int32),
                ("beam_delta", C.c_float), ("hash_ratio", C.c_float), ("exact_cutoff", C.c_int32), ("adm_bucket_bits", C.c_int32)]


class FasterResult(C.Structure):
    _fields_ = [("ok", C.c_int32), ("reached_final", C.c_int32), ("num_frames", C.c_int32),
                ("num_ali", C.c_int32), ("ali", C.POINTER(C.c_int32)),
                ("num_words", C.c_int32), ("words", C.POINTER(C.c_int32)),
                ("best_graph", C.c_float), ("best_acoustic", C.c_float)]


def build():
    """Compile libvb_oracle.so (gcc, seconds)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "libvb_oracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libvb_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.vbo_log_sum_exp.restype = C.c_float
        _LIB.vbo_compress_size.restype = C.c_size_t
        _LIB.vbo_first_sample_of_frame.restype = C.c_int64
    return _LIB


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def mfcc_opts(**kw):
    o = MfccOpts()
    lib().vbo_mfcc_opts_default(C.byref(o))
    o.dither = 0.0
    for k, v in kw.items():
        if k == "window_type" and isinstance(v, str):
            v = WIN[v]
        if not hasattr(o, k):
            raise KeyError(k)
        setattr(o, k, v)
    return o


def num_frames(n, opts):
    return lib().vbo_num_frames(C.c_int64(n), C.byref(opts))


def padded_window_size(opts):
    return lib().vbo_padded_window_size(C.byref(opts))


def mfcc(wave, opts, vtln_warp=1.0, stages=False):
    wave = _f32(wave)
    T = num_frames(len(wave), opts)
    out = np.zeros((T, opts.num_ceps), np.float32)
    if stages:
        P = padded_window_size(opts)
        win = np.zeros((T, P), np.float32)
        ps = np.zeros((T, P // 2 + 1), np.float32)
        mel = np.zeros((T, opts.num_bins), np.float32)
        rc = lib().vbo_mfcc(C.byref(opts), _p(wave, C.c_float), C.c_int64(len(wave)), C.c_float(vtln_warp),
                            _p(out, C.c_float), _p(win, C.c_float), _p(ps, C.c_float), _p(mel, C.c_float))
        assert rc == 0, rc
        return out, win, ps, mel
    rc = lib().vbo_mfcc(C.byref(opts), _p(wave, C.c_float), C.c_int64(len(wave)), C.c_float(vtln_warp),
                        _p(out, C.c_float), None, None, None)
    assert rc == 0, rc
    return out


def add_deltas(feats, order=2, window=2):
    feats = _f32(feats)
    T, D = feats.shape
    out = np.zeros((T, D * (order + 1)), np.float32)
    lib().vbo_add_deltas(order, window, _p(feats, C.c_float), T, D, _p(out, C.c_float))
    return out


def splice(feats, left=4, right=4):
    feats = _f32(feats)
    T, D = feats.shape
    out = np.zeros((T, D * (left + right + 1)), np.float32)
    lib().vbo_splice(_p(feats, C.c_float), T, D, left, right, _p(out, C.c_float))
    return out


def acc_cmvn_stats(feats):
    feats = _f32(feats)
    T, D = feats.shape
    st = np.zeros((2, D + 1), np.float64)
    lib().vbo_acc_cmvn_stats(_p(feats, C.c_float), T, D, _p(st, C.c_double))
    return st


def apply_cmvn(feats, stats, norm_vars=False):
    out = _f32(feats).copy()
    T, D = out.shape
    st = np.ascontiguousarray(stats, np.float64)
    rc = lib().vbo_apply_cmvn(_p(st, C.c_double), int(norm_vars), _p(out, C.c_float), T, D)
    assert rc == 0
    return out


def transform_feats(feats, xform):
    feats = _f32(feats)
    xform = _f32(xform)
    T, D = feats.shape
    R, Cc = xform.shape
    out = np.zeros((T, R), np.float32)
    rc = lib().vbo_transform_feats(_p(xform, C.c_float), R, Cc, _p(feats, C.c_float), T, D, _p(out, C.c_float))
    assert rc == 0
    return out


def compress(mat):
    mat = _f32(mat)
    r, c = mat.shape
    blob = np.zeros(lib().vbo_compress_size(r, c), np.uint8)
    rc = lib().vbo_compress(_p(mat, C.c_float), r, c, _p(blob, C.c_uint8))
    assert rc == 0
    return blob


def uncompress(blob):
    blob = np.ascontiguousarray(blob, np.uint8)
    hdr = np.frombuffer(blob[:16].tobytes(), dtype="<i4")
    r, c = int(hdr[2]), int(hdr[3])
    out = np.zeros((r, c), np.float32)
    lib().vbo_uncompress(_p(blob, C.c_uint8), _p(out, C.c_float))
    return out


def gmm_gconsts(weights, miv, iv):
    weights, miv, iv = _f32(weights), _f32(miv), _f32(iv)
    G, D = miv.shape
    out = np.zeros(G, np.float32)
    rc = lib().vbo_gmm_gconsts(G, D, _p(weights, C.c_float), _p(miv, C.c_float), _p(iv, C.c_float), _p(out, C.c_float))
    assert rc >= 0
    return out


def gmm_loglikes(offs, gconsts, miv, iv, feats):
    offs = np.ascontiguousarray(offs, np.int32)
    gconsts, miv, iv, feats = _f32(gconsts), _f32(miv), _f32(iv), _f32(feats)
    P = len(offs) - 1
    T, D = feats.shape
    out = np.zeros((T, P), np.float32)
    lib().vbo_gmm_loglikes(P, _p(offs, C.c_int32), D, _p(gconsts, C.c_float), _p(miv, C.c_float),
                           _p(iv, C.c_float), _p(feats, C.c_float), T, _p(out, C.c_float))
    return out


def make_fst(final, arc_offset, arcs, start=0):
    """final [S] f32, arc_offset [S+1] i64, arcs structured ARC_DTYPE [A]; returns (Fst, keepalive)."""
    final = _f32(final)
    arc_offset = np.ascontiguousarray(arc_offset, np.int64)
    arcs = np.ascontiguousarray(arcs, ARC_DTYPE)
    f = Fst(len(final), start, _p(final, C.c_float), _p(arc_offset, C.c_int64),
            arcs.ctypes.data_as(C.POINTER(Arc)))
    return f, (final, arc_offset, arcs)


def _arr(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype).copy()


def decode_lattice_faster(fst, loglikes, tid2pdf, acwt=0.083333, beam=13.0, max_active=7000,
                          min_active=200, lattice_beam=6.0, prune_interval=25, beam_delta=0.5,
                          hash_ratio=2.0, exact_cutoff=False, adm_bucket_bits=None):
    f, keep = fst
    cfg = LatCfg()
    lib().vbo_lat_cfg_default(C.byref(cfg))
    cfg.beam, cfg.max_active, cfg.min_active = beam, max_active, min_active
    cfg.lattice_beam, cfg.prune_interval, cfg.beam_delta = lattice_beam, prune_interval, beam_delta
    cfg.hash_ratio, cfg.exact_cutoff = hash_ratio, int(exact_cutoff)
    if adm_bucket_bits is not None:
        cfg.adm_bucket_bits = int(adm_bucket_bits)
    ll = _f32(loglikes)
    T, P = ll.shape
    t2p = np.ascontiguousarray(tid2pdf, np.int32)
    res = LatResult()
    rc = lib().vbo_decode_lattice_faster(C.byref(f), C.byref(cfg), _p(ll, C.c_float), T, P,
                                         _p(t2p, C.c_int32), len(t2p) - 1, C.c_float(acwt), C.byref(res))
    assert rc == 0
    out = dict(
        ok=res.ok, reached_final=res.reached_final, num_frames=res.num_frames,
        final_relative_cost=res.final_relative_cost, final_best_cost=res.final_best_cost,
        tok_frame=_arr(res.tok_frame, res.num_toks, np.int32), tok_state=_arr(res.tok_state, res.num_toks, np.int32),
        tok_cost=_arr(res.tok_cost, res.num_toks, np.float32), tok_extra=_arr(res.tok_extra, res.num_toks, np.float32),
        tok_final=_arr(res.tok_final, res.num_toks, np.float32),
        link_src=_arr(res.link_src, res.num_links, np.int32), link_dst=_arr(res.link_dst, res.num_links, np.int32),
        link_ilabel=_arr(res.link_ilabel, res.num_links, np.int32), link_olabel=_arr(res.link_olabel, res.num_links, np.int32),
        link_graph=_arr(res.link_graph, res.num_links, np.float32), link_acoustic=_arr(res.link_acoustic, res.num_links, np.float32),
        ali=_arr(res.ali, res.num_ali, np.int32), words=_arr(res.words, res.num_words, np.int32),
        best_graph=res.best_graph, best_acoustic=res.best_acoustic,
        n_tok_created=res.n_tok_created, n_links_created=res.n_links_created,
        n_frames_maxactive_bound=res.n_frames_maxactive_bound,
        n_frames_minactive_bound=res.n_frames_minactive_bound,
        cost_offsets=_arr(res.cost_offsets, res.num_frames, np.float32),
    )
    lib().vbo_lat_result_free(C.byref(res))
    return out


def decode_faster(fst, loglikes, tid2pdf, acwt=0.083333, beam=13.0, max_active=7000, min_active=20,
                  beam_delta=0.5, hash_ratio=2.0, exact_cutoff=False, adm_bucket_bits=6):
    """exact_cutoff: False / 0 = the reference's HashList order; True / 1 = order-independent admission; 3 = ordered admission"""
    f, keep = fst
    cfg = FasterCfg(beam, max_active, min_active, beam_delta, hash_ratio, int(exact_cutoff), adm_bucket_bits)
    ll = _f32(loglikes)
    T, P = ll.shape
    t2p = np.ascontiguousarray(tid2pdf, np.int32)
    res = FasterResult()
    rc = lib().vbo_decode_faster(C.byref(f), C.byref(cfg), _p(ll, C.c_float), T, P, _p(t2p, C.c_int32),
                                 len(t2p) - 1, C.c_float(acwt), C.byref(res))
    assert rc == 0
    out = dict(ok=res.ok, reached_final=res.reached_final, num_frames=res.num_frames,
               ali=_arr(res.ali, res.num_ali, np.int32), words=_arr(res.words, res.num_words, np.int32),
               best_graph=res.best_graph, best_acoustic=res.best_acoustic)
    lib().vbo_faster_result_free(C.byref(res))
    return out
