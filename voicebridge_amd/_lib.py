"""ctypes loader of voicebridge_amd/lib/libvoicebridge_hip.so (the C ABI of include/voicebridge_hip.h).

There is no CPU path: a missing library or a missing GPU is an error, never a fallback.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# VBH_LIB_PATH selects another build of the same library (kernel experiments, tools/variants.py)
LIB_PATH = os.environ.get("VBH_LIB_PATH") or os.path.join(HERE, "lib", "libvoicebridge_hip.so")

# every symbol include/voicebridge_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "vbh_abi_version", "vbh_last_error", "vbh_device_count", "vbh_ctx_create", "vbh_ctx_create_with_priority", "vbh_ctx_create_on_cus", "vbh_ctx_destroy", "vbh_ctx_scratch", "vbh_ctx_scratch_release",
    "vbh_ctx_sync", "vbh_ctx_wait_for", "vbh_ctx_mark", "vbh_ctx_wait_mark", "vbh_ctx_stream", "vbh_ctx_device", "vbh_dev_alloc", "vbh_dev_free", "vbh_copy_h2d",
    "vbh_copy_d2h", "vbh_memset", "vbh_timer_start", "vbh_timer_stop", "vbh_timer_elapsed_ms",
    "vbh_mfcc_opts_default", "vbh_mfcc_num_frames", "vbh_mfcc_create", "vbh_mfcc_destroy", "vbh_mfcc_compute", "vbh_mfcc_set_dither_stream", "vbh_mfcc_set_dither_ids",
    "vbh_featpost_out_dim", "vbh_featpost", "vbh_cmvn_stats",
    "vbh_gmm_create", "vbh_gmm_destroy", "vbh_gmm_num_pdfs", "vbh_gmm_dim", "vbh_gmm_get_gconsts",
    "vbh_gmm_set_lse_mode", "vbh_gmm_loglikes", "vbh_gmm_loglikes_ld", "vbh_gmm_padded_pdfs", "vbh_gmm_loglikes_pairs", "vbh_gmm_work",
    "vbh_graph_create", "vbh_graph_destroy", "vbh_decode_cfg_default", "vbh_decoder_create", "vbh_decoder_create_shared", "vbh_decoder_set_cfg", "vbh_decode_layout_for", "vbh_decoder_layout",
    "vbh_decoder_destroy", "vbh_decode_batch", "vbh_decode_batch_begin", "vbh_decode_batch_end", "vbh_decoder_result", "vbh_decoder_phase_cycles", "vbh_decoder_diag",
    "vbh_decoder_times", "vbh_decoder_utt_timeline", "vbh_decoder_batch_stats",
    "vbh_faster_cfg_default", "vbh_faster_decoder_create", "vbh_faster_decoder_create_ex", "vbh_faster_decoder_destroy", "vbh_faster_decode_batch", "vbh_faster_decode_batch_ex",
    "vbh_faster_decoder_result",
    "vbh_fmllr_accumulate", "vbh_fmllr_accumulate_gpost", "vbh_fmllr_work", "vbh_fmllr_update_full",
]


class MfccOpts(C.Structure):
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


class FeatPostCfg(C.Structure):
    _fields_ = [("in_dim", C.c_int32), ("apply_cmvn", C.c_int32), ("norm_vars", C.c_int32), ("mode", C.c_int32),
                ("delta_order", C.c_int32), ("delta_window", C.c_int32), ("splice_left", C.c_int32),
                ("splice_right", C.c_int32), ("lda_rows", C.c_int32), ("lda_cols", C.c_int32),
                ("spk_rows", C.c_int32), ("spk_cols", C.c_int32)]


class Arc(C.Structure):
    _fields_ = [("ilabel", C.c_int32), ("olabel", C.c_int32), ("weight", C.c_float), ("nextstate", C.c_int32)]


class DecodeCfg(C.Structure):
    _fields_ = [("beam", C.c_float), ("max_active", C.c_int32), ("min_active", C.c_int32),
                ("lattice_beam", C.c_float), ("prune_interval", C.c_int32), ("beam_delta", C.c_float),
                ("acoustic_scale", C.c_float), ("admission", C.c_int32)]      # admission: 0 = ordered (default), 1 = exact


ADMISSION = {"ordered": 0, "exact": 1}


class FasterCfg(C.Structure):
    _fields_ = [("beam", C.c_float), ("max_active", C.c_int32), ("min_active", C.c_int32), ("beam_delta", C.c_float),
                ("acoustic_scale", C.c_float), ("admission", C.c_int32)]      # admission: 0 = ordered (default), 1 = exact


class BestPath(C.Structure):
    _fields_ = [("ok", C.c_int32), ("reached_final", C.c_int32), ("num_frames", C.c_int32),
                ("num_ali", C.c_int32), ("ali", C.POINTER(C.c_int32)), ("num_words", C.c_int32), ("words", C.POINTER(C.c_int32)),
                ("best_graph", C.c_float), ("best_acoustic", C.c_float), ("n_tokens", C.c_int64),
                ("num_arcs", C.c_int32), ("arc_ilabel", C.POINTER(C.c_int32)), ("arc_olabel", C.POINTER(C.c_int32)),
                ("arc_graph", C.POINTER(C.c_float)), ("arc_acoustic", C.POINTER(C.c_float)), ("final_graph", C.c_float)]


class Lattice(C.Structure):
    _fields_ = [
        ("ok", C.c_int32), ("reached_final", C.c_int32), ("num_frames", C.c_int32),
        ("final_relative_cost", C.c_float),
        ("num_toks", C.c_int32), ("tok_frame", C.POINTER(C.c_int32)), ("tok_state", C.POINTER(C.c_int32)),
        ("tok_cost", C.POINTER(C.c_float)), ("tok_extra", C.POINTER(C.c_float)), ("tok_final", C.POINTER(C.c_float)),
        ("num_links", C.c_int32), ("link_src", C.POINTER(C.c_int32)), ("link_dst", C.POINTER(C.c_int32)),
        ("link_ilabel", C.POINTER(C.c_int32)), ("link_olabel", C.POINTER(C.c_int32)),
        ("link_graph", C.POINTER(C.c_float)), ("link_acoustic", C.POINTER(C.c_float)),
        ("num_ali", C.c_int32), ("ali", C.POINTER(C.c_int32)),
        ("num_words", C.c_int32), ("words", C.POINTER(C.c_int32)),
        ("best_graph", C.c_float), ("best_acoustic", C.c_float),
        ("n_tokens_expanded", C.c_int64), ("n_arcs_expanded", C.c_int64),
        ("n_tokens_created", C.c_int64), ("n_links_created", C.c_int64),
        ("n_frames_maxactive_bound", C.c_int32), ("n_frames_minactive_bound", C.c_int32),
    ]


_LIB = None


def load():
    """Load the HIP library; raises (never falls back) when it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("%s not found: build it with `python -m voicebridge_amd.build` "
                           "(there is no CPU implementation to fall back to)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.vbh_last_error.restype = C.c_char_p
    lib.vbh_ctx_stream.restype = C.c_void_p
    lib.vbh_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    for name in ("vbh_ctx_destroy", "vbh_mfcc_destroy", "vbh_gmm_destroy", "vbh_graph_destroy", "vbh_decoder_destroy"):
        if hasattr(lib, name):
            getattr(lib, name).argtypes = [C.c_void_p]
            getattr(lib, name).restype = None
    lib.vbh_mfcc_num_frames.argtypes = [C.POINTER(MfccOpts), C.c_int64]
    lib.vbh_mfcc_create.argtypes = [C.c_void_p, C.POINTER(MfccOpts), C.c_float, C.POINTER(C.c_void_p)]
    lib.vbh_mfcc_compute.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_int32, C.c_void_p,
                                     C.POINTER(C.c_int64)]
    lib.vbh_featpost.argtypes = [C.c_void_p, C.POINTER(FeatPostCfg), C.c_void_p, C.POINTER(C.c_int64), C.c_int32,
                                 C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_float),
                                 C.POINTER(C.c_float), C.c_void_p]
    lib.vbh_cmvn_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.c_int32,
                                   C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_double)]
    lib.vbh_gmm_create.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_float),
                                   C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_void_p)]
    lib.vbh_gmm_loglikes.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.vbh_gmm_loglikes_ld.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32]
    lib.vbh_gmm_padded_pdfs.argtypes = [C.c_void_p]
    lib.vbh_gmm_padded_pdfs.restype = C.c_int32
    lib.vbh_gmm_get_gconsts.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    lib.vbh_gmm_work.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.vbh_gmm_set_lse_mode.argtypes = [C.c_void_p, C.c_int32]
    lib.vbh_gmm_num_pdfs.argtypes = [C.c_void_p]
    lib.vbh_gmm_dim.argtypes = [C.c_void_p]
    lib.vbh_dev_alloc.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
    lib.vbh_dev_free.argtypes = [C.c_void_p, C.c_void_p]
    lib.vbh_copy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.vbh_copy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.vbh_memset.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t]
    lib.vbh_ctx_sync.argtypes = [C.c_void_p]
    lib.vbh_ctx_wait_for.argtypes = [C.c_void_p, C.c_void_p]
    lib.vbh_ctx_device.argtypes = [C.c_void_p]
    lib.vbh_ctx_stream.argtypes = [C.c_void_p]
    lib.vbh_timer_start.argtypes = [C.c_void_p, C.c_int]
    lib.vbh_timer_stop.argtypes = [C.c_void_p, C.c_int]
    lib.vbh_timer_elapsed_ms.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float)]
    if hasattr(lib, "vbh_graph_create"):
        lib.vbh_graph_create.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_int64),
                                         C.POINTER(Arc), C.POINTER(C.c_void_p)]
        lib.vbh_decoder_create.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(DecodeCfg), C.c_int32, C.c_int32,
                                           C.POINTER(C.c_void_p)]
        lib.vbh_decode_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.c_int32,
                                         C.POINTER(C.c_int32), C.c_int32]
        lib.vbh_decode_batch_begin.argtypes = lib.vbh_decode_batch.argtypes
        lib.vbh_decode_batch_end.argtypes = [C.c_void_p]
        lib.vbh_decoder_result.argtypes = [C.c_void_p, C.c_int32, C.POINTER(Lattice)]
        lib.vbh_decoder_phase_cycles.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]
        lib.vbh_decoder_diag.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]
        lib.vbh_faster_decoder_create.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(FasterCfg), C.c_int32, C.c_int32,
                                                  C.POINTER(C.c_void_p)]
        lib.vbh_faster_decoder_destroy.argtypes = [C.c_void_p]
        lib.vbh_faster_decode_batch.argtypes = lib.vbh_decode_batch.argtypes
        lib.vbh_faster_decoder_result.argtypes = [C.c_void_p, C.c_int32, C.POINTER(BestPath)]
        lib.vbh_faster_cfg_default.argtypes = [C.POINTER(FasterCfg)]
        lib.vbh_decoder_times.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        lib.vbh_decoder_utt_timeline.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]
    _LIB = lib
    return lib


class VbhError(RuntimeError):
    pass


def check(rc, what=""):
    if rc != 0:
        raise VbhError("%s failed: %s" % (what, load().vbh_last_error().decode()))
