"""The C-ABI library loads and exports every entry point include/voicebridge_hip.h declares
(no compute calls: this runs without a GPU), and refuses to run without one."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "voicebridge_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vbh_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from voicebridge_amd import _lib, build
    build.build(verbose=False)
    return _lib.load()


def test_every_declared_symbol_is_exported(lib):
    from voicebridge_amd import _lib
    names = declared_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), "missing export: " + n
    assert sorted(_lib.SYMBOLS) == names       # the Python binding tracks the header


def test_struct_layouts_match_the_header(lib):
    from voicebridge_amd import _lib
    assert C.sizeof(_lib.MfccOpts) == 22 * 4
    assert C.sizeof(_lib.FeatPostCfg) == 12 * 4
    assert C.sizeof(_lib.Arc) == 16               # fst::ArcTpl<TropicalWeight>, arc.h:27-36
    assert C.sizeof(_lib.DecodeCfg) == 8 * 4
    assert lib.vbh_abi_version() == 4      # 2: vbh_best_path grew (r02), vbh_mfcc_opts.dither is applied; 3: vbh_decode_cfg.admission (r04); 4: vbh_faster_cfg.admission (r05)
    o = _lib.MfccOpts()
    lib.vbh_mfcc_opts_default(C.byref(o))
    # reference defaults (feature-window.h:52-62, feature-mfcc.h:50-59) except dither
    assert (o.samp_freq, o.frame_shift_ms, o.frame_length_ms, o.num_bins, o.num_ceps) == (16000.0, 10.0, 25.0, 23, 13)
    assert o.dither == 0.0 and abs(o.preemph_coeff - 0.97) < 1e-7 and o.window_type == 2
    assert lib.vbh_mfcc_num_frames(C.byref(o), 480000) == 2998   # 30 s utterance, SURVEY §8
    assert lib.vbh_mfcc_num_frames(C.byref(o), 399) == 0
    d = _lib.DecodeCfg()
    lib.vbh_decode_cfg_default(C.byref(d))
    assert (d.beam, d.max_active, d.min_active, d.lattice_beam, d.prune_interval, d.admission) == (13.0, 7000, 200, 6.0, 25, 0)


def test_no_cpu_fallback(lib):
    """without a GPU the product path must fail loudly"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    assert lib.vbh_ctx_create(0, C.byref(h)) == -1
    assert b"no HIP device" in lib.vbh_last_error()
    from voicebridge_amd import api, _lib
    with pytest.raises(_lib.VbhError):
        api.Context(0)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "voicebridge_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".cc")):
                src = open(os.path.join(dp, f), errors="ignore").read()
                bad = re.search(r'#\s*include[^\n]*oracle|^\s*(from|import)\s+oracle|libvb_oracle|dlopen[^\n]*oracle', src, flags=re.M)
                assert bad is None, (os.path.join(dp, f), bad.group(0))


def test_search_layout_is_chosen_by_the_configuration(lib, monkeypatch):
    """vbh_decode_layout_for (no GPU needed): light searches (max_active <= 4000) get the two-workgroups-per-CU layout, the
    defaults the one-workgroup layout; VBH_DECODE_LAYOUT overrides either way"""
    from voicebridge_amd import _lib
    monkeypatch.delenv("VBH_DECODE_LAYOUT", raising=False)
    cfg = _lib.DecodeCfg()
    lib.vbh_decode_cfg_default(C.byref(cfg))
    lib.vbh_decode_layout_for.argtypes = [C.POINTER(_lib.DecodeCfg)]
    assert cfg.max_active == 7000 and lib.vbh_decode_layout_for(C.byref(cfg)) == 1
    cfg.max_active = 2000; cfg.beam = 10.0           # DecodeFmllr's first pass
    assert lib.vbh_decode_layout_for(C.byref(cfg)) == 2
    cfg.max_active = 4001
    assert lib.vbh_decode_layout_for(C.byref(cfg)) == 1
    monkeypatch.setenv("VBH_DECODE_LAYOUT", "2")
    assert lib.vbh_decode_layout_for(C.byref(cfg)) == 2
    monkeypatch.setenv("VBH_DECODE_LAYOUT", "1")
    cfg.max_active = 500
    assert lib.vbh_decode_layout_for(C.byref(cfg)) == 1
