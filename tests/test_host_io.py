"""File-format layer (voicebridge_amd/host/kaldi_io.cc) against files written by the REFERENCE's
own writers (tests/golden/files.npz, produced by oracle/ref_harness.cc `files`).  CPU only."""
import ctypes as C
import os

import numpy as np
import pytest

from tests.util import load


@pytest.fixture(scope="module")
def host():
    from voicebridge_amd import build
    build.build(verbose=False)
    lib = C.CDLL(build.HOST_LIB)
    lib.vbio_last_error.restype = C.c_char_p
    return lib


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    d = tmp_path_factory.mktemp("vbfiles")
    g = load("files")
    for k in g.files:
        if k.startswith("file_"):
            name = {"file_HCLG_fst": "HCLG.fst", "file_feats_ark": "feats.ark", "file_feats_scp": "feats.scp",
                    "file_feats_plain_ark": "feats_plain.ark", "file_feats_plain_scp": "feats_plain.scp",
                    "file_final_mdl": "final.mdl", "file_test_wav": "test.wav"}[k]
            (d / name).write_bytes(g[k].tobytes())
    # the scp files point into the harness's temp dir: re-point them here
    for scp, ark in (("feats.scp", "feats.ark"), ("feats_plain.scp", "feats_plain.ark")):
        lines = []
        for line in (d / scp).read_text().splitlines():
            key, rest = line.split(" ", 1)
            off = rest.rsplit(":", 1)[1]
            lines.append("%s %s:%s" % (key, d / ark, off))
        (d / scp).write_text("\n".join(lines) + "\n")
    return d, g


def read_fm_ark(path):
    """minimal parser of a binary archive of "FM " matrices"""
    raw = open(path, "rb").read()
    out, pos = {}, 0
    while pos < len(raw):
        sp = raw.index(b" ", pos)
        key = raw[pos:sp].decode()
        pos = sp + 1
        assert raw[pos:pos + 2] == b"\0B" and raw[pos + 2:pos + 5] == b"FM "
        pos += 5
        assert raw[pos] == 4
        rows = int.from_bytes(raw[pos + 1:pos + 5], "little")
        assert raw[pos + 5] == 4
        cols = int.from_bytes(raw[pos + 6:pos + 10], "little")
        pos += 10
        out[key] = np.frombuffer(raw[pos:pos + 4 * rows * cols], "<f4").reshape(rows, cols)
        pos += 4 * rows * cols
    return out


def test_compressed_archive_is_byte_identical_to_the_reference(host, files):
    d, g = files
    # reference wrote the same matrices plain and compressed: compressing its plain archive
    # must give its compressed archive byte for byte (CompressedMatrix is integer work)
    rc = host.vbio_copy_feats(("ark:%s" % (d / "feats_plain.ark")).encode(), ("ark:%s" % (d / "mine_cm.ark")).encode(), 1)
    assert rc == 0, host.vbio_last_error()
    assert (d / "mine_cm.ark").read_bytes() == (d / "feats.ark").read_bytes()


def test_read_compressed_and_scp(host, files):
    d, g = files
    for rspec in ("ark:%s" % (d / "feats.ark"), "scp:%s" % (d / "feats.scp"), "scp,p:%s" % (d / "feats.scp")):
        out = d / "back.ark"
        assert host.vbio_copy_feats(rspec.encode(), ("ark:%s" % out).encode(), 0) == 0, host.vbio_last_error()
        m = read_fm_ark(out)
        assert sorted(m) == ["utt1", "utt2"]
        # lossy codec: within one quantisation step of the original, identical for both keys
        assert np.abs(m["utt1"] - g["feats"]).max() < 0.06
        np.testing.assert_array_equal(m["utt1"], m["utt2"])
    plain = d / "plain_back.ark"
    assert host.vbio_copy_feats(("scp:%s" % (d / "feats_plain.scp")).encode(), ("ark,scp:%s,%s" % (plain, d / "plain_back.scp")).encode(), 0) == 0
    np.testing.assert_array_equal(read_fm_ark(plain)["utt2"], g["feats"])
    assert (d / "plain_back.ark").read_bytes() == (d / "feats_plain.ark").read_bytes()      # same bytes as the reference writer
    lines = (d / "plain_back.scp").read_text().split("\n")
    assert lines[0].startswith("utt1 ") and lines[0].endswith(":5")


def test_wave_reader(host, files):
    d, g = files
    n, sr, ch = C.c_longlong(), C.c_float(), C.c_int()
    buf = np.zeros(10000, np.float32)
    rc = host.vbio_read_wave(str(d / "test.wav").encode(), buf.ctypes.data_as(C.POINTER(C.c_float)), 10000, C.byref(n), C.byref(sr), C.byref(ch))
    assert rc == 0, host.vbio_last_error()
    assert (n.value, sr.value, ch.value) == (len(g["wave"]), 16000.0, 1)
    np.testing.assert_array_equal(buf[:n.value], g["wave"])


def test_model_reader(host, files):
    d, g = files
    np_, dim, nt, ng = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    rc = host.vbio_model_load(str(d / "final.mdl").encode(), C.byref(np_), C.byref(dim), C.byref(nt), C.byref(ng))
    assert rc == 0, host.vbio_last_error()
    assert [np_.value, dim.value] == g["mdl_meta"].tolist()
    assert nt.value == len(g["tid2pdf"]) - 1 and ng.value == len(g["mdl_weights"])
    t2p = np.zeros(nt.value + 1, np.int32); offs = np.zeros(np_.value + 1, np.int32)
    w = np.zeros(ng.value, np.float32); miv = np.zeros(ng.value * dim.value, np.float32); iv = np.zeros_like(miv)
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    host.vbio_model_get(P(t2p, C.c_int), P(offs, C.c_int), P(w, C.c_float), P(miv, C.c_float), P(iv, C.c_float))
    np.testing.assert_array_equal(t2p[1:], g["tid2pdf"][1:])       # TransitionIdToPdf, transition-model.h:324-328
    np.testing.assert_array_equal(offs, g["mdl_offs"])
    np.testing.assert_array_equal(w, g["mdl_weights"])
    np.testing.assert_array_equal(miv, g["mdl_miv"])
    np.testing.assert_array_equal(iv, g["mdl_iv"])


def test_text_mode_model(host, files, tmp_path):
    """the same model in text form, as the reference's gmm-copy --binary=false wrote it (tests/golden/model_text.npz): the topology in
    its own grammar (K/hmm/hmm-topology.cc:41-127), tuples, matrices in brackets; numbers as printed, i.e. to six digits"""
    d, g = files
    (tmp_path / "final_text.mdl").write_bytes(load("model_text")["file_final_text_mdl"].tobytes())
    np_, dim, nt, ng = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    assert host.vbio_model_load(str(tmp_path / "final_text.mdl").encode(), C.byref(np_), C.byref(dim), C.byref(nt), C.byref(ng)) == 0, host.vbio_last_error()
    assert [np_.value, dim.value] == g["mdl_meta"].tolist() and nt.value == len(g["tid2pdf"]) - 1 and ng.value == len(g["mdl_weights"])
    t2p = np.zeros(nt.value + 1, np.int32); offs = np.zeros(np_.value + 1, np.int32)
    w = np.zeros(ng.value, np.float32); miv = np.zeros(ng.value * dim.value, np.float32); iv = np.zeros_like(miv)
    P = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    host.vbio_model_get(P(t2p, C.c_int), P(offs, C.c_int), P(w, C.c_float), P(miv, C.c_float), P(iv, C.c_float))
    np.testing.assert_array_equal(t2p[1:], g["tid2pdf"][1:])
    np.testing.assert_array_equal(offs, g["mdl_offs"])
    np.testing.assert_allclose(w, g["mdl_weights"], rtol=2e-5)
    np.testing.assert_allclose(miv, g["mdl_miv"], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(iv, g["mdl_iv"], rtol=2e-5)


def test_fst_reader(host, files):
    d, g = files
    ns, na, st = C.c_int(), C.c_longlong(), C.c_int()
    rc = host.vbio_fst_load(str(d / "HCLG.fst").encode(), C.byref(ns), C.byref(na), C.byref(st))
    assert rc == 0, host.vbio_last_error()
    assert (ns.value, na.value, st.value) == (len(g["fst_final"]), len(g["fst_arcs_w"]), 0)
    fin = np.zeros(ns.value, np.float32); off = np.zeros(ns.value + 1, np.int64)
    arcs = np.zeros(na.value, np.dtype([("il", "<i4"), ("ol", "<i4"), ("w", "<f4"), ("ns", "<i4")]))
    host.vbio_fst_get(fin.ctypes.data_as(C.POINTER(C.c_float)), off.ctypes.data_as(C.POINTER(C.c_longlong)), arcs.ctypes.data_as(C.c_void_p))
    np.testing.assert_array_equal(fin, g["fst_final"])
    np.testing.assert_array_equal(off, g["fst_off"])
    np.testing.assert_array_equal(np.stack([arcs["il"], arcs["ol"], arcs["ns"]], 1), g["fst_arcs_i"])
    np.testing.assert_array_equal(arcs["w"], g["fst_arcs_w"])


def test_errors_are_reported_not_thrown(host, tmp_path):
    assert host.vbio_copy_feats(b"ark:/nonexistent/x.ark", ("ark:%s" % (tmp_path / "o.ark")).encode(), 0) == -1
    assert b"cannot open" in host.vbio_last_error()
    bad = tmp_path / "bad.fst"
    bad.write_bytes(b"not an fst at all")
    ns, na, st = C.c_int(), C.c_longlong(), C.c_int()
    assert host.vbio_fst_load(str(bad).encode(), C.byref(ns), C.byref(na), C.byref(st)) == -1
