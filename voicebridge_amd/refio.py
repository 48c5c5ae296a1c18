"""Writers for the reference's on-disk formats from numpy arrays: .mdl (TransitionModel + AmDiagGmm), OpenFst
VectorFst<StdArc>, feature archives, PCM-16 wav.  Used by the tests (inputs for the reference's own programs, oracle/_ref/bin)
and by bench.py's CPU baseline, which times those programs on a sample of the bench workload."""
import wave as _wave

import numpy as np


def _tok(s):
    return s.encode() + b" "


def _i32(v):
    return b"\x04" + np.int32(v).tobytes()


def _f32(v):
    return b"\x04" + np.float32(v).tobytes()


def _ivec(v):
    v = np.asarray(v, "<i4")
    return b"\x04" + np.int32(len(v)).tobytes() + v.tobytes()


def _fvec(v):
    v = np.asarray(v, "<f4")
    return b"FV " + _i32(len(v)) + v.tobytes()


def _fmat(m):
    m = np.asarray(m, "<f4")
    return b"FM " + _i32(m.shape[0]) + _i32(m.shape[1]) + m.tobytes()


def write_mdl(path, n_cd, hmm_states, offs, weights, miv, iv, self_p=0.6):
    """Binary .mdl: `n_cd` phones, each a left-to-right HMM of `hmm_states` emitting states (self-loop then forward
    transition; an int, or a list with one count per phone: one topology entry per distinct count) with the pdfs numbered
    phone by phone, state by state, i.e. the transition-ids synth.lexicon_graph() uses (pdf p: self-loop 2p+1, forward 2p+2);
    then the diagonal GMMs of those pdfs.  Readable by the reference (TransitionModel::Read, AmDiagGmm::Read) and by
    vbio::ReadModel."""
    Hs = [int(hmm_states)] * n_cd if np.isscalar(hmm_states) else [int(h) for h in hmm_states]
    assert len(Hs) == n_cd
    kinds = sorted(set(Hs))
    parts = [b"\0B" + _tok("<TransitionModel>") + _tok("<Topology>")]
    phones = list(range(1, n_cd + 1))
    parts.append(_ivec(phones) + _ivec([-1] + [kinds.index(h) for h in Hs]))
    parts.append(_i32(len(kinds)))             # one topology entry per distinct state count
    for H in kinds:
        parts.append(_i32(H + 1))
        for k in range(H):
            parts.append(_i32(k) + _i32(2) + _i32(k) + _f32(self_p) + _i32(k + 1) + _f32(1.0 - self_p))
        parts.append(_i32(-1) + _i32(0))       # final non-emitting state
    P = int(sum(Hs))
    parts.append(_tok("</Topology>") + _tok("<Triples>") + _i32(P))
    trip = np.zeros((P, 3), "<i4")
    trip[:, 0] = np.repeat(np.arange(1, n_cd + 1), Hs)
    trip[:, 1] = np.concatenate([np.arange(h) for h in Hs])
    trip[:, 2] = np.arange(P)
    rec = np.zeros((P, 3), dtype=[("s", "u1"), ("v", "<i4")])
    rec["s"] = 4
    rec["v"] = trip
    parts.append(rec.tobytes())
    parts.append(_tok("</Triples>") + _tok("<LogProbs>"))
    lp = [0.0] + [np.log(self_p), np.log(1.0 - self_p)] * P
    parts.append(_fvec(lp) + _tok("</LogProbs>") + _tok("</TransitionModel>"))
    D = miv.shape[1]
    parts.append(_tok("<DIMENSION>") + _i32(D) + _tok("<NUMPDFS>") + _i32(P))
    for p in range(P):
        a, b = int(offs[p]), int(offs[p + 1])
        parts.append(_tok("<DiagGMM>") + _tok("<WEIGHTS>") + _fvec(weights[a:b]) + _tok("<MEANS_INVVARS>") + _fmat(miv[a:b]) +
                     _tok("<INV_VARS>") + _fmat(iv[a:b]) + _tok("</DiagGMM>"))
    with open(path, "wb") as f:
        f.write(b"".join(parts))


def write_vector_fst(path, final, arc_offset, arcs, start=0):
    """OpenFst VectorFst<StdArc> binary file from the graph arrays of synth.*_graph() (vectorised: millions of states)."""
    with open(path, "wb") as f:
        f.write(vector_fst_bytes(final, arc_offset, arcs, start))


def write_const_fst(path, final, arc_offset, arcs, start=0):
    """OpenFst ConstFst<StdArc> binary file (what the reference's mkgraph leaves as HCLG.fst: `fstconvert(..., "const")`,
    VB/scr/utils/mkgraph.cpp:309; openfst const-fst.h: file version 2, unaligned): header, the states
    {final f32, first arc u32, #arcs u32, #input-epsilons u32, #output-epsilons u32}, the arcs."""
    S = len(final)
    a = np.ascontiguousarray(arcs)
    A = len(a)
    off = np.asarray(arc_offset, np.int64)
    narcs = np.diff(off)

    def s_(x):
        return np.int32(len(x)).tobytes() + x
    hdr = (np.int32(2125659606).tobytes() + s_(b"const") + s_(b"standard") + np.int32(2).tobytes() + np.int32(0).tobytes() +
           np.uint64(3).tobytes() + np.int64(start).tobytes() + np.int64(S).tobytes() + np.int64(A).tobytes())
    st = np.zeros(S, np.dtype([("final", "<f4"), ("pos", "<u4"), ("narcs", "<u4"), ("nie", "<u4"), ("noe", "<u4")]))
    st["final"] = np.asarray(final, "<f4"); st["pos"] = off[:-1]; st["narcs"] = narcs
    if A:
        state_of_arc = np.repeat(np.arange(S, dtype=np.int64), narcs)
        st["nie"] = np.bincount(state_of_arc, weights=(a["ilabel"] == 0), minlength=S).astype(np.uint32)
        st["noe"] = np.bincount(state_of_arc, weights=(a["olabel"] == 0), minlength=S).astype(np.uint32)
    with open(path, "wb") as f:
        f.write(hdr); f.write(st.tobytes()); f.write(a.tobytes())


def write_fst_table(path, graphs):
    """Binary archive of VectorFst<StdArc> (what compile-train-graphs writes and gmm-align-compiled reads: key, space, the FST
    as OpenFst writes it -- VectorFstTplHolder, K/fstext/kaldi-fst-io-inl.h:170-207).  graphs: {key: (final, arc_offset, arcs, start)}."""
    with open(path, "wb") as f:
        for k, (final, arc_offset, arcs, start) in graphs.items():
            f.write(k.encode() + b" " + vector_fst_bytes(final, arc_offset, arcs, start))


def vector_fst_bytes(final, arc_offset, arcs, start=0):
    S = len(final)
    a = np.ascontiguousarray(arcs)
    A = len(a)
    off = np.asarray(arc_offset, np.int64)
    narcs = np.diff(off)

    def s_(x):
        return np.int32(len(x)).tobytes() + x
    hdr = (np.int32(2125659606).tobytes() + s_(b"vector") + s_(b"standard") + np.int32(2).tobytes() + np.int32(0).tobytes() +
           np.uint64(3).tobytes() + np.int64(start).tobytes() + np.int64(S).tobytes() + np.int64(A).tobytes())
    body = np.zeros(12 * S + 16 * A, np.uint8)
    spos = 12 * np.arange(S, dtype=np.int64) + 16 * off[:-1]          # byte position of every state's {final f32, narcs i64}
    fin = np.asarray(final, "<f4").view(np.uint8).reshape(S, 4)
    na = narcs.astype("<i8").view(np.uint8).reshape(S, 8)
    idx = spos[:, None] + np.arange(12)[None, :]
    body[idx[:, :4]] = fin
    body[idx[:, 4:]] = na
    if A:
        state_of_arc = np.repeat(np.arange(S, dtype=np.int64), narcs)
        apos = 12 * (state_of_arc + 1) + 16 * np.arange(A, dtype=np.int64)
        ab = a.view(np.uint8).reshape(A, 16)
        body[apos[:, None] + np.arange(16)[None, :]] = ab
    return hdr + body.tobytes()


def write_feats_ark(path, feats):
    with open(path, "wb") as f:
        for k, m in feats.items():
            f.write(k.encode() + b" \0B" + _fmat(np.asarray(m, "<f4")))


def write_wav(path, samples, rate=16000):
    with _wave.open(str(path), "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(rate)
        f.writeframes(np.asarray(samples, "<i2").tobytes())


def write_raw_lattice_ark(path, lattices, graph_start=0):
    """Binary archive of state-level lattices (OpenFst VectorFst<LatticeArc>, arc type "lattice4") from search results in the
    canonical form of vbh_lattice / the oracle (tokens sorted by (frame, HCLG state); links by source): one state per
    token, the start token becomes state 0, arcs (ilabel, olabel, (graph, acoustic)), final weights (final cost, 0) --
    what GetRawLattice builds (K/decoder/lattice-faster-decoder.cc:110-192).  lattices: {key: result dict}."""
    def s_(x):
        return np.int32(len(x)).tobytes() + x
    with open(path, "wb") as f:
        for key, r in lattices.items():
            tf, ts = np.asarray(r["tok_frame"]), np.asarray(r["tok_state"])
            n = len(tf)
            start_idx = 0
            for i in range(n):
                if tf[i] != 0:
                    break
                if ts[i] == graph_start:
                    start_idx = i
            remap = np.arange(n)
            remap[:start_idx] += 1
            if n:
                remap[start_idx] = 0
            src = remap[np.asarray(r["link_src"], np.int64)]
            dst = remap[np.asarray(r["link_dst"], np.int64)]
            order = np.argsort(src, kind="stable")
            cnt = np.bincount(src, minlength=n)
            fin = np.full(n, np.inf, np.float32)
            fin[remap] = np.asarray(r["tok_final"], np.float32)
            arcs = np.zeros(len(src), dtype=[("il", "<i4"), ("ol", "<i4"), ("g", "<f4"), ("a", "<f4"), ("nx", "<i4")])
            arcs["il"] = np.asarray(r["link_ilabel"])[order]
            arcs["ol"] = np.asarray(r["link_olabel"])[order]
            arcs["g"] = np.asarray(r["link_graph"], np.float32)[order]
            arcs["a"] = np.asarray(r["link_acoustic"], np.float32)[order]
            arcs["nx"] = dst[order]
            f.write(key.encode() + b" ")
            f.write(np.int32(2125659606).tobytes() + s_(b"vector") + s_(b"lattice4") + np.int32(2).tobytes() + np.int32(0).tobytes() +
                    np.uint64(3).tobytes() + np.int64(0 if n else -1).tobytes() + np.int64(n).tobytes() + np.int64(len(src)).tobytes())
            pos = 0
            for s in range(n):
                w0 = fin[s]
                f.write(np.float32(w0).tobytes() + np.float32(0.0 if np.isfinite(w0) else np.inf).tobytes() + np.int64(cnt[s]).tobytes())
                f.write(arcs[pos:pos + cnt[s]].tobytes())
                pos += int(cnt[s])
