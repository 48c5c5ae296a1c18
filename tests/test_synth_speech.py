"""The bench's speech-like workload (voicebridge_amd/synth_speech.py): transcripts follow the lexicon graph, hop labels are
consistent with them, the numpy synthesis is the one the speech fixture was made with."""
import numpy as np
import pytest

from tests.util import load
from voicebridge_amd import synth, synth_speech as ss


@pytest.fixture(scope="module")
def lex_graph():
    return synth.lexicon_graph(3000, 300, seed=11, return_lexicon=True)


def test_lexicon_is_the_graph(lex_graph):
    """return_lexicon changes nothing in the graph; a word's pdf sequence is a path of the graph that ends in the word's label"""
    g, t2p, lex = lex_graph
    g0, t0 = synth.lexicon_graph(3000, 300, seed=11)
    for k in ("final", "arc_offset", "arcs"):
        assert np.array_equal(g[k], g0[k])
    assert np.array_equal(t2p, t0)
    ao, arcs = g["arc_offset"], g["arcs"]

    def pdf_of(state):          # pdf of an emitting state = pdf of its self-loop (its first arc); -1 for the non-emitting ones
        a = arcs[ao[state]]
        return int(t2p[a["ilabel"]]) if a["ilabel"] != 0 else -1

    def walk(state, seq, i, w):  # several children of a node may carry the same phone: try them all
        if pdf_of(state) != seq[i]:
            return False
        out = arcs[ao[state] + 1:ao[state + 1]]
        if i + 1 == len(seq):
            return any(arcs[ao[x["nextstate"]]]["olabel"] == w + 1 for x in out)
        return any(walk(int(x["nextstate"]), seq, i + 1, w) for x in out)

    for w in (0, 17, 2999):
        seq = [int(p) for p in lex["word_pdf"][lex["word_off"][w]:lex["word_off"][w + 1]]]
        assert len(seq) % lex["hmm_states"] == 0 and len(seq) >= 9
        assert any(walk(int(a["nextstate"]), seq, 0, w) for a in arcs[ao[0]:ao[1]]), w


def test_transcripts_fill_the_utterance_with_whole_words(lex_graph):
    _g, _t2p, lex = lex_graph
    n_hops = 1500
    hop, words = ss.transcripts(lex, np.arange(6), n_hops)
    hop2, words2 = ss.transcripts(lex, np.arange(6), n_hops)
    assert np.array_equal(hop, hop2) and all(np.array_equal(a, b) for a, b in zip(words, words2))      # seeded
    for u in range(6):
        seq = np.concatenate([lex["word_pdf"][lex["word_off"][w - 1]:lex["word_off"][w]] for w in words[u]])
        runs = hop[u][np.flatnonzero(np.r_[True, hop[u][1:] != hop[u][:-1]])]
        # the hop labels are the words' states in order, every state at least one hop (equal neighbours merge in `runs`)
        j = 0
        for p in seq:
            if j < len(runs) and runs[j] == p:
                j += 1
            else:
                assert j > 0 and runs[j - 1] == p
        assert j == len(runs)
        assert len(words[u]) > 10
    hu, wu = ss.transcripts(lex, np.arange(3), n_hops, uniform=True)
    assert not np.array_equal(hu[0], hop[0])


def test_waves_numpy_is_what_the_speech_fixture_was_made_with():
    """float64 numpy synthesis: the same samples on every machine (the GPU test regenerates them instead of storing 7.7 MB)"""
    from oracle import gen_golden_speech as sp
    fx = load("speech_refpaths")
    # the fixture's graph takes a minute to build; the first utterance's transcript needs only the lexicon
    g, t2p, lex = synth.lexicon_graph(sp.WORDS, sp.P, seed=11, return_lexicon=True)
    ids, spk, waves, hop, truth = sp.test_set(lex)
    crc = [int(np.bitwise_xor.reduce(w.view(np.uint16).astype(np.uint64) * (np.arange(w.size, dtype=np.uint64) % 65521 + 1))) for w in waves]
    assert crc == [int(x) for x in fx["waves_crc"]]
    assert waves.dtype == np.int16 and waves.shape == (sp.N_UTT, sp.N_HOPS * ss.HOP)
    assert 1500 < waves.astype(np.float64).std() < 6000


def test_gmm_fit_keeps_mean_and_variance():
    import torch
    rng = np.random.default_rng(0)
    P, D = 7, 5
    mu = rng.standard_normal((P, D))              # (spread small enough for the variance floor, 5 % of the global variance, not to bind)
    sd = rng.uniform(0.5, 2.0, (P, D))
    lab = rng.integers(0, P - 1, 40000)            # pdf P-1 is never seen
    x = mu[lab] + sd[lab] * rng.standard_normal((40000, D))
    fit = ss.GmmFit(P, D, "cpu")
    fit.add(torch.from_numpy(x[:20000]), lab[:20000])
    fit.add(torch.from_numpy(x[20000:]), lab[20000:])
    m, v, unseen = fit.stats()
    assert unseen == 1
    np.testing.assert_allclose(m[:P - 1], mu[:P - 1], atol=0.08)
    np.testing.assert_allclose(np.sqrt(v[:P - 1]), sd[:P - 1], rtol=0.05)
    offs, w, miv, iv = ss.mixture_from_stats(m, v, 40)
    assert offs[-1] == 40 and np.all(iv > 0)
    for p in range(P - 1):
        sl = slice(offs[p], offs[p + 1])
        np.testing.assert_allclose(w[sl].sum(), 1.0, rtol=1e-5)
