"""Which of the 64 utterances of tests/golden/speech_e2e.npz does the admission rule move?  Oracle chain from PCM (pinned to the
reference's programs), search in the reference's order (exact_cutoff=0) and in the order-independent mode the GPU computes
(exact_cutoff=1), words of both against the reference's programs'.  CPU only, build container or anywhere."""
import os, sys, time
import numpy as np
from concurrent.futures import ProcessPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def work(args):
    lo, hi = args
    from oracle import vbo, gen_golden_e2e as e2e, gen_golden_speech as gs
    from voicebridge_amd import synth
    g, t2p, lex = synth.lexicon_graph(e2e.WORDS, e2e.P, seed=11, return_lexicon=True)
    ids, spk, waves, hop, truth = e2e.test_set(lex)
    offs, w, miv, iv = e2e.model()
    gc = vbo.gmm_gconsts(w, miv, iv)
    fst = vbo.make_fst(g["final"], g["arc_offset"], g["arcs"], g["start"])
    o = vbo.mfcc_opts()
    out = {}
    for s_ in range(lo, hi):          # speakers
        us = [u for u in ids if spk[u] == s_]
        raws = {u: vbo.mfcc(waves[u].astype(np.float32), o) for u in us}
        st = sum(vbo.acc_cmvn_stats(raws[u]) for u in us)
        xf = synth.affine(e2e.D, seed=int(s_), scale=gs.XFORM_SCALE)
        for u in us:
            f = vbo.transform_feats(vbo.add_deltas(vbo.apply_cmvn(raws[u], st)), xf)
            ll = vbo.gmm_loglikes(offs, gc, miv, iv, f)
            r0 = vbo.decode_lattice_faster(fst, ll, t2p, exact_cutoff=False, **e2e.CFG)
            r1 = vbo.decode_lattice_faster(fst, ll, t2p, exact_cutoff=True, **e2e.CFG)
            r2 = vbo.decode_lattice_faster(fst, ll, t2p, exact_cutoff=2, **e2e.CFG)
            out[int(u)] = (np.asarray(r0["words"]), np.asarray(r1["words"]), float(r0["best_graph"] + r0["best_acoustic"]), float(r1["best_graph"] + r1["best_acoustic"]), truth[u],
                           np.asarray(r2["words"]), float(r2["best_graph"] + r2["best_acoustic"]), [int(r["n_tok_created"]) for r in (r0, r1, r2)], [int(r["n_links_created"]) for r in (r0, r1, r2)])
    return out

if __name__ == "__main__":
    t0 = time.time()
    fx = np.load(os.path.join(ROOT, "tests", "golden", "speech_e2e.npz"))
    with ProcessPoolExecutor(8) as ex:
        parts = list(ex.map(work, [(2 * i, 2 * i + 2) for i in range(8)]))
    res = {}
    for p in parts: res.update(p)
    from voicebridge_amd import dist as vdist
    n0 = n1 = n01 = n2 = 0
    toks = np.zeros(3); links = np.zeros(3)
    for u in sorted(res):
        w0, w1, c0, c1, tr, w2, c2, nt, nl = res[u]
        toks += nt; links += nl
        e2 = np.array_equal(w2, fx["ref_words_%d" % u]); n2 += e2
        if not e2:
            print("utt %d: superset rule differs from the programs: path cost %.4f (ref-order %.4f)" % (u, c2, c0))
        ref = fx["ref_words_%d" % u]
        e0, e1 = np.array_equal(w0, ref), np.array_equal(w1, ref)
        n0 += e0; n1 += e1; n01 += np.array_equal(w0, w1)
        if not (e0 and e1):
            er = lambda h: sum(vdist.edit_counts(tr, h))
            print("utt %d: reference-order == ref programs: %s; order-independent == ref programs: %s; path cost ref-order %.4f order-indep %.4f; errors ref %d, ref-order %d, order-indep %d"
                  % (u, e0, e1, c0, c1, er(ref), er(w0), er(w1)))
    print("superset rule (admission bound = the reference's initial next_cutoff): the programs' words on %d; tokens created ref-order / order-indep / superset %s, links %s" % (n2, toks, links))
    print("of %d utterances: oracle in reference order gives the programs' words on %d, order-independent on %d; the two modes agree on %d  [%.0f s]" % (len(res), n0, n1, n01, time.time() - t0))
