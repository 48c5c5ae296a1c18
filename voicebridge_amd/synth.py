"""Seeded synthetic inputs for the hot path: 16 kHz PCM-16 utterances,
random diagonal-GMM acoustic models and HCLG-shaped decoding graphs.

Used by tests/, bench.py and oracle/gen_golden.py.  Pure numpy; no reference code involved.
"""
import numpy as np

ARC_DTYPE = np.dtype([("ilabel", "<i4"), ("olabel", "<i4"), ("weight", "<f4"), ("nextstate", "<i4")])


def wave(i, seconds=30.0, sr=16000, seed=1234):
    """Utterance i: 0.3 sin(2 pi (200+5i) t) + 0.1 N(0,1), scaled by 8000, int16."""
    n = int(round(seconds * sr))
    rng = np.random.default_rng(seed + 7919 * i)
    t = np.arange(n, dtype=np.float64) / sr
    x = 0.3 * np.sin(2 * np.pi * (200 + 5 * (i % 400)) * t) + 0.1 * rng.standard_normal(n)
    # slow amplitude modulation so that frames differ in energy like speech does
    x *= 0.6 + 0.4 * np.sin(2 * np.pi * (0.7 + 0.01 * (i % 50)) * t)
    return np.clip(np.round(x * 8000.0), -32768, 32767).astype(np.int16)


def chirp(seconds=2.0, sr=16000):
    n = int(seconds * sr)
    t = np.arange(n) / sr
    x = 0.5 * np.sin(2 * np.pi * (100 * t + 0.5 * 3500 / seconds * t * t))
    return np.round(x * 12000).astype(np.int16)


def gmm_model(P, G, D, seed=0, mean_scale=1.0, dim_scale=None, zero_weight=False):
    """Random AmDiagGmm arrays: (offs[P+1], weights[G], means[G,D], inv_vars[G,D]).

    Means N(0,1)*mean_scale, variances exp(N(0,1))+1e-5 (the distribution of
    gmm/diag-gmm-test.cc:107-111), weights uniform, normalised per pdf.  dim_scale (optional
    [D]) rescales each dimension (means by s, variances by s^2) to sit in a given feature space.
    """
    rng = np.random.default_rng(seed)
    avg = G / P
    lo, hi = max(1, int(avg // 2)), max(1, int(np.ceil(avg * 1.5)))
    cnt = rng.integers(lo, hi + 1, size=P)
    # adjust to the requested total
    diff = G - int(cnt.sum())
    while diff != 0:
        j = rng.integers(0, P)
        if diff > 0:
            cnt[j] += 1
            diff -= 1
        elif cnt[j] > 1:
            cnt[j] -= 1
            diff += 1
    offs = np.zeros(P + 1, np.int32)
    offs[1:] = np.cumsum(cnt)
    means = (rng.standard_normal((G, D)) * mean_scale).astype(np.float32)
    var = (np.exp(rng.standard_normal((G, D))) + 1e-5).astype(np.float32)
    if dim_scale is not None:
        s = np.asarray(dim_scale, np.float32)
        means = means * s
        var = var * s * s
    w = rng.uniform(0.1, 1.0, size=G).astype(np.float32)
    if zero_weight and cnt[0] > 1:
        w[offs[0]] = 0.0  # exercises the -inf gconst branch (gmm/diag-gmm.cc:141-146)
    for p in range(P):
        w[offs[p]:offs[p + 1]] /= w[offs[p]:offs[p + 1]].sum()
    inv_vars = (1.0 / var).astype(np.float32)
    return offs, w, means, inv_vars


def decoding_graph(S, num_tids, seed=0, min_deg=2, max_deg=4, eps_frac=0.25, olabel_frac=0.05,
                   num_words=20000, final_frac=0.05, max_weight=5.0, self_loops=True,
                   unreachable_final=False):
    """HCLG-shaped CSR graph.  Returns dict(final[S] f32 (inf = non-final), arc_offset[S+1] i64,
    arcs[A] ARC_DTYPE, start).

    Every state gets min_deg..max_deg emitting arcs (ilabel uniform over 1..num_tids; the first
    one is a self-loop when self_loops), eps_frac of the states get one extra input-epsilon arc
    to a higher-numbered state (so there are no epsilon cycles), olabel_frac of the arcs carry a
    word label, final_frac of the states are final with a U(0, max_weight) final cost.
    """
    rng = np.random.default_rng(seed)
    deg = rng.integers(min_deg, max_deg + 1, size=S)
    has_eps = (rng.random(S) < eps_frac) & (np.arange(S) < S - 1)
    n_arcs = deg + has_eps
    arc_offset = np.zeros(S + 1, np.int64)
    arc_offset[1:] = np.cumsum(n_arcs)
    A = int(arc_offset[-1])
    arcs = np.zeros(A, ARC_DTYPE)
    src = np.repeat(np.arange(S), n_arcs)
    pos = np.arange(A) - arc_offset[src]
    arcs["ilabel"] = rng.integers(1, num_tids + 1, size=A)
    arcs["nextstate"] = rng.integers(0, S, size=A)
    arcs["weight"] = rng.uniform(0.0, max_weight, size=A).astype(np.float32)
    ol = rng.random(A) < olabel_frac
    arcs["olabel"] = np.where(ol, rng.integers(1, num_words + 1, size=A), 0)
    if self_loops:
        first = pos == 0
        arcs["nextstate"][first] = src[first]
        arcs["olabel"][first] = 0
    # epsilon arc = last arc of the states that have one
    is_eps = has_eps[src] & (pos == n_arcs[src] - 1)
    arcs["ilabel"][is_eps] = 0
    e_src = src[is_eps]
    arcs["nextstate"][is_eps] = e_src + 1 + (rng.random(e_src.size) * (S - 1 - e_src)).astype(np.int64)
    arcs["nextstate"] = np.minimum(arcs["nextstate"], S - 1)
    final = np.full(S, np.inf, np.float32)
    if not unreachable_final:
        fs = rng.random(S) < final_frac
        final[fs] = rng.uniform(0.0, max_weight, size=int(fs.sum())).astype(np.float32)
        if not fs.any():
            final[S - 1] = 0.0
    return dict(final=final, arc_offset=arc_offset, arcs=arcs, start=0)


def tid2pdf(num_tids, P, seed=0):
    """TransitionIdToPdf table (hmm/transition-model.h:324-328): index 0 unused (-1)."""
    rng = np.random.default_rng(seed)
    t = np.empty(num_tids + 1, np.int32)
    t[0] = -1
    t[1:] = rng.integers(0, P, size=num_tids)
    return t


def affine(D, seed=0, scale=0.05):
    """Per-speaker fMLLR-like transform: identity + N(0, scale), offset N(0, scale); [D x (D+1)]."""
    rng = np.random.default_rng(seed)
    A = np.eye(D, D + 1, dtype=np.float32)
    A += (rng.standard_normal((D, D + 1)) * scale).astype(np.float32)
    return A
