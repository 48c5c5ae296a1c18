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


def wave_batch_device(ids, seconds, device, sr=16000, seed=1234):
    """The same family of utterances as wave(), synthesised on the device (torch), one distinct waveform per id: tone
    frequency, modulation rate and the noise stream all depend on the id.  Returns an int16 tensor [len(ids) x n] on
    `device`.  The noise comes from the device's Philox generator seeded per utterance, so a given (id, device type)
    always yields the same samples; they are not the samples wave() draws with numpy for that id."""
    import torch
    n = int(round(seconds * sr))
    out = torch.empty((len(ids), n), dtype=torch.int16, device=device)
    t = torch.arange(n, dtype=torch.float64, device=device) / sr
    g = torch.Generator(device=device)
    for k, i in enumerate(ids):
        i = int(i)
        g.manual_seed(seed + 7919 * i)
        f0 = 200.0 + 5.0 * (i % 400) + 0.37 * ((i // 400) % 13)
        x = 0.3 * torch.sin(2 * np.pi * f0 * t) + 0.1 * torch.randn(n, generator=g, device=device, dtype=torch.float32).double()
        x = x * (0.6 + 0.4 * torch.sin(2 * np.pi * (0.7 + 0.01 * (i % 50) + 0.003 * ((i // 50) % 17)) * t))
        out[k] = torch.clamp(torch.round(x * 8000.0), -32768, 32767).to(torch.int16)
    return out


def chirp(seconds=2.0, sr=16000):
    n = int(seconds * sr)
    t = np.arange(n) / sr
    x = 0.5 * np.sin(2 * np.pi * (100 * t + 0.5 * 3500 / seconds * t * t))
    return np.round(x * 12000).astype(np.int16)


def gmm_model(P, G, D, seed=0, mean_scale=1.0, dim_scale=None, zero_weight=False, var_spread=1.0):
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
    var = (np.exp(rng.standard_normal((G, D)) * var_spread) + 1e-5).astype(np.float32)
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


def lexicon_graph(num_words, num_pdfs, seed=0, n_phones=40, min_len=3, max_len=9, hmm_states=3, return_lexicon=False):
    """HCLG-shaped graph with the recombination structure of a real decoding graph: a lexicon
    prefix tree over random phone strings, every tree node a left-to-right HMM of `hmm_states`
    states (self-loop + forward arc), word ends returning to the root through one
    input-epsilon arc that carries the word label and a Zipf-like unigram cost.

    Returns (graph dict as decoding_graph(), tid2pdf).  Transition-ids: pdf p has the
    self-loop tid 2p+1 and the forward tid 2p+2 (hmm/transition-model.h: two transitions per
    emitting HMM state); each tree node draws a context-dependent phone, i.e. `hmm_states`
    consecutive pdfs.

    return_lexicon: also return dict(word_off[W+1], word_pdf[...], lm_cost[W]): the pdf sequence a word's path through
    the graph emits (one entry per HMM state), for synth_speech's utterances; the graph is the same either way.
    """
    rng = np.random.default_rng(seed)
    n_cd = num_pdfs // hmm_states
    # ---- prefix tree: child[(node, phone)] -> node
    lens = rng.integers(min_len, max_len + 1, size=num_words)
    phones = rng.integers(0, n_phones, size=int(lens.sum()))
    child = {}
    node_cd = [0]          # node 0 = root (no HMM)
    word_end_node = np.zeros(num_words, np.int64)
    word_nodes = np.zeros(int(lens.sum()), np.int64) if return_lexicon else None
    pos = 0
    for w in range(num_words):
        cur = 0
        for k in range(lens[w]):
            key = (cur, int(phones[pos + k]))
            nxt = child.get(key)
            if nxt is None:
                nxt = len(node_cd)
                child[key] = nxt
                node_cd.append(0)
            cur = nxt
            if word_nodes is not None:
                word_nodes[pos + k] = cur
        pos += lens[w]
        word_end_node[w] = cur
    n_nodes = len(node_cd)
    node_cd = rng.integers(0, n_cd, size=n_nodes)
    # ---- states: root = 0; node n (>= 1) owns states 1 + (n-1)*H .. ; word-end states after that
    H = hmm_states
    first_state = lambda n: 1 + (n - 1) * H
    n_hmm_states = (n_nodes - 1) * H
    end_state0 = 1 + n_hmm_states
    S = end_state0 + num_words
    # children lists
    kids = [[] for _ in range(n_nodes)]
    for (parent, _ph), c in child.items():
        kids[parent].append(c)
    ends = [[] for _ in range(n_nodes)]
    for w in range(num_words):
        ends[word_end_node[w]].append(w)
    self_w, fwd_w = np.float32(-np.log(0.6)), np.float32(-np.log(0.4))
    lm_cost = (np.log(1.0 + rng.permutation(num_words)) * 1.1 + 1.0).astype(np.float32)   # Zipf-like
    arc_lists = [[] for _ in range(S)]

    def enter_arcs(n, from_pdf_tid):
        """arcs that leave a state towards the first HMM state of node n's children / word ends"""
        out = []
        for c in kids[n]:
            out.append((from_pdf_tid, 0, fwd_w, first_state(c)))
        for w in ends[n]:
            out.append((from_pdf_tid, 0, fwd_w, end_state0 + w))
        return out

    # the root enters its children through the children's own first emitting arc: model the
    # root as a non-emitting hub: epsilon arcs root -> first state of each first phone
    for c in kids[0]:
        arc_lists[0].append((0, 0, np.float32(0.0), first_state(c)))
    for n in range(1, n_nodes):
        base = first_state(n)
        for h in range(H):
            pdf = int(node_cd[n]) * H + h
            st = base + h
            arc_lists[st].append((2 * pdf + 1, 0, self_w, st))
            if h + 1 < H:
                arc_lists[st].append((2 * pdf + 2, 0, fwd_w, st + 1))
            else:
                arc_lists[st].extend(enter_arcs(n, 2 * pdf + 2))
    for w in range(num_words):
        arc_lists[end_state0 + w].append((0, w + 1, lm_cost[w], 0))
    n_arcs = np.array([len(a) for a in arc_lists], np.int64)
    arc_offset = np.zeros(S + 1, np.int64)
    arc_offset[1:] = np.cumsum(n_arcs)
    arcs = np.zeros(int(arc_offset[-1]), ARC_DTYPE)
    flat = [a for lst in arc_lists for a in lst]
    arcs["ilabel"] = [a[0] for a in flat]
    arcs["olabel"] = [a[1] for a in flat]
    arcs["weight"] = [a[2] for a in flat]
    arcs["nextstate"] = [a[3] for a in flat]
    final = np.full(S, np.inf, np.float32)
    final[0] = 0.0
    num_tids = 2 * num_pdfs
    t2p = np.empty(num_tids + 1, np.int32)
    t2p[0] = -1
    t2p[1:] = np.repeat(np.arange(num_pdfs, dtype=np.int32), 2)
    g = dict(final=final, arc_offset=arc_offset, arcs=arcs, start=0)
    if not return_lexicon:
        return g, t2p
    word_off = np.zeros(num_words + 1, np.int64)
    word_off[1:] = np.cumsum(lens) * H
    word_pdf = (node_cd[word_nodes][:, None] * H + np.arange(H)[None, :]).reshape(-1).astype(np.int32)
    return g, t2p, dict(word_off=word_off, word_pdf=word_pdf, lm_cost=lm_cost, hmm_states=H)


def lda_matrix():
    """a 40 x 117 LDA+MLLT-like matrix (final.mat) over 9 spliced frames of 13 cepstra (BASELINE config 4's feature chain: splice +-4,
    VB/src/featbin/splice-feats.cpp:30, then transform-feats final.mat, VB/scr/steps/decode_gmm.cpp:243-267).  Not a random projection
    (that throws away two thirds of the information and the recogniser with it: WER 27 % against 5.5 %): the rows compute what a
    discriminant projection of spliced frames mostly amounts to -- the centre frame, its first and second regression derivatives (the
    delta windows of K/feat/feature-functions.cc:54-110 written as one linear map) -- plus the mean of C0 over the splice, so the
    recogniser works as well as with deltas and the kernels do the 117 -> 40 contraction."""
    m = np.zeros((40, 9, 13), np.float32)
    d1 = np.array([-2.0, -1.0, 0.0, 1.0, 2.0]) / 10.0
    d2 = np.convolve(d1, d1)                      # 9 taps: offsets -4..4
    for c in range(13):
        m[c, 4, c] = 1.0
        m[13 + c, 2:7, c] = d1
        m[26 + c, :, c] = d2
    m[39, :, 0] = 1.0 / 9.0
    return np.ascontiguousarray(m.reshape(40, 117))
