"""Speech-like synthetic corpus for the bench: utterances that SAY something.

The tone-plus-noise waveforms of synth.wave() exercise the kernels but decode to a word or two per 30 s, so lattices,
posteriors and WER say little about the search at a real recogniser's operating point.  Here an utterance is a word
sequence drawn from the lexicon graph's own unigram distribution; every word is the path of HMM states (pdfs) the graph
spells it with; every HMM state lasts a geometric number of 10 ms hops; and every hop is noise shaped by its pdf's
spectral envelope (a random low-order cepstrum on the mel scale), times a per-speaker envelope.  The acoustic model is
then FIT to such data through the product's own front end (fit_gmm): per-pdf mean and variance of the labelled frames,
spread over the pdf's mixture components.  `depth` (the standard deviation of the pdfs' cepstra, in nats) is the
difficulty knob: it sets how far apart the pdfs' features lie relative to the estimation noise of a 25 ms spectrum.

Pure numpy + torch; no reference code involved.  Everything is seeded: a given (id, device type) yields the same samples.
"""
import numpy as np

HOP = 160                 # samples per 10 ms hop at 16 kHz = the MFCC frame shift
NBIN = HOP // 2 + 1       # rfft bins of one hop (100 Hz apart)
N_CEP = 12


def _mel(f):
    return 1127.0 * np.log(1.0 + f / 700.0)


def cepstral_basis(sr=16000):
    """[N_CEP x NBIN] cosines on the mel scale: log-envelope(bin) = cepstrum @ basis"""
    x = _mel(np.arange(NBIN) * (sr / HOP)) / _mel(sr / 2.0)
    return np.cos(np.pi * np.outer(np.arange(1, N_CEP + 1), x)).astype(np.float32)


def pdf_gains(num_pdfs, depth, seed=21, hmm_states=3, dtype=np.float32):
    """amplitude gain per (pdf, rfft bin) [P x NBIN]: exp(0.5 * log-power envelope).  The HMM states of one phone share a
    phone-level cepstrum (half the variance) and differ by a state-level one, so neighbouring states sound alike."""
    rng = np.random.default_rng(seed)
    n_ph = -(-num_pdfs // hmm_states)
    ph = rng.standard_normal((n_ph, N_CEP))
    st = rng.standard_normal((n_ph * hmm_states, N_CEP))
    c = (np.repeat(ph, hmm_states, 0) + st)[:num_pdfs] * (depth / np.sqrt(2.0))
    # higher cepstral coefficients carry less: 1/sqrt(k) taper as in speech
    c = c / np.sqrt(np.arange(1, N_CEP + 1))[None, :] * np.sqrt(N_CEP / np.sum(1.0 / np.arange(1, N_CEP + 1)))
    g = np.exp(0.5 * (c @ cepstral_basis().astype(np.float64)))
    g[:, 0] = 0.0          # no DC
    return g.astype(dtype)


def speaker_gains(spk_ids, depth=0.25, seed=33, dtype=np.float32):
    """per-speaker envelope [n x NBIN] (channel / vocal-tract colour: what CMVN and fMLLR are there to remove)"""
    out = np.empty((len(spk_ids), NBIN), dtype)
    basis = cepstral_basis()[:4].astype(np.float64)
    for k, s in enumerate(spk_ids):
        rng = np.random.default_rng(seed + 15485863 * int(s))
        c = rng.standard_normal(4) * depth
        lvl = rng.uniform(-0.5, 0.5)
        out[k] = np.exp(0.5 * (c @ basis) + lvl)
    return out


def transcripts(lex, ids, n_hops, seed=55, uniform=False, mean_dur=3.0):
    """Word sequences and hop labels for the utterances `ids`.  Words are drawn from the graph's unigram distribution
    (p ~ exp(-lm_cost); uniform=True: every word alike, for training data that covers the phones evenly); an HMM state lasts
    a geometric number of hops (mean mean_dur); the last word is dropped and the hops that are left are spread over the
    states, so that every utterance ends at a word end.  Returns (hop_pdf [B x n_hops] int32, [B] lists of word ids, 1-based
    as the graph's olabels)."""
    off, wp = lex["word_off"], lex["word_pdf"]
    W = len(off) - 1
    if uniform:
        cdf = None
    else:
        p = np.exp(-(lex["lm_cost"].astype(np.float64) - float(np.min(lex["lm_cost"]))))
        cdf = np.cumsum(p / p.sum())
    hop_pdf = np.empty((len(ids), n_hops), np.int32)
    words = []
    for k, i in enumerate(ids):
        rng = np.random.default_rng(seed + 104729 * int(i))
        n_draw = max(8, int(2.5 * n_hops / (mean_dur * 15)))
        while True:
            ws = rng.integers(0, W, n_draw) if cdf is None else np.minimum(np.searchsorted(cdf, rng.random(n_draw)), W - 1)
            n_st = off[ws + 1] - off[ws]
            st_end = np.cumsum(n_st)
            dur = rng.geometric(1.0 / mean_dur, size=int(st_end[-1]))
            t_end = np.cumsum(dur)[st_end - 1]              # hop at which each word ends
            n_ok = int(np.searchsorted(t_end, n_hops, side="right"))     # words that fit completely
            if n_ok < n_draw:
                break
            n_draw *= 2
        n_ok = max(n_ok, 1)
        ws = ws[:n_ok]
        n_states = int(st_end[n_ok - 1])
        dur = dur[:n_states]
        if n_ok == 1 and dur.sum() > n_hops:                # a single word longer than the utterance: squeeze it
            dur = np.ones(n_states, np.int64)
        left = n_hops - int(dur.sum())
        if left > 0:
            dur = dur + rng.multinomial(left, np.full(n_states, 1.0 / n_states))
        states = np.concatenate([wp[off[w]:off[w + 1]] for w in ws])
        hop_pdf[k] = np.repeat(states, dur)[:n_hops]
        words.append((ws + 1).astype(np.int32))
    return hop_pdf, words


def waves_device(hop_pdf, gains, ids, device, spk_gain=None, seed=4321, level=22000.0):
    """int16 [B x n_hops*HOP] on `device`: per hop, complex Gaussian noise on the hop's rfft bins times the pdf's (and the
    speaker's) gain, transformed back.  gains: torch [P x NBIN] on the device; spk_gain: None or torch [B x NBIN]."""
    import torch
    B, H = hop_pdf.shape
    lab = torch.as_tensor(hop_pdf, device=device).long()
    out = torch.empty((B, H * HOP), dtype=torch.int16, device=device)
    g = torch.Generator(device=device)
    # the inverse real FFT of a hop as two matrix products (x = Re Z . C - Im Z . S): plain GEMMs instead of an FFT library call
    n = torch.arange(HOP, device=device, dtype=torch.float64)[None, :]
    kk = torch.arange(NBIN, device=device, dtype=torch.float64)[:, None]
    wgt = torch.full((NBIN, 1), 2.0 / HOP, device=device, dtype=torch.float64)
    wgt[0, 0] = 1.0 / HOP
    wgt[NBIN - 1, 0] = 1.0 / HOP
    Cm = (wgt * torch.cos(2.0 * np.pi * kk * n / HOP)).float()
    Sm = (wgt * torch.sin(2.0 * np.pi * kk * n / HOP)).float()
    Sm[0] = 0.0
    Sm[NBIN - 1] = 0.0
    for k in range(B):
        g.manual_seed(seed + 7919 * int(ids[k]))
        z = torch.randn((H, NBIN, 2), generator=g, device=device, dtype=torch.float32)
        amp = gains[lab[k]]
        if spk_gain is not None:
            amp = amp * spk_gain[k][None, :]
        x = ((z[..., 0] * amp) @ Cm - (z[..., 1] * amp) @ Sm).reshape(-1)
        out[k] = torch.clamp(torch.round(x * level), -32768, 32767).to(torch.int16)
    return out


def waves_numpy(hop_pdf, gains, ids, spk_gain=None, seed=4321, level=22000.0):
    """The same construction in float64 numpy (gains, spk_gain: float64 arrays): int16 [B x n_hops*HOP].  For fixtures that
    must come out the same on every machine (numpy's generator and pocketfft; no device RNG, no float32 rounding on the
    way); not the samples waves_device() draws for the same ids."""
    B, H = hop_pdf.shape
    out = np.empty((B, H * HOP), np.int16)
    for k in range(B):
        rng = np.random.default_rng(seed + 7919 * int(ids[k]))
        z = rng.standard_normal((H, NBIN, 2))
        amp = gains[hop_pdf[k]]
        if spk_gain is not None:
            amp = amp * spk_gain[k][None, :]
        x = np.fft.irfft((z[..., 0] + 1j * z[..., 1]) * amp, n=HOP, axis=1).reshape(-1)
        out[k] = np.clip(np.round(x * level), -32768, 32767).astype(np.int16)
    return out


def frame_labels(hop_pdf, n_frames):
    """pdf of each 25 ms frame = the pdf of the hop at its centre (frame t covers hops t .. t+2.5)"""
    return hop_pdf[:, 1:1 + n_frames]


class GmmFit:
    """running per-pdf sums of labelled frames (on the device), then a mixture per pdf"""

    def __init__(self, num_pdfs, dim, device):
        import torch
        self.P, self.D = num_pdfs, dim
        self.n = torch.zeros(num_pdfs, dtype=torch.float64, device=device)
        self.s1 = torch.zeros((num_pdfs, dim), dtype=torch.float64, device=device)
        self.s2 = torch.zeros((num_pdfs, dim), dtype=torch.float64, device=device)

    def add(self, feats, labels):
        import torch
        lab = torch.as_tensor(labels, device=feats.device).long().reshape(-1)
        f = feats.double()
        self.n.index_add_(0, lab, torch.ones_like(lab, dtype=torch.float64))
        self.s1.index_add_(0, lab, f)
        self.s2.index_add_(0, lab, f * f)

    def stats(self, min_frames=8):
        """(mean [P x D], variance [P x D], number of pdfs with fewer than min_frames frames: they get the global mean / variance)"""
        n = self.n.cpu().numpy()
        tot = max(float(n.sum()), 1.0)
        g_mu = (self.s1.sum(0) / tot).cpu().numpy()
        g_var = np.maximum((self.s2.sum(0) / tot).cpu().numpy() - g_mu ** 2, 1e-6)
        ok = n >= min_frames
        nn = np.maximum(n, 1.0)[:, None]
        mu = np.where(ok[:, None], self.s1.cpu().numpy() / nn, g_mu[None, :])
        var = np.where(ok[:, None], self.s2.cpu().numpy() / nn - mu ** 2, g_var[None, :])
        var = np.maximum(var, 0.05 * g_var[None, :])
        return mu.astype(np.float32), var.astype(np.float32), int((~ok).sum())

    def model(self, num_gauss, seed=1, spread=0.5, min_frames=8):
        """(the model as mixture_from_stats returns it, number of pdfs unseen in the data)"""
        mu, var, n_unseen = self.stats(min_frames)
        return mixture_from_stats(mu, var, num_gauss, seed, spread), n_unseen


def mixture_from_stats(mu, var, num_gauss, seed=1, spread=0.5):
    """(offs[P+1], weights[G], means_invvars[G x D], inv_vars[G x D]) as vbh_gmm_create takes them: a pdf's components sit at
    mean + spread * sigma * N(0,1) with variance (1 - spread^2) sigma^2, so the mixture keeps the pdf's mean and variance."""
    from voicebridge_amd import synth
    P, D = mu.shape
    offs, w, _m, _iv = synth.gmm_model(P, num_gauss, D, seed=seed)
    rng = np.random.default_rng(seed + 1000)
    pdf_of = np.repeat(np.arange(P), np.diff(offs))
    sd = np.sqrt(var.astype(np.float64))[pdf_of]
    means = (mu.astype(np.float64)[pdf_of] + spread * sd * rng.standard_normal((num_gauss, D))).astype(np.float32)
    iv = (1.0 / ((1.0 - spread * spread) * var.astype(np.float64)[pdf_of])).astype(np.float32)
    return offs, w, (means * iv).astype(np.float32), iv
